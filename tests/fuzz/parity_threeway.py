#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (imports oracle/; lives under tests/ for that reason).  VERDICT r05 item 2: is the same-sign difference of the late training loss / held-out accuracy between the HIP path and the CPU
oracle chance (two correct fp32 implementations drifting apart through eight batch-normalised layers) or a bias of the HIP path?

Three implementations of the run of tests/test_gpu_accuracy_parity.py (120 steps of 6 patches of 20 x 20, Dilated8Pooling, then the
held-out tile labelled by sliding window with each side's own trained variables), PAIRED by seed -- same initial variables
(net.initial_params), same patches:
    fp64   oracle/torch_ref.py TorchNet in float64      (the reference arithmetic without rounding)
    torch  the same in float32                          (a second correct fp32 implementation: its distance from fp64 is what
                                                         fp32 rounding + chaos alone does to these statistics)
    hip    the product library (libdrs_hip.so)
The sides are independent given the seed, so they run where their hardware is and are merged afterwards:
    python tests/fuzz/parity_threeway.py side=hip   first=1000 seeds=600 out=gpurun_out/r06/threeway_hip.npz          (GPU box)
    python tests/fuzz/parity_threeway.py side=cpu   first=1000 seeds=600 out=threeway_cpu_0.npz [threads=8] [order=desc]   (any host; no GPU; resumes from `out`)
    python tests/fuzz/parity_threeway.py merge hip.npz cpu_a.npz cpu_b.npz ...                                        (report)
Per seed and side: loss of steps 0..7 (the trajectory before the chaos), mean loss of steps 20-59 and of steps 100-119, held-out accuracy.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NET, CH, K, B, S, STEPS, LR, WD = "dilated_grsl_rate8", 5, 6, 6, 20, 120, 0.01, 0.0005


def tiles():
    from drs_amd.synthetic import make_tile
    tile, lab = make_tile(160, 160, CH, K, seed=3, n_seeds=24, class_signal=0.6)
    held, held_lab = make_tile(96, 96, CH, K, seed=4, n_seeds=12, class_signal=0.6)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)), tile[:, :, :3].std(axis=(0, 1))
    return tile, lab, held, held_lab, mean, std


def stats(losses, acc):
    losses = np.asarray(losses, dtype=np.float64)
    return np.concatenate([losses[:8], [losses[20:60].mean(), losses[100:120].mean(), acc]])


def initial_variables(seed):
    """name -> array of the variables DilatedNet(seed=21 + seed) starts from, made on the host (no GPU)"""
    from drs_amd.net import initial_params
    from drs_amd.nets import Plan
    plan = Plan(NET, CH, K, first_cin_pad=8)
    host = initial_params(plan, 21 + seed)
    p = {}
    for name, (off, shape) in plan.offsets.items():
        p[name] = host[off:off + int(np.prod(shape))].reshape(shape).copy()
    for L in plan.layers:
        p[L.name + "/moving_mean"] = np.zeros(L.cout, dtype=np.float32)
        p[L.name + "/moving_variance"] = np.ones(L.cout, dtype=np.float32)
    return p


def run_cpu(seed, dtype, data):
    import torch
    from oracle import host_ref as H
    from oracle.torch_ref import TorchNet
    from drs_amd.synthetic import grid_instances
    tile, lab, held, held_lab, mean, std = data
    npdt = np.float32 if dtype == torch.float32 else np.float64
    inst = grid_instances(tile.shape[0], tile.shape[1], S, 8, B * STEPS, seed=100 + seed)
    t = TorchNet(NET, CH, K, params=initial_variables(seed), dtype=dtype)
    m5, s5 = list(mean) + [0, 0], list(std) + [1, 1]
    losses = []
    for i in range(STEPS):
        x, y, _ = H.dynamically_create_patches([tile], [lab], inst[i * B:(i + 1) * B], S, is_train=False)
        x = x.copy()
        H.normalize_images(x, m5, s5)
        # (the HIP path is fed float32 patches: both CPU sides start from the same float32 values)
        losses.append(t.train_step(x.astype(np.float32).astype(npdt), y, LR, WD)[0])
    st = H.stride_for(S)
    hh, hw = held_lab.shape
    nh, nw = H.window_counts(hh, hw, S, st)
    batches = []
    for i in range(-(-nh * nw // B)):
        p, _, pos = H.create_patches_per_map(held, held_lab, S, st, i, B)
        p = p.copy()
        H.normalize_images(p, m5, s5)
        batches.append((t.forward(p.astype(np.float32).astype(npdt), False).detach().numpy(), pos))
    _, _, pred = H.stitch_tile(hh, hw, K, S, batches)
    return stats(losses, float((pred == held_lab).mean()))


def run_hip(seed, data, arith="f32"):
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import grid_instances
    tile, lab, held, held_lab, mean, std = data
    inst = grid_instances(tile.shape[0], tile.shape[1], S, 8, B * STEPS, seed=100 + seed)
    d = DilatedNet(NET, CH, K, WD, b_max=B, s_max=S, device="cuda:0", seed=21 + seed, arith=arith)
    pool = P.TilePool([tile], [lab], "cuda:0")
    losses = []
    for i in range(STEPS):
        P.crop_to_net(d, pool, inst[i * B:(i + 1) * B], S, mean, std)
        losses.append(d.loss_value(d.train_step(B, S, LR)["loss_parts"]))
    hpool = P.TilePool([held], [held_lab], "cuda:0")
    pred, _ = loops.predict_tile(d, hpool, 0, S, B, mean, std)
    return stats(losses, float((pred.cpu().numpy() == held_lab).mean()))


def report(rows):
    """rows: seed -> dict(side -> stats[11])"""
    seeds = sorted(s for s, r in rows.items() if all(k in r for k in ("fp64", "torch", "hip")))
    print("seeds with all three sides: %d (%s..%s)" % (len(seeds), seeds[0] if seeds else "-", seeds[-1] if seeds else "-"))
    if not seeds:
        return
    a = {k: np.asarray([rows[s][k] for s in seeds]) for k in ("fp64", "torch", "hip")}
    n = len(seeds)
    names = (("loss, steps 20-59", 8), ("late loss (steps 100-119)", 9), ("held-out accuracy", 10))
    print("population means: " + ";  ".join("%s fp64 %.4f torch %.4f hip %.4f" % (nm, a["fp64"][:, c].mean(), a["torch"][:, c].mean(), a["hip"][:, c].mean())
                                             for nm, c in names))
    for x, y in (("hip", "fp64"), ("torch", "fp64"), ("hip", "torch")):
        for nm, c in names:
            d = a[x][:, c] - a[y][:, c]
            se = d.std(ddof=1) / np.sqrt(n)
            print("%-26s %5s - %-5s  mean %+.5f  standard error %.5f  (%+.2f SE)  %d of %d positive" % (nm, x, y, d.mean(), se, d.mean() / se, int((d > 0).sum()), n))
    # the start of the trajectory (before rounding differences are amplified): relative loss difference per step
    for x in ("hip", "torch"):
        rel = np.abs(a[x][:, :8] / a["fp64"][:, :8] - 1.0)
        print("|loss %5s / loss fp64 - 1| over steps 0..7, median over seeds: %s   max over seeds: %s"
              % (x, np.array2string(np.median(rel, axis=0), precision=2), np.array2string(rel.max(axis=0), precision=2)))
    # is the HIP path further from fp64 than a second fp32 implementation is?  spread of the paired differences
    for nm, c in names:
        dh, dt = a["hip"][:, c] - a["fp64"][:, c], a["torch"][:, c] - a["fp64"][:, c]
        print("%-26s spread of (x - fp64) over seeds: hip %.4f  torch %.4f" % (nm, dh.std(ddof=1), dt.std(ddof=1)))


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "merge":
        rows = {}
        for f in argv[1:]:
            z = np.load(f)
            if "seeds" in z.files:           # the compact form committed under profiles/: seeds[n], <side>[n, 11]
                for side in ("hip", "torch", "fp64"):          # (hip_bf16x6, the opt-in split-kernel arm, is in the file for the record only)
                    for sd, row in zip(z["seeds"], z[side]):
                        rows.setdefault(int(sd), {})[side] = row
                continue
            for key in z.files:
                side, seed = key.rsplit("_", 1)
                rows.setdefault(int(seed), {})[side] = z[key]
        report(rows)
        return
    kw = dict(a.split("=") for a in argv)
    side, first, n, out = kw["side"], int(kw.get("first", 1000)), int(kw.get("seeds", 8)), kw["out"]
    import time
    import torch
    data = tiles()
    res = {}
    if os.path.exists(out):
        z = np.load(out)
        res = {k: z[k] for k in z.files}
    t0 = time.time()
    if side == "cpu":
        torch.set_num_threads(int(kw.get("threads", os.cpu_count() or 1)))
    order = range(first, first + n) if kw.get("order", "asc") == "asc" else range(first + n - 1, first - 1, -1)
    done = 0
    for seed in order:
        if side == "hip":
            key = ("hip" if kw.get("arith", "f32") == "f32" else kw["arith"]) + "_%d" % seed      # (arith=bf16x6: the opt-in three-term split kernels, op-level path)
            if key in res:
                continue
            res[key] = run_hip(seed, data, kw.get("arith", "f32"))
        else:
            if "fp64_%d" % seed in res and "torch_%d" % seed in res:
                continue
            res["torch_%d" % seed] = run_cpu(seed, torch.float32, data)
            res["fp64_%d" % seed] = run_cpu(seed, torch.float64, data)
        done += 1
        if done % (10 if side == "hip" else 2) == 0 or seed == order[-1]:
            np.savez(out + ".tmp.npz", **res)
            os.replace(out + ".tmp.npz", out)
            print("%s: seed %d done (%d this run), %.0f s" % (side, seed, done, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
