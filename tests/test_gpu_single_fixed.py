"""-m gpu: `single_fixed` through the step loops and the command lines (BASELINE configs[0] and configs[1]).

The reference's `single_fixed` branch draws no size (`cur_patch_size = int(values[0])`, isprs:1735-1736), keeps no score arrays
(isprs:2054-2064 builds them for the other three distribution types only), writes a checkpoint without the three `.npy` side files
(isprs:1798-1802 saves them inside the sized branch), and validates / tests at `values[0]` (isprs:1805-1811).  These tests drive
exactly that branch of loops.train / loops_indexed.train / cli.main* on the HIP path and hold its first step to the oracle."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

from gpu_util import DEV   # noqa: E402


class _Spy(object):
    """wraps EngineNet.train_step: keeps, for every step, what the step consumed (input slab, labels, masks, variables before) and
    what it decided (ReLU signs, pool winners) and returned (loss) -- the material for a decision-aligned oracle step"""

    def __init__(self, monkeypatch, keep=1, decisions=True):
        from drs_amd.engine import EngineNet
        self.steps, self.keep = [], keep
        real = EngineNet.train_step
        spy = self

        def train_step(net, B, S, lr0, **kw):
            rec = None
            if len(spy.steps) < spy.keep:
                rec = dict(B=B, S=S, lr=lr0, kw=dict(kw), x0=net.x0.clone(), labels=net.labels.clone(), acc_mask=net.acc_mask.clone(),
                           loss_mask=net.loss_mask.clone(), params=net.params.clone(), mom=net.mom.clone(), bn=net.bn.clone(), gs=net.global_step)
            out = real(net, B, S, lr0, **kw)
            if rec is not None:
                torch.cuda.synchronize()
                rec["loss"] = net.loss_value(out["loss_parts"])
                rec["conf"] = out["conf"].cpu().numpy().copy()
                M = B * S * S
                rec["dec"] = []
                for i, L in enumerate(net.plan.layers if decisions else []):      # (gigabytes per step at batch 128: only where they are used)
                    z = net.z[i][:M * L.cout].cpu().numpy().reshape(B, S, S, L.cout)
                    mr = net.mean_rstd[i].cpu().numpy().reshape(L.cout, 2)
                    d = {"pos": (z - mr[:, 0]) * mr[:, 1] > 0}
                    if net._is_max(i):
                        d["idx"] = net.idx[i][:M * L.cout].cpu().numpy().reshape(B, S, S, L.cout)
                    rec["dec"].append(d)
                rec["net"] = net
                spy.steps.append(rec)
            return out
        monkeypatch.setattr(EngineNet, "train_step", train_step)


def _patches_of(rec, channels):
    """the [B, S, S, C] patches and [B, S, S] labels a recorded step consumed, read back from conv1's zero-haloed input slab"""
    net, B, S = rec["net"], rec["B"], rec["S"]
    C0, P0 = net.plan.buffers["x0"]
    x = rec["x0"][:B * (S + 2 * P0) * (S + 2 * P0) * C0].cpu().numpy().reshape(B, S + 2 * P0, S + 2 * P0, C0)
    assert np.all(x[:, :P0] == 0) and np.all(x[:, :, :P0] == 0) and np.all(x[..., channels:] == 0)      # halo and padded bands are zero
    return x[:, P0:P0 + S, P0:P0 + S, :channels].astype(np.float64), rec["labels"][:B * S * S].cpu().numpy().reshape(B, S, S).astype(np.int64)


def _oracle_first_step(rec, net_type, channels, K, wd):
    net = rec["net"]
    o = T.OracleNet(net_type, channels, K, dtype=np.float64, seed=0)
    flat, bn = rec["params"].cpu().numpy(), rec["bn"].cpu().numpy()
    for name, (off, shape) in net.plan.offsets.items():
        o.p[name] = flat[off:off + int(np.prod(shape))].reshape(shape).astype(np.float64)
    for L in net.plan.layers:
        b0 = net.plan.bn_offsets[L.name]
        o.p[L.name + "/moving_mean"] = bn[b0:b0 + L.cout].astype(np.float64)
        o.p[L.name + "/moving_variance"] = bn[b0 + L.cout:b0 + 2 * L.cout].astype(np.float64)
    x, y = _patches_of(rec, channels)
    loss, _ = o.train_step(x, y, rec["lr"], wd, decisions=rec["dec"])
    return loss, o


def _torch_step_loss(rec, net_type, channels, K, wd):
    """the loss of a recorded step from the CPU oracle's fp64 PyTorch restatement (oracle/torch_ref.py) on the very patches, labels and
    variables the step consumed: forward pass in training mode + loss_def, free-running (its own ReLU signs and pool winners: near-tie
    flips move a mean over >= 10^5 pixels by far less than the 1e-4 the callers allow)"""
    from oracle.torch_ref import TorchNet
    net = rec["net"]
    x, y = _patches_of(rec, channels)
    params = {}
    flat, bn = rec["params"].cpu().numpy(), rec["bn"].cpu().numpy()
    for name, (off, shape) in net.plan.offsets.items():
        params[name] = flat[off:off + int(np.prod(shape))].reshape(shape).astype(np.float64)
    for L in net.plan.layers:
        b0 = net.plan.bn_offsets[L.name]
        params[L.name + "/moving_mean"], params[L.name + "/moving_variance"] = bn[b0:b0 + L.cout], bn[b0 + L.cout:b0 + 2 * L.cout]
    tn = TorchNet(net_type, channels, K, params=params, dtype=torch.float64)
    with torch.no_grad():
        return float(tn.loss(tn.forward(x, True), y, wd))


def _setup_isprs(h, w, bands, seed, ref_crop=25, ref_stride=10, classes=6):
    from drs_amd import sampling as SP
    from drs_amd.synthetic import make_tile
    a, b = make_tile(h, w, bands, classes, seed=seed, n_seeds=40), make_tile(h // 2, w // 2, bands, classes, seed=seed + 1, n_seeds=20)
    dist = SP.create_distributions_over_classes([a[1]], ref_crop, ref_stride, num_classes=classes)
    tdist = SP.create_distributions_over_classes([b[1]], ref_crop, ref_crop, num_classes=classes)
    rot = SP.create_rotation_distribution(dist)
    return a, b, dist, tdist, rot


def test_config1_single_fixed_through_the_training_loop(tmp_path, capsys, monkeypatch):
    """BASELINE configs[0] at its own shape: dilated_icpr_original (Dilated6), single_fixed 25 x 25, 3-band 256 x 256 tile, batch 16."""
    from drs_amd import loops
    from drs_amd.cli import init_size_scores
    from drs_amd.net import DilatedNet
    NET, CH, K, B, S, WD, LR = "dilated_icpr_original", 3, 6, 16, 25, 0.005, 0.01
    random.seed(11)
    np.random.seed(11)
    a, b, dist, tdist, rot = _setup_isprs(256, 256, CH, 51)
    acc, occ, chosen, probs = init_size_scores("single_fixed", [S])
    assert acc is None and occ is None and chosen is None and probs is None            # isprs:2054-2064: no score arrays for single_fixed
    monkeypatch.setattr(loops, "VAL_INTERVAL", 6)
    val_sizes = []
    real_val = loops.validation
    monkeypatch.setattr(loops, "validation", lambda net, pool, inst, m, s, bs, step, crop, comm=None: (val_sizes.append((step, crop)), real_val(net, pool, inst, m, s, bs, step, crop, comm))[1])
    spy = _Spy(monkeypatch, keep=1)
    out = str(tmp_path) + "/"
    mean, std = [0.45, 0.5, 0.4], [0.2, 0.25, 0.2]
    args = ([a[0]], [a[1]], dist, rot, [b[0]], [b[1]], tdist, ["b"], LR, B, 12, WD, mean, std, "acc", "single_fixed", [S], acc, occ, chosen,
            probs, 20, out, 4, NET, "vaihingen")
    net = loops.train(*args, "none", device=DEV, val_cache_dir=str(tmp_path))
    text = capsys.readouterr().out
    # -- the branch itself
    assert net.global_step == 12 and net.s_max == S and net.b_max == B
    assert [t for t in text.split("\n") if t.strip().isdigit()] == ["25"] * 12            # the reference prints the size every step
    assert val_sizes == [(6, S), (12, S), (12, S)]                                       # validation at values[0]: at the interval and at the end
    files = sorted(os.listdir(out))
    assert "model-6.npz" in files and "model-12.npz" in files
    assert not [f for f in files if f.startswith("patch_")]                              # no score side files (isprs:1798-1802)
    assert "Current patch size" not in text                                              # select_best_patch_size never ran
    assert text.count("Training Minibatch: Loss=") == 3 and text.count("Validation: Overall Accuracy=") == 3
    # -- the first step against the oracle on the same 16 patches, under the device's own discrete decisions
    rec = spy.steps[0]
    assert (rec["B"], rec["S"], rec["gs"]) == (B, S, 0)
    loss_ref, _ = _oracle_first_step(rec, NET, CH, K, WD)
    assert abs(rec["loss"] - loss_ref) < 1e-4 * abs(loss_ref), (rec["loss"], loss_ref)
    assert int(rec["conf"].sum()) == int(rec["acc_mask"][:B * S * S].sum().item())       # masked count (rotated-in corners excluded, isprs:510-531)

    # -- checkpoint -> resume reproduces the next step bitwise: continue in memory from `net` and, separately, resume from
    # model-12 through the loop (former_model_path contains 'model': isprs:1708-1715); both must take the same step on the same batch
    spy2 = _Spy(monkeypatch, keep=1)
    random.seed(12)
    np.random.seed(12)
    net_r = loops.train(*args[:10], 12, *args[11:], out + "model-12", device=DEV, val_cache_dir=str(tmp_path))
    assert net_r.global_step == 13                                                       # resumes AT step 12: range(12, 13) is one step
    r2 = spy2.steps[0]
    assert r2["gs"] == 12 and torch.equal(r2["params"], net.params) and torch.equal(r2["mom"], net.mom) and torch.equal(r2["bn"], net.bn)
    cont = DilatedNet(NET, CH, K, WD, b_max=B, s_max=S, device=DEV, seed=999)
    cont.load_state_dict(net.state_dict())
    assert cont.global_step == 12
    for name in ("x0", "labels", "acc_mask", "loss_mask"):
        getattr(cont, name).copy_(r2[name])
    from drs_amd.engine import EngineNet
    monkeypatch.undo()                                                                   # the plain train_step again
    cont.train_step(B, S, LR)
    torch.cuda.synchronize()
    assert isinstance(cont, EngineNet)
    assert torch.equal(cont.params, net_r.params) and torch.equal(cont.mom, net_r.mom) and torch.equal(cont.bn, net_r.bn)
    assert not [f for f in os.listdir(out) if f.startswith("patch_")]


def test_config2_single_fixed_64_through_the_training_loop(tmp_path, capsys, monkeypatch):
    """BASELINE configs[1] as a LOOP: dilated_grsl (Dilated6Pooling), single_fixed 64 x 64, 5 bands, batch 64, 2048 x 2048 tile."""
    from drs_amd import loops, sampling as SP
    from drs_amd.synthetic import make_tile
    NET, CH, K, B, S, WD = "dilated_grsl", 5, 6, 64, 64, 0.005
    random.seed(13)
    np.random.seed(13)
    tile, lab = make_tile(2048, 2048, CH, K, seed=1234)
    vt, vl = make_tile(256, 256, CH, K, seed=77, n_seeds=30)
    dist = SP.create_distributions_over_classes([lab], 64, 64)
    tdist = SP.create_distributions_over_classes([vl], 64, 64)
    rot = SP.create_rotation_distribution(dist)
    spy = _Spy(monkeypatch, keep=1)
    os.makedirs(str(tmp_path / "out"))
    out = str(tmp_path / "out") + "/"
    mean, std = tile[:, :, :3].mean(axis=(0, 1)).tolist() + [0, 0], tile[:, :, :3].std(axis=(0, 1)).tolist() + [1, 1]
    net = loops.train([tile], [lab], dist, rot, [vt], [vl], tdist, ["v"], 0.01, B, 3, WD, mean, std, "acc", "single_fixed", [S], None, None,
                      None, None, 20, out, 1, NET, "vaihingen", "none", device=DEV, val_cache_dir=str(tmp_path))
    text = capsys.readouterr().out
    assert net.global_step == 3 and net.s_max == S and len(net.plan.layers) == 6
    assert sorted(f for f in os.listdir(out)) == ["model-3.npz"]
    losses = [float(t.split("Loss= ")[1].split()[0]) for t in text.split("\n") if "Training Minibatch" in t]
    assert len(losses) == 3 and all(np.isfinite(losses))
    rec = spy.steps[0]
    assert (rec["B"], rec["S"]) == (B, S)
    # the first step's loss against the CPU oracle on the same 64 patches: the fp64 PyTorch-CPU restatement (oracle/torch_ref.py),
    # forward pass in training mode + loss_def -- free-running (its own ReLU signs and pool winners; near-tie flips move a mean over
    # 262 144 pixels by far less than the bound)
    x, y = _patches_of(rec, CH)
    assert x.shape == (B, S, S, CH) and np.isfinite(x).all() and y.min() >= 0 and y.max() < K
    loss_ref = _torch_step_loss(rec, NET, CH, K, WD)
    assert abs(rec["loss"] - loss_ref) < 1e-4 * abs(loss_ref), (rec["loss"], loss_ref)
    assert abs(losses[0] - loss_ref) < 1e-5 * abs(loss_ref) + 1e-6                           # what the loop printed IS that step's loss
    # bands 0..2 normalised, 3..4 not (isprs:74-81); the zero corners a rotation brings in are normalised with the rest, as in the reference
    inside = rec["acc_mask"][:B * S * S].cpu().numpy().reshape(B, S, S).astype(bool)
    assert abs(float(x[..., :3][inside].mean())) < 0.5 and abs(float(x[..., 3][inside].mean()) - float(tile[..., 3].mean())) < 0.1
    assert int(rec["conf"].sum()) == int(rec["acc_mask"][:B * S * S].sum().item())
    assert "Validation: Overall Accuracy=" in text and "Current patch size" not in text


def test_single_fixed_through_the_isprs_command_line(tmp_path, monkeypatch, capsys):
    """process = training -> validate_test -> generate_final_maps with distribution_type single_fixed (isprs:1987-2138): no score
    files are written and none are read; the test / final-map window is values[0]"""
    from drs_amd import cli, loops
    monkeypatch.chdir(tmp_path)
    out = str(tmp_path) + "/sf_"
    common = ["isprs_dilated_random.py", "synthetic:96x110x3/vaihingen/", out]
    tail = ["a", "c", "0.01", "0.005", "16", "4", "25", "10", "dilated_icpr_original", "single_fixed", "25", "acc"]
    sizes = []
    real = loops.predict_tile
    monkeypatch.setattr(loops, "predict_tile", lambda net, pool, k, crop, *a, **kw: (sizes.append(crop), real(net, pool, k, crop, *a, **kw))[1])
    random.seed(2)
    np.random.seed(2)
    net = cli.main(common + ["none"] + tail + ["training"], device=DEV)
    assert net.global_step == 4 and net.s_max == 25 and net.plan.channels == 3
    assert os.path.isfile(out + "model-4.npz") and not [f for f in os.listdir(str(tmp_path)) if "patch_" in f]
    cm, maps = cli.main(common + [out + "model-4"] + tail + ["validate_test"], device=DEV)
    assert cm.sum() == 96 * 110 - int((maps[0] < 0).sum()) and maps[0].shape == (96, 110)
    maps2 = cli.main(common + [out + "model-4"] + tail + ["generate_final_maps"], device=DEV)
    np.testing.assert_array_equal(maps2[0], maps[0])
    assert sizes == [25, 25]
    text = capsys.readouterr().out
    assert "Test ALL MAPS" in text and "Current patch size" not in text


def test_single_fixed_through_the_coffee_and_contest_command_lines(tmp_path, capsys):
    """coffee:1106-1150 / contest:1229-1271 with single_fixed: the indexed loops keep no score arrays, the side files in either
    naming style are not written, test runs at values[0]"""
    from drs_amd import cli
    out = str(tmp_path) + "/"
    random.seed(3)
    np.random.seed(3)
    net = cli.main_coffee(["coffee_dilated_random.py", "synthetic:2x60x60x3/", "synthetic:1x60x60x3/", out, "none", "0.01", "0.001", "6", "3",
                           "25", "10", "dilated_icpr_rate6", "single_fixed", "25", "acc"], device=DEV)
    assert net.plan.K == 2 and net.global_step == 3 and net.s_max == 25
    assert sorted(os.listdir(out)) == ["model-3.npz"]
    os.remove(out + "model-3.npz")
    net = cli.main_contest(["contest_dilated_random.py", "synthetic:80x70x3/", out, "none", "0.01", "0.001", "4", "3", "25", "10", "dilated_grsl",
                            "single_fixed", "25", "acc", "train"], device=DEV)
    assert net.plan.K == 7 and net.global_step == 3 and sorted(os.listdir(out)) == ["model-3.npz"]
    cm, maps = cli.main_contest(["contest_dilated_random.py", "synthetic:80x70x3/", out, out + "model-3", "0.01", "0.001", "4", "3", "25", "10",
                                 "dilated_grsl", "single_fixed", "25", "acc", "test"], device=DEV)
    assert cm.shape == (7, 7) and maps[0].shape == (80, 70)
    text = capsys.readouterr().out
    assert "Test ALL MAPS" in text and "Current patch size" not in text
