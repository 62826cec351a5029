"""-m gpu: BASELINE.json's closing requirement -- "pixel-accuracy parity on a held-out synthetic tile".
The same short training run (same initial weights, same patches, same schedule) on the HIP path and on the CPU oracle
(oracle/torch_ref.py, fp32), then both models label a held-out tile by sliding window.  The two trajectories are not
bit-identical (ReLU / pool decisions flip on near-ties), so the comparison is on what the north star
names: the loss curve and the held-out pixel accuracy.  300 SGD steps are chaotic: implementations that differ only in the
ORDER of an fp32 sum end up as measurably different models (observed held-out accuracies over the revisions of the classifier
kernel and the three convolution arithmetics: 0.848 ... 0.903, CPU oracle 0.885), so the accuracy band is 0.05 while the loss
curves are held to 5 %."""
import numpy as np
import pytest
import torch

from oracle import host_ref as H
from oracle import tf_ops as T
from oracle.torch_ref import TorchNet

pytestmark = pytest.mark.gpu

from gpu_util import DEV   # noqa: E402


def test_heldout_pixel_accuracy_matches_cpu_oracle():
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import make_tile, grid_instances
    net_type, ch, K, B, S, steps, lr, wd = "dilated_icpr_rate6_small", 5, 6, 8, 24, 300, 0.01, 0.0005
    tile, lab = make_tile(160, 160, ch, K, seed=3, n_seeds=24, class_signal=0.35)
    held, held_lab = make_tile(96, 96, ch, K, seed=4, n_seeds=12, class_signal=0.35)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)), tile[:, :, :3].std(axis=(0, 1))
    inst = grid_instances(160, 160, S, 8, B * steps, seed=1)
    # the exact-fp32 arithmetic and the two split-bf16 arithmetics of the convolutions, all from the same initial weights
    ariths = ["f32", "bf16x3", "bf16x6"]
    nets = [DilatedNet(net_type, ch, K, wd, b_max=B, s_max=S, device=DEV, seed=21, arith=a) for a in ariths]
    d = nets[0]
    o = T.OracleNet(net_type, ch, K, dtype=np.float32, seed=0)
    o.p = {n: d.get_variable(n) for n in d.variable_names()}
    t = TorchNet(net_type, ch, K, params=o.p, dtype=torch.float32)
    pool = P.TilePool([tile], [lab], DEV)
    loss_d, loss_t = [], []
    loss_x = {a: [] for a in ariths[1:]}
    for i in range(steps):
        rows = inst[i * B:(i + 1) * B]
        P.crop_to_net(d, pool, rows, S, mean, std)
        out = d.train_step(B, S, lr)
        loss_d.append(d.loss_value(out["loss_parts"]))
        for a, dn in zip(ariths[1:], nets[1:]):
            P.crop_to_net(dn, pool, rows, S, mean, std)
            loss_x[a].append(dn.loss_value(dn.train_step(B, S, lr)["loss_parts"]))
        x, y, _ = H.dynamically_create_patches([tile], [lab], rows, S, is_train=False)
        x = x.copy()
        H.normalize_images(x, list(mean) + [0, 0], list(std) + [1, 1])
        lt, _ = t.train_step(x.astype(np.float32), y, lr, wd)
        loss_t.append(lt)
    print("loss HIP  ", np.round(loss_d[::30], 4))
    print("loss torch", np.round(loss_t[::30], 4))
    for a in loss_x:
        print("loss %-6s" % a, np.round(loss_x[a][::30], 4))
        assert abs(loss_x[a][0] - loss_t[0]) < 1e-4 * loss_t[0]
        assert abs(np.mean(loss_x[a][-5:]) - np.mean(loss_t[-5:])) < 0.05 * np.mean(loss_t[-5:])
    assert np.mean(loss_d[-5:]) < 0.7 * loss_d[0]                          # it learns
    assert abs(loss_d[0] - loss_t[0]) < 1e-4 * loss_t[0]                   # identical start
    assert abs(np.mean(loss_d[-5:]) - np.mean(loss_t[-5:])) < 0.05 * np.mean(loss_t[-5:])
    # held-out tile, sliding window at stride S/2 (isprs:1241-1284) on both
    hpool = P.TilePool([held], [held_lab], DEV)
    pred_d, _ = loops.predict_tile(d, hpool, 0, S, B, mean, std)
    st = H.stride_for(S)
    nh, nw = H.window_counts(96, 96, S, st)
    batches = []
    for i in range(-(-nh * nw // B)):
        p, _, pos = H.create_patches_per_map(held, held_lab, S, st, i, B)
        p = p.copy()
        H.normalize_images(p, list(mean) + [0, 0], list(std) + [1, 1])
        batches.append((t.forward(p.astype(np.float32), False).detach().numpy(), pos))
    _, _, pred_t = H.stitch_tile(96, 96, K, S, batches)
    acc_d = float((pred_d.cpu().numpy() == held_lab).mean())
    acc_t = float((pred_t == held_lab).mean())
    print("held-out pixel accuracy: HIP %.4f  CPU oracle %.4f  (chance %.3f)" % (acc_d, acc_t, 1.0 / K))
    assert acc_t > 2.0 / K and abs(acc_d - acc_t) < 0.05
    for a, dn in zip(ariths[1:], nets[1:]):
        pred_x, _ = loops.predict_tile(dn, hpool, 0, S, B, mean, std)
        acc_x = float((pred_x.cpu().numpy() == held_lab).mean())
        print("held-out pixel accuracy: HIP %s %.4f" % (a, acc_x))
        assert abs(acc_x - acc_t) < 0.05
