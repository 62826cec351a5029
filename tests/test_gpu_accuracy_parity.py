"""-m gpu: BASELINE.json's closing requirement -- "pixel-accuracy parity on a held-out synthetic tile" -- on the net it names,
dilated_grsl_rate8 (Dilated8Pooling, 5 bands, 6 classes).

The same short training run (same initial weights, same patches, same schedule) on the HIP path and on the CPU oracle
(oracle/torch_ref.py, fp32 PyTorch-CPU), then both models label a held-out tile by sliding window (isprs:1241-1284).
SGD through eight batch-normalised layers is chaotic: two correct fp32 implementations that differ only in the ORDER of a sum
drift apart after a few dozen steps, so one pair of runs says little.  The test therefore
  * holds the START of the trajectory tightly (the first steps, before rounding differences are amplified), and
  * repeats the run over several seeds (weights, batch order) on both sides and compares the two populations PAIRED by seed (the two
    sides of a seed share initial weights and patches, so most of the seed-to-seed spread -- +-0.12 of accuracy at this length of
    run, where the moving statistics of decay 0.999 are a tenth of the way in -- is common to both and cancels in the difference):
    the mean per-seed difference of the held-out accuracies (and of the late losses) must lie inside a band set by the measured
    spread of those differences.  (r05: twelve seeds, paired -- the oracle's PyTorch-CPU side now runs with a thread per GRANTED
    core, tests/conftest.py, 4 s a seed instead of 60; the unpaired band of three seeds was 0.30 of accuracy wide -- a build that
    labelled at chance would have passed it -- this one is ~0.12.  r06: the band is 3.3 standard errors of the spread measured over 1100
    paired seeds, no floors; tests/fuzz/parity_threeway.py: HIP, PyTorch-CPU fp32 and an fp64 run of the same seeds,
    profiles/r06/accuracy_parity_threeway.txt.)
"""
import numpy as np
import pytest
import torch

from oracle import host_ref as H
from oracle import tf_ops as T
from oracle.torch_ref import TorchNet

pytestmark = pytest.mark.gpu

from gpu_util import DEV   # noqa: E402

NET, CH, K, B, S, STEPS, LR, WD = "dilated_grsl_rate8", 5, 6, 6, 20, 120, 0.01, 0.0005
SEEDS = tuple(range(12))
# per-seed standard deviation of (HIP - PyTorch-CPU fp32) over 1100 paired seeds of this very run (tests/fuzz/parity_threeway.py,
# profiles/r06/accuracy_parity_threeway.txt): mean loss of steps 100-119, held-out pixel accuracy
POP_SD_LATE_LOSS, POP_SD_ACCURACY = 0.074, 0.123


def _run(seed, tile, lab, held, held_lab, mean, std):
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import grid_instances
    inst = grid_instances(tile.shape[0], tile.shape[1], S, 8, B * STEPS, seed=100 + seed)
    d = DilatedNet(NET, CH, K, WD, b_max=B, s_max=S, device=DEV, seed=21 + seed)
    o = T.OracleNet(NET, CH, K, dtype=np.float32, seed=0)
    o.p = {n: d.get_variable(n) for n in d.variable_names()}
    t = TorchNet(NET, CH, K, params=o.p, dtype=torch.float32)
    pool = P.TilePool([tile], [lab], DEV)
    loss_d, loss_t = [], []
    m5, s5 = list(mean) + [0, 0], list(std) + [1, 1]
    for i in range(STEPS):
        rows = inst[i * B:(i + 1) * B]
        P.crop_to_net(d, pool, rows, S, mean, std)
        loss_d.append(d.loss_value(d.train_step(B, S, LR)["loss_parts"]))
        x, y, _ = H.dynamically_create_patches([tile], [lab], rows, S, is_train=False)
        x = x.copy()
        H.normalize_images(x, m5, s5)
        loss_t.append(t.train_step(x.astype(np.float32), y, LR, WD)[0])
    hpool = P.TilePool([held], [held_lab], DEV)
    pred_d, _ = loops.predict_tile(d, hpool, 0, S, B, mean, std)
    st = H.stride_for(S)
    hh, hw = held_lab.shape
    nh, nw = H.window_counts(hh, hw, S, st)
    batches = []
    for i in range(-(-nh * nw // B)):
        p, _, pos = H.create_patches_per_map(held, held_lab, S, st, i, B)
        p = p.copy()
        H.normalize_images(p, m5, s5)
        batches.append((t.forward(p.astype(np.float32), False).detach().numpy(), pos))
    _, _, pred_t = H.stitch_tile(hh, hw, K, S, batches)
    return (np.asarray(loss_d), np.asarray(loss_t), float((pred_d.cpu().numpy() == held_lab).mean()), float((pred_t == held_lab).mean()))


def test_heldout_pixel_accuracy_matches_cpu_oracle_over_seeds():
    from drs_amd.synthetic import make_tile
    tile, lab = make_tile(160, 160, CH, K, seed=3, n_seeds=24, class_signal=0.6)
    held, held_lab = make_tile(96, 96, CH, K, seed=4, n_seeds=12, class_signal=0.6)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)), tile[:, :, :3].std(axis=(0, 1))
    acc_d, acc_t, late_d, late_t = [], [], [], []
    for seed in SEEDS:
        ld, lt, ad, at = _run(seed, tile, lab, held, held_lab, mean, std)
        print("seed %d  loss HIP   %s" % (seed, np.round(ld[::10], 4)))
        print("seed %d  loss torch %s" % (seed, np.round(lt[::10], 4)))
        print("seed %d  held-out pixel accuracy: HIP %.4f  CPU oracle %.4f" % (seed, ad, at))
        # the start of the trajectory, before the chaos: identical first loss, the next steps close
        dev = np.abs(ld[:8] / lt[:8] - 1.0)
        print("seed %d  |loss HIP / loss torch - 1| over the first steps: %s" % (seed, np.array2string(dev, precision=5)))
        assert dev[0] < 1e-4 and dev[1:4].max() < 1e-2, dev
        assert np.mean(ld[-20:]) < 0.8 * ld[0] and np.mean(lt[-20:]) < 0.8 * lt[0]          # both learn (the last 20 of 120 noisy steps: 0.45-0.65 of the first loss over the seeds)
        acc_d.append(ad); acc_t.append(at)
        late_d.append(np.mean(ld[-20:])); late_t.append(np.mean(lt[-20:]))
    n = len(SEEDS)
    acc_d, acc_t, late_d, late_t = map(np.asarray, (acc_d, acc_t, late_d, late_t))
    d_acc, d_loss = acc_d - acc_t, late_d - late_t                       # per-seed differences (same weights, same patches on both sides)
    se_acc = d_acc.std(ddof=1) / np.sqrt(n)                              # standard error of the mean paired difference
    se_loss = d_loss.std(ddof=1) / np.sqrt(n)
    print("held-out accuracy  HIP %.4f +- %.4f   CPU oracle %.4f +- %.4f   (chance %.3f); paired differences %s: mean %.4f, standard error %.4f"
          % (acc_d.mean(), acc_d.std(ddof=1), acc_t.mean(), acc_t.std(ddof=1), 1.0 / K, np.round(d_acc, 4), d_acc.mean(), se_acc))
    print("late loss          HIP %.4f +- %.4f   CPU oracle %.4f +- %.4f   paired differences %s: mean %.4f, standard error %.4f"
          % (late_d.mean(), late_d.std(ddof=1), late_t.mean(), late_t.std(ddof=1), np.round(d_loss, 4), d_loss.mean(), se_loss))
    assert acc_t.mean() > 2.0 / K and acc_d.mean() > 2.0 / K
    # The two populations agree: the mean paired difference lies inside the band that n seeds justify -- 3.3 standard errors, the standard
    # error from the spread of the per-seed differences MEASURED over 1100 seeds (profiles/r06/accuracy_parity_threeway.txt: HIP - PyTorch-CPU
    # fp32, late loss 0.074, held-out accuracy 0.123 per seed; the two fp32 implementations are each as far from an fp64 run of the same
    # seeds; their mean difference there: late loss +0.0018 +- 0.0022, accuracy -0.0066 +- 0.0037).  No floors (round 5 added 0.02 / 3 % to a band
    # built on the 12-seed sample's own spread, which is only known to +-20 %): the population's spread is known, and twelve seeds of
    # it are what the band is made of.  The sample's spread must itself look like the population's.
    band_loss, band_acc = 3.3 * POP_SD_LATE_LOSS / np.sqrt(n), 3.3 * POP_SD_ACCURACY / np.sqrt(n)
    print("bands: late loss +-%.4f, accuracy +-%.4f" % (band_loss, band_acc))
    assert abs(d_loss.mean()) <= band_loss, (d_loss.mean(), band_loss)
    assert abs(d_acc.mean()) <= band_acc, (d_acc.mean(), band_acc)
    assert 0.3 * POP_SD_LATE_LOSS < d_loss.std(ddof=1) < 2.2 * POP_SD_LATE_LOSS          # (1 % .. 99 % of 12-seed samples: 0.51 .. 1.58 of it)
    assert 0.3 * POP_SD_ACCURACY < d_acc.std(ddof=1) < 2.2 * POP_SD_ACCURACY


@pytest.mark.parametrize("arith", ["f32", "bf16x6"])
def test_trained_weights_label_the_heldout_tile_alike(arith):
    """The tight half of the accuracy parity (on the default exact-fp32 kernels and on the opt-in three-term bf16 arithmetic): train on the HIP path until the net has learnt the synthetic classes, then label the
    held-out tile by sliding window (isprs:1241-1284) with the HIP path and with the CPU oracle FROM THE SAME trained variables
    (moving statistics included).  No chaos between the two here, so the label maps and the pixel accuracies must agree closely."""
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import grid_instances, make_tile
    B2, S2, steps = 8, 32, 260
    tile, lab = make_tile(192, 192, CH, K, seed=3, n_seeds=24, class_signal=0.6)
    held, held_lab = make_tile(96, 96, CH, K, seed=4, n_seeds=12, class_signal=0.6)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)), tile[:, :, :3].std(axis=(0, 1))
    inst = grid_instances(192, 192, S2, 8, B2 * steps, seed=7)
    d = DilatedNet(NET, CH, K, WD, b_max=B2, s_max=S2, device=DEV, seed=33, arith=arith)
    pool = P.TilePool([tile], [lab], DEV)
    losses = []
    for i in range(steps):
        P.crop_to_net(d, pool, inst[i * B2:(i + 1) * B2], S2, mean, std)
        out = d.train_step(B2, S2, LR)
        if i % 20 == 0 or i == steps - 1:
            losses.append(d.loss_value(out["loss_parts"]))
    print("HIP training loss every 20 steps:", np.round(losses, 3))
    assert losses[-1] < 0.5 * losses[0]
    t = TorchNet(NET, CH, K, params={n: d.get_variable(n) for n in d.variable_names()}, dtype=torch.float32)
    hpool = P.TilePool([held], [held_lab], DEV)
    pred_d, _ = loops.predict_tile(d, hpool, 0, S2, B2, mean, std)
    prob_d, occ_d, _ = loops.predict_tile(d, hpool, 0, S2, B2, mean, std, return_sums=True)
    st = H.stride_for(S2)
    nh, nw = H.window_counts(96, 96, S2, st)
    m5, s5 = list(mean) + [0, 0], list(std) + [1, 1]
    batches = []
    for i in range(-(-nh * nw // B2)):
        p, _, pos = H.create_patches_per_map(held, held_lab, S2, st, i, B2)
        p = p.copy()
        H.normalize_images(p, m5, s5)
        batches.append((t.forward(p.astype(np.float32), False).detach().numpy(), pos))
    prob_t, occ_t, pred_t = H.stitch_tile(96, 96, K, S2, batches)
    got = pred_d.cpu().numpy()
    acc_d, acc_t = float((got == held_lab).mean()), float((pred_t == held_lab).mean())
    agree = float((got == pred_t).mean())
    avg_d = (prob_d.view(96, 96, K) / occ_d.view(96, 96, 1).float()).cpu().numpy()
    avg_t = prob_t / occ_t
    err = float(np.abs(avg_d - avg_t).max() / np.abs(avg_t).max())
    print(arith, "held-out tile, same trained variables: pixel accuracy HIP %.4f  CPU oracle %.4f  (chance %.3f); label maps agree on %.4f of the "
          "pixels; averaged logits differ by %.2e of their range" % (acc_d, acc_t, 1.0 / K, agree, err))
    assert acc_t > 3.0 / K                                  # the net has learnt the task
    assert err < 1e-3                                       # north star: logits within 1e-3 relative
    assert agree > 0.998 and abs(acc_d - acc_t) < 0.002
