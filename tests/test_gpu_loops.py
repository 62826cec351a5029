"""-m gpu: the step loops (train / validation / sliding-window inference) end to end on small synthetic tiles."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import host_ref as H
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

from gpu_util import DEV, rel_err   # noqa: E402


def _tiles():
    from drs_amd.synthetic import make_tile
    a = make_tile(96, 110, 5, 6, seed=1, n_seeds=30)
    b = make_tile(80, 96, 5, 6, seed=2, n_seeds=30)
    return [a[0], b[0]], [a[1], b[1]]


def test_train_loop_checkpoint_resume_and_validation(tmp_path, capsys):
    from drs_amd import loops, sampling as SP
    from drs_amd.cli import init_size_scores
    data, labels = _tiles()
    random.seed(0)
    np.random.seed(0)
    dist = SP.create_distributions_over_classes(labels, 25, 10)
    rot = SP.create_rotation_distribution(dist)
    mean, std = SP.dynamically_calculate_mean_and_std(data, dist, 25)
    values = [9, 13]
    acc, occ, chosen, probs = init_size_scores("multi_fixed", values)
    out = str(tmp_path) + "/"
    net = loops.train(data, labels, dist, rot, data, labels, dist, ["a", "b"], 0.01, 8, 6, 0.005, mean, std, "acc", "multi_fixed",
                      values, acc, occ, chosen, probs, 20, out, 2, "dilated_grsl", "vaihingen", "none", device=DEV,
                      val_cache_dir=str(tmp_path))
    text = capsys.readouterr().out
    assert "Training Minibatch: Loss=" in text and "Validation: Overall Accuracy=" in text and "Optimization Finished!" in text
    assert occ.sum() >= 6 and net.global_step == 6
    for f in ("model-6.npz", "patch_acc_loss_step_6.npy", "patch_occur_step_6.npy", "patch_chosen_values_step_6.npy"):
        assert os.path.isfile(out + f), f
    w_before = net.get_variable("conv3/weights")
    # resume: step is parsed from the '-6' suffix, state and the size scores come back (isprs:1708-1715)
    net2 = loops.train(data, labels, dist, rot, data, labels, dist, ["a", "b"], 0.01, 8, 8, 0.005, mean, std, "acc", "multi_fixed",
                       values, None, None, None, probs, 20, out, 2, "dilated_grsl", "vaihingen", out + "model-6", device=DEV,
                       val_cache_dir=str(tmp_path))
    assert net2.global_step == 6 + 3          # resumes AT step 6 (range(current_iter, niter+1)), like the reference
    assert not np.array_equal(net2.get_variable("conv3/weights"), w_before)
    assert np.isfinite(net2.get_variable("conv1/moving_variance")).all()


def test_training_reduces_loss_on_a_fixed_batch():
    from drs_amd.net import DilatedNet
    rng = np.random.default_rng(0)
    B, S, ch, K = 4, 16, 5, 6
    d = DilatedNet("dilated8_grsl", ch, K, 0.0005, b_max=B, s_max=S, device=DEV)
    x = rng.normal(size=(B, S * S * ch)).astype(np.float32)
    y = (rng.integers(0, 2, size=(B, S * S)) * 3).astype(np.int64)
    losses = []
    for i in range(25):
        d.feed(x, y, S)
        out = d.train_step(B, S, 0.05)
        losses.append(d.loss_value(out["loss_parts"]))
    assert losses[-1] < 0.5 * losses[0], losses


def test_sliding_window_tile_matches_oracle():
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    rng = np.random.default_rng(4)
    net_type, ch, K, S, bs = "dilated_grsl", 5, 6, 16, 5
    tile = rng.uniform(size=(40, 53, ch))
    lab = rng.integers(0, 7, size=(40, 53)).astype(np.uint8)          # 6 = eroded boundary
    mean, std = np.array([0.5, 0.5, 0.5, 0, 0]), np.array([0.25, 0.25, 0.25, 1, 1])
    d = DilatedNet(net_type, ch, K, 0.005, b_max=bs, s_max=S, device=DEV, seed=3)
    o = T.OracleNet(net_type, ch, K, seed=3)
    for n in d.variable_names():
        v = d.get_variable(n)
        if n.endswith("moving_mean"):
            v = (rng.normal(size=v.shape) * 0.05).astype(np.float32)
            d.set_variable(n, v)
        o.p[n] = v.astype(np.float64)
    pool = P.TilePool([tile], [lab], DEV)
    pred, total = loops.predict_tile(d, pool, 0, S, bs, mean, std)
    st = H.stride_for(S)
    nh, nw = H.window_counts(40, 53, S, st)
    assert total == nh * nw
    batches = []
    for i in range(-(-nh * nw // bs)):
        p, _, pos = H.create_patches_per_map(tile, lab, S, st, i, bs)
        p = p.copy()
        H.normalize_images(p, mean, std)
        batches.append((o.forward(p.astype(np.float32).astype(np.float64), False).astype(np.float32), pos))
    prob, occur, am = H.stitch_tile(40, 53, K, S, batches)
    avg = prob / occur
    srt = np.sort(avg, axis=2)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-3 * np.abs(avg).max()
    got = pred.cpu().numpy()
    assert clear.mean() > 0.95
    np.testing.assert_array_equal(got[clear], am[clear])
    # scores as validate_test prints them, eroded label skipped
    cm, maps = loops.validate_test(d, [tile], [lab], ["t0"], bs, mean, std, S, 0, pool=pool)
    keep = lab != 6
    want = np.zeros((K, K), dtype=np.int64)
    np.add.at(want, (lab[keep], got[keep]), 1)
    np.testing.assert_array_equal(cm, want)
    np.testing.assert_array_equal(maps[0], got)


def test_validation_confusion_is_consistent():
    from drs_amd import loops, patches as P, sampling as SP
    from drs_amd.net import DilatedNet
    data, labels = _tiles()
    random.seed(1)
    np.random.seed(1)
    dist = SP.create_distributions_over_classes(labels, 25, 10)
    inst = SP.select_super_batch_instances(dist, batch_size=4, super_batch=3)
    d = DilatedNet("dilated_icpr_original", 5, 6, 0.005, b_max=4, s_max=25, device=DEV)
    pool = P.TilePool(data, labels, DEV)
    cm, px = loops.validation(d, pool, inst, [0.5] * 5, [0.2] * 5, 4, 0, 25)
    assert cm.sum() == px == 12 * 25 * 25
    _, lab_ref, _ = H.dynamically_create_patches(data, labels, inst, 25, is_train=False)
    np.testing.assert_array_equal(cm.sum(axis=1), np.bincount(lab_ref.reshape(-1), minlength=6))


def test_config3_dense_multinomial_loss_update(tmp_path, capsys):
    """BASELINE configs[3] in miniature: DenseDilated6, multinomial size distribution, update_type=loss,
    4-band tiles, 2 classes."""
    from drs_amd import loops, sampling as SP
    from drs_amd.cli import init_size_scores
    from drs_amd.synthetic import make_tile
    a = make_tile(90, 90, 4, 2, seed=5, n_seeds=20)
    data, labels = [a[0]], [a[1]]
    random.seed(3)
    np.random.seed(3)
    dist = SP.create_distributions_over_classes(labels, 25, 10, num_classes=2)
    rot = SP.create_rotation_distribution(dist)
    values = [25, 30]
    acc, occ, chosen, probs = init_size_scores("multinomial", values)
    assert len(probs) == 6 and abs(probs.sum() - 1) < 1e-12
    out = str(tmp_path) + "/"
    net = loops.train(data, labels, dist, rot, data, labels, dist, ["a"], 0.01, 4, 5, 0.001, [0.5] * 4, [0.2] * 4, "loss",
                      "multinomial", values, acc, occ, chosen, probs, 20, out, 5, "dilated_icpr_rate6_densely", "vaihingen", "none",
                      num_classes=2, device=DEV, val_cache_dir=str(tmp_path))
    assert net.plan.dense and net.global_step == 5
    assert occ.sum() >= 5 and np.all(acc >= 0) and acc.sum() > 0          # loss * epoch/10 accumulated per drawn size
    assert "Validation: Overall Accuracy=" in capsys.readouterr().out


def test_config4_sliding_window_dilated8_at_64():
    """BASELINE configs[4] in miniature: Dilated8Pooling, 64x64 windows at stride 32 over a 5-band mosaic."""
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import make_tile
    tile, lab = make_tile(200, 260, 5, 6, seed=9, n_seeds=40)
    d = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=16, s_max=64, device=DEV)
    pool = P.TilePool([tile], [lab], DEV)
    pred, total = loops.predict_tile(d, pool, 0, 64, 16, [0.5] * 5, [0.2] * 5)
    assert total == P.window_counts(200, 260, 64, 32)[0] * P.window_counts(200, 260, 64, 32)[1] == 6 * 8
    got = pred.cpu().numpy()
    assert got.shape == (200, 260) and got.max() < 6


def test_multiscale_evaluation_matches_oracle():
    """validate_test_multiscale (isprs:1347-1474): per-size softmax maps summed."""
    from drs_amd import loops, patches as P
    from drs_amd.net import DilatedNet
    rng = np.random.default_rng(8)
    net_type, ch, K, bs = "dilated_grsl", 5, 6, 6
    tile = rng.uniform(size=(44, 50, ch))
    lab = rng.integers(0, 6, size=(44, 50)).astype(np.uint8)
    mean, std = np.array([0.5, 0.5, 0.5, 0, 0]), np.array([0.25, 0.25, 0.25, 1, 1])
    d = DilatedNet(net_type, ch, K, 0.005, b_max=bs, s_max=25, device=DEV, seed=5)
    o = T.OracleNet(net_type, ch, K, seed=5)
    for n in d.variable_names():
        o.p[n] = d.get_variable(n).astype(np.float64)
    sizes = loops.best_sizes("multi_fixed", [13, 25, 18], np.array([0.5, 0.9, 0.7], dtype=np.float32), np.array([1, 1, 1]), "acc", 2)
    assert sizes == [25, 18]
    pool = P.TilePool([tile], [lab], DEV)
    pred = loops.predict_tile_multiscale(d, pool, 0, sizes, bs, mean, std).cpu().numpy()
    maps = []
    for S in sizes:
        st = H.stride_for(S)
        nh, nw = H.window_counts(44, 50, S, st)
        batches = []
        for i in range(-(-nh * nw // bs)):
            p, _, pos = H.create_patches_per_map(tile, lab, S, st, i, bs)
            p = p.copy()
            H.normalize_images(p, mean, std)
            batches.append((o.forward(p.astype(np.float32).astype(np.float64), False).astype(np.float32), pos))
        prob, occur, _ = H.stitch_tile(44, 50, K, S, batches)
        maps.append(prob / occur.astype(float))
    want = H.multiscale_argmax(maps)
    sm = np.sum([H.softmax_lastaxis(m.astype(np.float32)) for m in maps], axis=0)
    srt = np.sort(sm, axis=2)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert clear.mean() > 0.9
    np.testing.assert_array_equal(pred[clear], want[clear])
    cm, _ = loops.validate_test(d, [tile], [lab], ["t"], bs, mean, std, None, 0, pool=pool, crop_sizes=sizes)
    assert cm.sum() == 44 * 50


def test_cli_training_then_validate_test_then_final_maps(tmp_path, monkeypatch, capsys):
    """The reference's three `process` values through the same 16 positional arguments (isprs:1987-2042)."""
    from drs_amd import cli
    monkeypatch.chdir(tmp_path)           # the reference keeps its .npy caches in the cwd (isprs:2087-2115)
    out = str(tmp_path) + "/out_"
    common = ["isprs_dilated_random.py", "synthetic:70x80x5/vaihingen/", out]
    tail = ["a,b", "c", "0.01", "0.005", "4", "3", "25", "10", "dilated8_grsl", "multi_fixed", "9,13", "acc"]
    random.seed(0)
    np.random.seed(0)
    net = cli.main(common + ["none"] + tail + ["training"], device=DEV)
    assert net.global_step == 3 and os.path.isfile(out + "model-3.npz")
    for f in ("_rotation.npy", "_mean.npy", "_std.npy"):
        assert os.path.isfile(os.path.join(str(tmp_path), "dataset_vaihingen_crop_25_stride_10" + f))
    cm, maps = cli.main(common + [out + "model-3"] + tail + ["validate_test"], device=DEV)
    assert cm.sum() > 0 and maps[0].shape == (70, 80)
    maps2 = cli.main(common + [out + "model-3"] + tail + ["generate_final_maps"], device=DEV)
    np.testing.assert_array_equal(maps2[0], maps[0])
    assert os.path.isfile(out + "top_mosaic_09cm_areac_class.npy")
    from PIL import Image
    from drs_amd.datasets import ISPRS_PALETTE
    rgb = np.asarray(Image.open(out + "top_mosaic_09cm_areac_class.tif"))          # the reference's file name and palette (isprs:1950, 118-139)
    np.testing.assert_array_equal(rgb, ISPRS_PALETTE[maps2[0]])
    text = capsys.readouterr().out
    assert "Test ALL MAPS" in text and "net_type" in text
    with pytest.raises(SystemExit):
        cli.main(["isprs_dilated_random.py", "too", "few"], device=DEV)


def test_coffee_and_contest_command_lines(tmp_path, capsys):
    """coffee_dilated_random.py:1106-1150 (14 arguments) and contest_dilated_random.py:1229-1271 (14 incl. operation)."""
    from drs_amd import cli
    out = str(tmp_path) + "/"
    random.seed(1)
    np.random.seed(1)
    net = cli.main_coffee(["coffee_dilated_random.py", "synthetic:2x60x60x3/", "synthetic:1x60x60x3/", out, "none", "0.01", "0.001", "6",
                           "4", "25", "10", "dilated_icpr_rate6_small", "multi_fixed", "9,13", "loss"], device=DEV)
    assert net.plan.K == 2 and net.global_step == 4 and net.lr_decay_factor == 0.1
    for f in ("model-4.npz", "errorAcc_step_4.npy", "errorOccur_step_4.npy", "chosenValues_step_4.npy"):     # coffee:1341-1343
        assert os.path.isfile(out + f), f
    net = cli.main_contest(["contest_dilated_random.py", "synthetic:80x70x3/", out, "none", "0.01", "0.001", "4", "3", "25", "10",
                            "dilated_grsl", "multi_fixed", "9,13", "acc", "train"], device=DEV)
    assert net.plan.K == 7 and net.global_step == 3
    assert np.load(out + "patch_occur_step_3.npy").sum() == 2 + 3                   # starts at ones (contest:1275)
    cm, maps = cli.main_contest(["contest_dilated_random.py", "synthetic:80x70x3/", out, out + "model-3", "0.01", "0.001", "4", "3", "25",
                                 "10", "dilated_grsl", "multi_fixed", "9,13", "acc", "test"], device=DEV)
    assert cm.shape == (7, 7) and maps[0].shape == (80, 70)
    assert "Test ALL MAPS" in capsys.readouterr().out


def test_void_label_is_masked_out_of_loss_and_accuracy():
    from drs_amd import patches as P
    from drs_amd.net import DilatedNet
    rng = np.random.default_rng(0)
    tile = rng.uniform(size=(40, 40, 3)).astype(np.float32)
    lab = rng.integers(0, 8, size=(40, 40)).astype(np.uint8)
    pool = P.TilePool([tile], [lab], DEV, dtype=np.float32)
    d = DilatedNet("dilated_grsl", 3, 7, 0.0, b_max=2, s_max=16, device=DEV)
    inst = np.array([[0, 0, 0], [0, 20, 24]])
    P.crop_to_net(d, pool, inst, 16, [0, 0, 0], [1, 1, 1], void_label=7)
    M = 2 * 16 * 16
    want = np.stack([lab[0:16, 0:16], lab[20:36, 24:40]]) != 7
    np.testing.assert_array_equal(d.acc_mask[:M].cpu().numpy().reshape(2, 16, 16).astype(bool), want)
    d.loss_mask[:M].copy_(d.acc_mask[:M])
    out = d.train_step(2, 16, 0.01, use_loss_mask=True, global_pixels=int(want.sum()), apply_update=False)
    assert int(out["conf"].sum().item()) == int(want.sum())
    assert np.isfinite(d.loss_value(out["loss_parts"]))


def test_tensorflow_bundle_checkpoint_roundtrip(tmp_path):
    """A net saved in the reference's checkpoint format (tf.train.Saver V2 bundle) restores to identical logits."""
    from drs_amd import loops, tf_checkpoint
    from drs_amd.net import DilatedNet
    rng = np.random.default_rng(1)
    B, S = 2, 15
    a = DilatedNet("dilated8_grsl", 5, 6, 0.005, b_max=B, s_max=S, device=DEV, seed=9)
    x = rng.normal(size=(B, S * S * 5)).astype(np.float32)
    y = rng.integers(0, 6, size=(B, S * S))
    a.feed(x, y, S)
    a.train_step(B, S, 0.01)                      # non-trivial momentum, moving statistics and step counter
    prefix = str(tmp_path / "model-1")
    tf_checkpoint.save_tf_checkpoint(a, prefix)
    assert os.path.isfile(prefix + ".index") and os.path.isfile(prefix + ".data-00000-of-00001")
    b = DilatedNet("dilated8_grsl", 5, 6, 0.005, b_max=B, s_max=S, device=DEV, seed=123)
    loops.load_checkpoint(b, prefix)              # picks the bundle because <prefix>.index exists
    assert b.global_step == 1
    assert torch.equal(a.params, b.params) and torch.equal(a.mom, b.mom) and torch.equal(a.bn, b.bn)
    a.feed(x, None, S)
    b.feed(x, None, S)
    assert torch.equal(a.forward(B, S)[1], b.forward(B, S)[1])


@pytest.mark.parametrize("script,args,expect", [
    ("isprs_dilated_random.py", ["synthetic:70x80x5/vaihingen/", "OUT", "none", "a,b", "c", "0.01", "0.005", "4", "2", "25", "10", "dilated8_grsl",
                                 "multi_fixed", "9,13", "acc", "training"], "model-2.npz"),
    ("coffee_dilated_random.py", ["synthetic:2x60x60x3/", "synthetic:1x60x60x3/", "OUT", "none", "0.01", "0.001", "6", "2", "25", "10",
                                  "dilated_icpr_rate6_small", "multi_fixed", "9,13", "loss"], "model-2.npz"),
    ("contest_dilated_random.py", ["synthetic:80x70x3/", "OUT", "none", "0.01", "0.001", "4", "2", "25", "10", "dilated_grsl", "multi_fixed", "9,13",
                                   "acc", "train"], "model-2.npz"),
])
def test_entry_scripts_run_as_processes(tmp_path, script, args, expect):
    """the three scripts at the repository root, started the way a user of the reference starts them (own process, positional
    arguments): exit status 0, the reference's progress lines, the checkpoint where the reference puts it"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path) + "/"
    argv = [out if a == "OUT" else a for a in args]
    r = subprocess.run([sys.executable, os.path.join(root, script)] + argv, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.isfile(out + expect), os.listdir(out)
    assert "Iter" in r.stdout or "iter" in r.stdout or "Step" in r.stdout, r.stdout[-1000:]
    r = subprocess.run([sys.executable, os.path.join(root, script), "too", "few"], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
