"""Pin oracle/host_ref.py against fixtures produced by the reference's own helpers
(tests/golden/make_goldens.py).  CPU only."""
import os
import random

import numpy as np

from oracle import host_ref as H


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_multinomial_probs(golden_dir):
    g = _g(golden_dir, "multinomial_probs.npz")
    p = H.define_multinomial_probs(list(g["values"]))
    assert len(p) == 76
    np.testing.assert_array_equal(p, g["probs"])
    np.testing.assert_array_equal(H.define_multinomial_probs(list(g["values2"])), g["probs2"])
    assert abs(p.sum() - 1.0) < 1e-12 and abs(p[0] - 2 / 76.0) < 1e-15


def test_windows(golden_dir):
    g = _g(golden_dir, "windows.npz")
    tile, lab = g["tile"], g["lab"]
    for tag in "abcde":
        s, st, idx, bs = [int(v) for v in g["args_" + tag]]
        p, cl, pos = H.create_patches_per_map(tile, lab, s, st, idx, bs)
        np.testing.assert_array_equal(np.asarray(pos), g["pos_" + tag])
        np.testing.assert_array_equal(p.reshape(len(p), -1).sum(axis=1), g["psum_" + tag])
        np.testing.assert_array_equal(p[0], g["p0_" + tag])
        np.testing.assert_array_equal(p[-1], g["plast_" + tag])
        np.testing.assert_array_equal(cl, g["cl_" + tag])
        assert cl.dtype == g["cl_" + tag].dtype
    # known answer from SURVEY 8c(2): 5 rows x 7 cols, shift-back at both borders
    pos = g["pos_a"]
    assert len(pos) == 35
    assert sorted(set(pos[:, 0])) == [0, 12, 24, 36, 45]
    assert sorted(set(pos[:, 1])) == [0, 12, 24, 36, 48, 60, 65]


def test_best_size(golden_dir):
    g = _g(golden_dir, "best_size.npz")
    for i in range(4):
        for mode in ("loss", "acc"):
            sums = g["case%d_sums" % i].copy()
            cnt = g["case%d_cnt" % i].copy()
            ch = np.zeros(len(sums), dtype=np.int32)
            best = H.select_best_patch_size(str(g["case%d_dist" % i]), list(g["case%d_vals" % i]), sums, cnt, mode, ch)
            assert best == int(g["case%d_%s" % (i, mode)][0])
            np.testing.assert_array_equal(cnt, g["case%d_%s_occur_after" % (i, mode)])
            np.testing.assert_array_equal(ch, g["case%d_%s_chosen" % (i, mode)])
    # SURVEY 8c(3)
    assert int(g["case0_loss"][0]) == 85 and int(g["case0_acc"][0]) == 25


def test_normalize(golden_dir):
    g = _g(golden_dir, "normalize.npz")
    x = g["x"].copy()
    H.normalize_images(x, g["mean"], g["std"])
    np.testing.assert_array_equal(x, g["out"])
    np.testing.assert_array_equal(x[..., 3:], g["x"][..., 3:])


def test_select_batch(golden_dir):
    g = _g(golden_dir, "select_batch.npz")
    random.seed(0)
    shuffle, it = np.arange(10), 8
    for row in g["rec"]:
        shuffle, batch, it = H.select_batch(shuffle, 4, it, 10)
        np.testing.assert_array_equal(np.concatenate([batch, [it], shuffle]), row)
    np.testing.assert_array_equal(g["rec"][0][:5], [8, 9, 6, 9, 2])


def test_patches_eval_and_train(golden_dir):
    g = _g(golden_dir, "patches.npz")
    tiles, labs, inst = [g["tile0"], g["tile1"]], [g["lab0"], g["lab1"]], g["inst"]
    for s in (9, 12, 25):
        p, c, m = H.dynamically_create_patches(tiles, labs, inst, s, is_train=False)
        np.testing.assert_array_equal(p, g["eval_p_%d" % s])
        np.testing.assert_array_equal(c, g["eval_c_%d" % s])
        np.testing.assert_array_equal(m, g["eval_m_%d" % s])
        np.random.seed(1234 + s)
        p, c, m = H.dynamically_create_patches(tiles, labs, inst, s, is_train=True)
        np.testing.assert_array_equal(p, g["train_p_%d" % s])
        np.testing.assert_array_equal(c, g["train_c_%d" % s])
        np.testing.assert_array_equal(m, g["train_m_%d" % s])


def test_confusion(golden_dir):
    g = _g(golden_dir, "confusion.npz")
    for fn in (H.calc_accuracy_by_crop, H.calc_accuracy_by_crop_loop):
        track = np.zeros((6, 6), dtype=np.uint32)
        acc, accn, loc = fn(g["t"], g["p"], track, g["m"])
        assert acc == int(g["acc"]) and abs(accn - float(g["accn"])) < 1e-15
        np.testing.assert_array_equal(loc, g["loc"])
        np.testing.assert_array_equal(track, g["track"])
        track = np.zeros((6, 6), dtype=np.uint32)
        acc, accn, loc = fn(g["t"], g["p"], track, None)
        assert acc == int(g["acc2"]) and abs(accn - float(g["accn2"])) < 1e-15
        np.testing.assert_array_equal(loc, g["loc2"])
        track = np.zeros((6, 6), dtype=np.uint32)
        acc, accn, loc = fn(g["t3"], g["p"], track, g["m"])
        assert acc == int(g["acc3"]) and abs(accn - float(g["accn3"])) < 1e-15
        np.testing.assert_array_equal(loc, g["loc3"])


def test_stitch_average_is_of_logits():
    rng = np.random.default_rng(0)
    h, w, K, s = 30, 41, 6, 12
    st = H.stride_for(s)
    nh, nw = H.window_counts(h, w, s, st)
    tile = rng.uniform(size=(h, w, 3))
    lab = np.zeros((h, w), dtype=np.uint8)
    batches, truth = [], rng.normal(size=(h, w, K)).astype(np.float32)
    nb = -(-nh * nw // 5)
    for i in range(nb):
        p, _, pos = H.create_patches_per_map(tile, lab, s, st, i, 5)
        lg = np.stack([truth[int(a):int(a) + s, int(b):int(b) + s] for a, b in pos])
        batches.append((lg, pos))
    prob, occur, am = H.stitch_tile(h, w, K, s, batches)
    assert occur.min() >= 1
    np.testing.assert_allclose(prob / occur, truth, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(am, truth.argmax(axis=2))
