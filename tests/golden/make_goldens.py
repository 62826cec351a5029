#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's numpy host helpers.

Runs only in the build container (needs /root/reference).  The reference's three
un-installable imports (tensorflow, gdal, skimage) are satisfied by inert placeholder
modules: none of the functions exercised below touches them -- they are pure
numpy/scipy.  One alias removed from numpy 2 is restored (``np.int``, used at
isprs_dilated_random.py:327).  Outputs are data only (inputs + expected outputs).

    python tests/golden/make_goldens.py
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Inert(object):
    def __getattr__(self, k):
        return _Inert()

    def __call__(self, *a, **k):
        return _Inert()


def import_reference():
    for name in ("tensorflow", "gdal", "skimage"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage"].img_as_float = lambda x: x
    tf = sys.modules["tensorflow"]
    tf.contrib = _Inert()
    tf.float32 = "float32"
    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, REF)
    import isprs_dilated_random as isprs
    return isprs


def main():
    ref = import_reference()
    out = {}

    # (1) multinomial probabilities
    np.savez(os.path.join(HERE, "multinomial_probs.npz"),
             values=np.array([25, 50, 75, 100]), probs=ref.define_multinomial_probs([25, 50, 75, 100]),
             values2=np.array([25, 45, 65, 85]), probs2=ref.define_multinomial_probs([25, 45, 65, 85]))

    # (2) sliding-window tiler: ramp tile, shift-back at both borders, batch offsets
    h, w, c = 70, 90, 5
    tile = (np.arange(h * w * c, dtype=np.float64).reshape(h, w, c)) / 1000.0
    lab = (np.arange(h * w).reshape(h, w) % 6).astype(np.uint8)
    d = dict(tile=tile, lab=lab)
    for tag, (s, st, idx, bs) in dict(a=(25, 12, 0, 64), b=(25, 12, 1, 16), c=(25, 12, 2, 16),
                                      d=(20, 10, 0, 8), e=(20, 10, 3, 8)).items():
        p, cl, pos = ref.create_patches_per_map(tile, lab, s, st, idx, bs)
        d["args_" + tag] = np.array([s, st, idx, bs])
        d["pos_" + tag] = np.asarray(pos)
        d["psum_" + tag] = p.reshape(len(p), -1).sum(axis=1)
        d["p0_" + tag] = p[0]
        d["plast_" + tag] = p[-1]
        d["cl_" + tag] = cl
    np.savez_compressed(os.path.join(HERE, "windows.npz"), **d)

    # (3) best patch size
    d = {}
    cases = [("multi_fixed", [25, 45, 65, 85], [3, 1, 2, 0], [1, 1, 0, 2]),
             ("multi_fixed", [25, 45, 65, 85], [3.5, 1.25, 2, 4], [2, 1, 3, 2]),
             ("uniform", [25, 30], [0.5, 0.9, 0.2, 0.0, 0.7, 0.1], [1, 2, 1, 0, 1, 1]),
             ("multinomial", [25, 30], [2.5, 0.9, 1.2, 3.0, 0.7, 1.1], [3, 2, 1, 4, 1, 1])]
    for i, (dist, vals, sums, cnt) in enumerate(cases):
        for mode in ("loss", "acc"):
            s_ = np.array(sums, dtype=np.float32)
            c_ = np.array(cnt, dtype=np.int32)
            ch = np.zeros(len(sums), dtype=np.int32)
            best = ref.select_best_patch_size(dist, vals, s_, c_, mode, ch)
            d["case%d_%s" % (i, mode)] = np.array([best])
            d["case%d_%s_occur_after" % (i, mode)] = c_
            d["case%d_%s_chosen" % (i, mode)] = ch
        d["case%d_vals" % i] = np.array(vals)
        d["case%d_sums" % i] = np.array(sums, dtype=np.float32)
        d["case%d_cnt" % i] = np.array(cnt, dtype=np.int32)
        d["case%d_dist" % i] = np.array(dist)
    np.savez(os.path.join(HERE, "best_size.npz"), **d)

    # (4) normalise: channels 0..2 only
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1, size=(3, 6, 6, 5))
    mean = np.array([0.5, 0.4, 0.3, 9.0, 9.0])
    std = np.array([0.25, 0.2, 0.1, 9.0, 9.0])
    xn = x.copy()
    ref.normalize_images(xn, mean, std)
    np.savez(os.path.join(HERE, "normalize.npz"), x=x, mean=mean, std=std, out=xn)

    # (5) select_batch walk with wrap-around
    random.seed(0)
    shuffle = np.arange(10)
    it = 8
    rec = []
    for step in range(6):
        shuffle, batch, it = ref.select_batch(shuffle, 4, it, 10)
        rec.append(np.concatenate([batch, [it], shuffle]))
    np.savez(os.path.join(HERE, "select_batch.npz"), rec=np.asarray(rec))

    # (6)+(7) patch crop, eval and train (RNG order + scipy rotate pinned)
    rng = np.random.default_rng(11)
    tiles = [rng.uniform(0, 1, size=(80, 100, 5)), rng.uniform(0, 1, size=(64, 70, 5))]
    labs = [rng.integers(0, 6, size=(80, 100)).astype(np.uint8), rng.integers(0, 6, size=(64, 70)).astype(np.uint8)]
    inst = np.array([[0, 0, 0, 30], [0, 70, 90, 45], [0, 10, 95, 200], [1, 60, 3, 359], [1, 20, 20, 90],
                     [0, 33, 41, 17], [1, 5, 50, 123], [0, 79, 99, 270]])
    d = dict(tile0=tiles[0], tile1=tiles[1], lab0=labs[0], lab1=labs[1], inst=inst)
    for s in (9, 12, 25):
        p, cl, mk = ref.dynamically_create_patches(tiles, labs, inst, s, is_train=False)
        d["eval_p_%d" % s], d["eval_c_%d" % s], d["eval_m_%d" % s] = p, cl, mk
        np.random.seed(1234 + s)
        p, cl, mk = ref.dynamically_create_patches(tiles, labs, inst, s, is_train=True)
        d["train_p_%d" % s], d["train_c_%d" % s], d["train_m_%d" % s] = p, cl, mk
    np.savez_compressed(os.path.join(HERE, "patches.npz"), **d)

    # (8) confusion matrix / accuracies
    rng = np.random.default_rng(3)
    t = rng.integers(0, 6, size=(3, 7, 7))
    p = rng.integers(0, 6, size=(3, 7, 7))
    m = rng.integers(0, 2, size=(3, 7, 7)).astype(bool)
    track = np.zeros((6, 6), dtype=np.uint32)
    acc, accn, loc = ref.calc_accuracy_by_crop(t, p, track, m)
    track2 = np.zeros((6, 6), dtype=np.uint32)
    acc2, accn2, loc2 = ref.calc_accuracy_by_crop(t, p, track2, None)
    # a case with an empty class row (class 5 never true)
    t3 = np.minimum(t, 4)
    track3 = np.zeros((6, 6), dtype=np.uint32)
    acc3, accn3, loc3 = ref.calc_accuracy_by_crop(t3, p, track3, m)
    np.savez(os.path.join(HERE, "confusion.npz"), t=t, p=p, m=m, acc=acc, accn=accn, loc=loc, track=track,
             acc2=acc2, accn2=accn2, loc2=loc2, t3=t3, acc3=acc3, accn3=accn3, loc3=loc3)

    # (10) class-balanced sampling helpers that feed the hot path
    rng = np.random.default_rng(5)
    labels = [rng.integers(0, 6, size=(60, 75)).astype(np.uint8), rng.integers(0, 6, size=(50, 50)).astype(np.uint8)]
    # make region structure so that majority classes differ
    labels[0][:30, :40] = 1
    labels[0][30:, 40:] = 3
    labels[1][:, :25] = 5
    dist = ref.create_distributions_over_classes(labels, 25, 5)
    d = dict(lab0=labels[0], lab1=labels[1])
    for k in range(6):
        d["class%d" % k] = np.asarray(dist[k], dtype=np.int64).reshape(-1, 3)
    np.random.seed(99)
    rot = ref.create_rotation_distribution(dist)
    for k in range(6):
        d["rot%d" % k] = np.asarray(rot[k])
    random.seed(21)
    np.random.seed(21)
    sb = ref.select_super_batch_instances(dist, rot, batch_size=7, super_batch=5)
    d["super_batch"] = sb
    data = [rng.uniform(0, 1, size=(60, 75, 5)), rng.uniform(0, 1, size=(50, 50, 5))]
    mean_full, std_full = ref.dynamically_calculate_mean_and_std(data, dist, 25)
    d["data0"], d["data1"], d["mean_full"], d["std_full"] = data[0], data[1], mean_full, std_full
    np.savez_compressed(os.path.join(HERE, "sampling.npz"), **d)
    coffee_goldens()
    contest_goldens()
    print("goldens written to", HERE)


class _NpPy2(object):
    """`np` as the coffee script sees it: its `np.empty([len(files) / 2, ...])` (coffee:85-86) relies on Python 2's integer `/`."""

    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def empty(shape, *a, **k):
        return np.empty([int(v) for v in shape], *a, **k)


def coffee_goldens():
    """coffee_dilated_random.py: the flip-by-index patch sampler with its float16 cast (:241-293), normalisation of those float16
    patches (:67-74), the square-tile window enumeration (:296-349), class distributions (:358-372), mean / std (:352-355) and the
    Torch-ASCII reader (:84-125)."""
    import importlib
    import tempfile
    import_reference()
    cf = importlib.import_module("coffee_dilated_random")
    cf.np = _NpPy2()
    sys.path.insert(0, HERE)
    from torch_ascii_fixture import write_torch_ascii
    rng = np.random.default_rng(101)
    data = rng.uniform(0, 1, size=(2, 40, 40, 3)).astype(np.float32)
    mask = np.zeros((2, 40, 40, 1), dtype=np.float32)
    mask[0, :22, :17] = 1
    mask[1, 10:, 25:] = 1
    mask[1, 30:, :8] = 1
    d = dict(data=data, mask=mask)
    dist = cf.create_distributions_over_classes([mask[0], mask[1]], 9, 4)
    d["dist"] = np.array([(k, i, j) for (k, (i, j)) in dist], dtype=np.int64)
    mean, std = cf.create_mean_and_std(data, mask, 9, 4)
    d["mean"], d["std"] = mean, std
    # windows were enumerated at the reference crop size (9) but are cut at the step's size (13): the border ones shift back
    n = len(dist)
    shuffle = np.array([0, n - 1, n + 3, 2 * n - 1, 2 * n, 3 * n - 1, 7, n + 7, 2 * n + 7, n // 2, n + n // 2, 2 * n + n // 2, 1, n + 1])
    for s in (9, 13):
        p, c = cf.dynamically_create_patches(data, mask, s, dist, shuffle)
        d["patches16_%d" % s], d["classes_%d" % s] = p, c                      # float16 / int8 as returned
        pn = p.copy()
        cf.normalize_images(pn, mean, std)                                       # in place, on the float16 array
        d["normalized16_%d" % s] = pn
    d["shuffle"] = shuffle
    for tag, (s, st, idx, bs) in dict(a=(13, 6, 0, 64), b=(13, 6, 1, 10), c=(9, 4, 3, 7)).items():
        p, c, pos = cf.create_patches_per_map(data[0], mask[0], s, st, idx, bs)
        d["win_args_" + tag] = np.array([s, st, idx, bs])
        d["win_pos_" + tag] = np.asarray(pos)
        d["win_psum_" + tag] = p.reshape(len(p), -1).sum(axis=1)
    # Torch-ASCII reader: two (image, mask) pairs of the hard-wired 500 x 500 size, written from seeds (torch_ascii_fixture.py)
    with tempfile.TemporaryDirectory() as tmp:
        for i, nm in enumerate(["B_img.txt", "b_mask.txt", "a_img.txt", "A_mask.txt"]):     # case-insensitive sort: a_img, A_mask, B_img, b_mask
            write_torch_ascii(os.path.join(tmp, nm), seed=500 + i, c=1 if "mask" in nm else 3, h=500, w=500, is_mask="mask" in nm)
        imgs, masks = cf.load_images_torch(tmp + "/")
    d["torch_names"] = np.array(["B_img.txt", "b_mask.txt", "a_img.txt", "A_mask.txt"])
    d["torch_seeds"] = np.array([500, 501, 502, 503])
    d["torch_img_shape"], d["torch_mask_shape"] = np.array(imgs.shape), np.array(masks.shape)
    d["torch_img_sample"] = imgs[:, ::37, ::41, :].copy()
    d["torch_img_sum"] = imgs.astype(np.float64).sum(axis=(1, 2))
    d["torch_mask_sample"] = masks[:, ::37, ::41, :].copy()
    d["torch_mask_sum"] = masks.astype(np.float64).sum(axis=(1, 2, 3))
    np.savez_compressed(os.path.join(HERE, "coffee.npz"), **d)


def contest_goldens():
    """contest_dilated_random.py: class distributions with their quirks (:172-190), mean / std over those windows (:99-113), the
    flip-by-index sampler with the void mask (:192-254), the non-square window enumeration with masks (:257-324), the PGM label
    reader (:119-143) and the accuracy function that never counts (:327-345: `mask[i, j, k] is True` is False for numpy bools)."""
    import importlib
    import_reference()
    ct = importlib.import_module("contest_dilated_random")
    rng = np.random.default_rng(202)
    data = rng.uniform(0, 1, size=(60, 50, 3)).astype(np.float32)
    lab = rng.integers(0, 7, size=(60, 50)).astype(np.int64)
    lab[:20, :30] = 3                     # a uniform region (windows inside it are all one class)
    lab[40:, 20:] = 7                     # a void region
    lab[25:35, 5:15] = 6
    d = dict(data=data, lab=lab)
    dist = ct.create_distributions_over_classes(lab, 9, 4)
    d["dist"] = np.array(dist, dtype=np.int64).reshape(-1, 2)
    mean, std = ct.create_mean_and_std(data, dist, 9)
    d["mean"], d["std"] = mean, std
    n = len(dist)
    shuffle = np.array([0, n - 1, n + 2, 2 * n - 1, 2 * n, 3 * n - 1, 5, n + 5, 2 * n + 5, n // 3, n + n // 3, 2 * n + n // 3])
    for s in (9, 14):
        p, c, m = ct.dynamically_create_patches(data, lab, s, dist, shuffle)
        d["patches_%d" % s], d["classes_%d" % s], d["masks_%d" % s] = p, c, m
    d["shuffle"] = shuffle
    for tag, (s, st, idx, bs) in dict(a=(14, 7, 0, 64), b=(14, 7, 1, 9), c=(9, 4, 5, 8)).items():
        p, c, m, pos = ct.create_patches_per_map(data, lab, s, st, idx, bs)
        d["win_args_" + tag] = np.array([s, st, idx, bs])
        d["win_pos_" + tag] = np.asarray(pos)
        d["win_psum_" + tag] = p.reshape(len(p), -1).sum(axis=1)
        d["win_mask_" + tag] = m
    with open(os.path.join(HERE, "contest_gt.pgm")) as fh:
        d["pgm"] = ct.read_pgm(fh)
    t = rng.integers(0, 7, size=(2, 5, 5))
    p = rng.integers(0, 7, size=(2, 5, 5))
    m = np.ones((2, 5, 5), dtype=bool)
    track = np.zeros((7, 7), dtype=np.uint32)
    res = ct.calc_cccuracy_by_crop(t, p, m, track)
    d["acc_quirk"] = np.array([float(res[0])])        # 0: documented quirk, deliberately not reproduced
    d["acc_quirk_track_sum"] = np.array([int(track.sum())])
    np.savez_compressed(os.path.join(HERE, "contest.npz"), **d)


if __name__ == "__main__":
    main()
