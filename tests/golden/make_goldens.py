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
    print("goldens written to", HERE)


if __name__ == "__main__":
    main()
