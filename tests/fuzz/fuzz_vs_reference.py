#!/usr/bin/env python3
"""Differential fuzz of the oracle's host half (oracle/host_ref.py) and the product's host functions (drs_amd.patches / sampling /
loops) against the REFERENCE'S OWN FUNCTIONS, imported from /root/reference the way tests/golden/make_goldens.py does (inert
placeholders for tensorflow / gdal / skimage; np.int restored).  Random inputs, same numpy / random seeds on both sides, exact
equality.  Runs only where /root/reference exists (the build container): nothing here travels to the GPU box, and nothing of the
reference is copied -- it is called.      python tests/fuzz/fuzz_vs_reference.py [n=300] [seed=0]"""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if not os.path.isdir("/root/reference"):
    print("no /root/reference here: nothing to compare against")
    sys.exit(0)
from make_goldens import import_reference
from oracle import host_ref as H
from drs_amd import patches as P
from drs_amd import sampling as SM
from drs_amd import loops as L

ref = import_reference()
fails = []


def check(name, ok, detail=""):
    if not ok:
        fails.append((name, detail))
        print("FAIL", name, detail, flush=True)


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.size == 0 and b.size == 0:           # (an empty batch: the reference returns [] where the build returns shape (0, 2))
        return True
    return a.shape == b.shape and np.array_equal(a, b)


def one(rng, i):
    # ---- crop + augmentation
    C = int(rng.choice([3, 4, 5]))
    h, w = int(rng.integers(30, 120)), int(rng.integers(30, 120))
    S = int(rng.integers(5, min(h, w, 35) + 1))
    B = int(rng.integers(1, 7))
    tiles = [rng.uniform(size=(h, w, C)) for _ in range(2)]
    labs = [rng.integers(0, 6, size=(h, w)) for _ in range(2)]
    inst = np.stack([rng.integers(0, 2, size=B), rng.integers(0, h - 1, size=B), rng.integers(0, w - 1, size=B), rng.integers(0, 360, size=B)], 1)
    for train in (False, True):
        seed = int(rng.integers(0, 2 ** 31))
        np.random.seed(seed)
        a = ref.dynamically_create_patches(tiles, labs, inst, S, is_train=train)
        np.random.seed(seed)
        b = H.dynamically_create_patches(tiles, labs, inst, S, is_train=train)
        check("dynamically_create_patches", all(eq(x, y) for x, y in zip(a, b)), "train=%s S=%d %dx%d" % (train, S, h, w))
    mean, std = rng.uniform(0.2, 0.6, size=C), rng.uniform(0.1, 0.4, size=C)
    x1 = rng.uniform(size=(B, S, S, C)); x2 = x1.copy()
    ref.normalize_images(x1, mean, std); H.normalize_images(x2, mean, std)
    check("normalize_images", eq(x1, x2))
    # ---- windows
    S2 = int(rng.integers(4, 34)); st = int(rng.integers(1, S2 + 1))
    h2, w2 = int(rng.integers(S2, S2 + 80)), int(rng.integers(S2, S2 + 80))
    bs = int(rng.integers(1, 40))
    tile = rng.uniform(size=(h2, w2, 2)); lab = rng.integers(0, 6, size=(h2, w2))
    nh, nw = H.window_counts(h2, w2, S2, st)
    idx = int(rng.integers(0, max(1, -(-nh * nw // bs))))
    a = ref.create_patches_per_map(tile, lab, S2, st, idx, bs)
    b = H.create_patches_per_map(tile, lab, S2, st, idx, bs)
    check("create_patches_per_map", eq(a[0], b[0]) and eq(a[1], b[1]) and eq(np.asarray(a[2]), np.asarray(b[2])), "%dx%d S%d st%d idx%d bs%d" % (h2, w2, S2, st, idx, bs))
    check("window_positions", eq(np.asarray(a[2]).astype(np.int64), P.window_positions(h2, w2, S2, st, idx, bs)), "%dx%d S%d st%d idx%d bs%d" % (h2, w2, S2, st, idx, bs))
    # ---- accuracy / confusion
    K = 6
    t = rng.integers(0, K, size=(B, S, S)); p = rng.integers(0, K, size=(B, S, S)); m = rng.integers(0, 2, size=(B, S, S)).astype(bool)
    c1 = np.zeros((K, K), dtype=np.uint32); c2 = c1.copy()
    a = ref.calc_accuracy_by_crop(t, p, c1, m)
    b = H.calc_accuracy_by_crop(t, p, c2, m, K)
    check("calc_accuracy_by_crop", eq(a[0], b[0]) and np.allclose(a[1], b[1], rtol=0, atol=0) and eq(a[2], b[2]) and eq(c1, c2))
    # ---- batch walk
    n = int(rng.integers(3, 60)); bsz = int(rng.integers(1, n + 1)); it = int(rng.integers(0, n)); sd = int(rng.integers(0, 10 ** 6))
    sh = rng.permutation(n)
    random.seed(sd); np.random.seed(sd)
    a = ref.select_batch(sh.copy(), bsz, it, n)
    random.seed(sd); np.random.seed(sd)
    b = H.select_batch(sh.copy(), bsz, it, n)
    random.seed(sd); np.random.seed(sd)
    c = P.select_batch(sh.copy(), bsz, it, n)
    check("select_batch", all(eq(x, y) for x, y in zip(a, b)) and all(eq(x, y) for x, y in zip(a, c)), "n=%d bs=%d it=%d" % (n, bsz, it))
    # ---- size distributions and the best-size rule
    v0 = int(rng.integers(20, 40)); vals = sorted(set(int(v) for v in rng.integers(v0, v0 + 80, size=int(rng.integers(2, 6)))))
    if len(vals) >= 2:
        def attempt(f):
            try:
                return f(vals)
            except ZeroDivisionError:        # the reference divides by interval - len(values): consecutive values leave nothing to share
                return "ZeroDivisionError"
        ra, rb, rc = attempt(ref.define_multinomial_probs), attempt(H.define_multinomial_probs), attempt(P.define_multinomial_probs)
        check("define_multinomial_probs", (isinstance(ra, str) and ra == rb == rc) or (not isinstance(ra, str) and eq(ra, rb) and eq(ra, rc)), str(vals))
        for dist in ("multi_fixed", "uniform", "multinomial"):
            ln = len(vals) if dist == "multi_fixed" else vals[-1] - vals[0] + 1
            sums = rng.uniform(0, 5, size=ln).astype(np.float32); cnt = rng.integers(0, 4, size=ln).astype(np.int32)
            for kind in ("acc", "loss"):
                c1, c2, c3 = cnt.copy(), cnt.copy(), cnt.copy()
                a = ref.select_best_patch_size(dist, vals, sums.copy(), c1, kind)
                b = H.select_best_patch_size(dist, vals, sums.copy(), c2, kind)
                c = L.select_best_patch_size(dist, vals, sums.copy(), c3, kind)
                check("select_best_patch_size", a == b == c and eq(c1, c2) and eq(c1, c3), "%s %s %s" % (dist, kind, vals))
    # ---- class distributions, super-batches, mean / std
    if i % 5 == 0:
        labs2 = [rng.integers(0, 6, size=(int(rng.integers(40, 90)), int(rng.integers(40, 90)))) for _ in range(2)]
        cs, sc = int(rng.integers(8, 26)), int(rng.integers(3, 20))
        a = ref.create_distributions_over_classes(labs2, cs, sc)
        b = SM.create_distributions_over_classes(labs2, cs, sc)
        check("create_distributions_over_classes", len(a) == len(b) and all(eq(np.asarray(x).reshape(-1, 3) if len(x) else np.zeros((0, 3)), np.asarray(y).reshape(-1, 3) if len(y) else np.zeros((0, 3))) for x, y in zip(a, b)), "crop %d stride %d" % (cs, sc))
        if all(len(x) > 0 for x in a):
            sd = int(rng.integers(0, 10 ** 6))
            np.random.seed(sd); random.seed(sd)
            rot_a = ref.create_rotation_distribution(a)
            ia = ref.select_super_batch_instances(a, rot_a, batch_size=12, super_batch=5)
            np.random.seed(sd); random.seed(sd)
            rot_b = SM.create_rotation_distribution(b)
            ib = SM.select_super_batch_instances(b, rot_b, batch_size=12, super_batch=5)
            check("select_super_batch_instances", eq(np.asarray(ia), np.asarray(ib)))
            data = [rng.uniform(size=(l.shape[0], l.shape[1], 4)) for l in labs2]
            ma = ref.dynamically_calculate_mean_and_std(data, a, cs)
            mb = SM.dynamically_calculate_mean_and_std(data, b, cs)
            check("dynamically_calculate_mean_and_std", all(eq(x, y) for x, y in zip(ma, mb)))


_flav = {}


def flavours(rng):
    """coffee_dilated_random.py / contest_dilated_random.py: class distributions (with their quirks), mean / std, the flip-by-index
    sampler (float16 cast; void mask), window enumeration -- the reference's functions against drs_amd.loops_indexed /
    oracle.host_ref / drs_amd.patches"""
    import importlib
    from make_goldens import _NpPy2
    from drs_amd import loops_indexed as LI
    if not _flav:
        _flav["cf"] = importlib.import_module("coffee_dilated_random")
        _flav["cf"].np = _NpPy2()
        _flav["ct"] = importlib.import_module("contest_dilated_random")
    cf, ct = _flav["cf"], _flav["ct"]
    # ---- coffee: square tiles, binary masks
    n, hw = int(rng.integers(1, 4)), int(rng.integers(24, 60))
    data = rng.uniform(0, 1, size=(n, hw, hw, 3)).astype(np.float32)
    mask = np.zeros((n, hw, hw, 1), dtype=np.float32)
    for k in range(n):
        for _ in range(int(rng.integers(1, 4))):
            a, b, c, d = sorted(rng.integers(0, hw, size=2)), sorted(rng.integers(0, hw, size=2)), 0, 0
            mask[k, a[0]:a[1] + 1, b[0]:b[1] + 1] = 1
    cs, st = int(rng.integers(5, 12)), int(rng.integers(2, 8))
    da = cf.create_distributions_over_classes([mask[k] for k in range(n)], cs, st)
    db = LI.create_distributions_over_classes([mask[k] for k in range(n)], cs, st, 2)
    flat = np.array([(k, i, j) for (k, (i, j)) in da], dtype=np.int64).reshape(-1, 3)
    check("coffee create_distributions_over_classes", eq(flat, np.asarray(db, dtype=np.int64).reshape(-1, 3)), "crop %d stride %d" % (cs, st))
    ma = cf.create_mean_and_std(data, mask, cs, st)
    mb = LI.create_mean_and_std(data, cs, st)
    check("coffee create_mean_and_std", np.allclose(ma[0], mb[0], rtol=2e-6, atol=0) and np.allclose(ma[1], mb[1], rtol=2e-6, atol=0))
    if len(da):
        nd = len(da)
        shuffle = rng.integers(0, 3 * nd, size=int(rng.integers(1, 12)))
        S = int(rng.integers(cs, min(hw, cs + 8) + 1))
        pa, ca = cf.dynamically_create_patches(data, mask, S, da, shuffle)
        pb, cb, _ = H.indexed_create_patches(data, mask[..., 0], S, [(int(k), int(i), int(j)) for k, i, j in flat], shuffle, float16=True)
        check("coffee dynamically_create_patches", pa.dtype == pb.dtype and eq(pa, pb) and eq(ca[..., 0], cb), "S %d" % S)
        qa = pa.copy(); cf.normalize_images(qa, ma[0], ma[1])
        check("coffee normalize_images (float16, in place)", eq(qa, H.normalize_images_f16(pb, ma[0], ma[1])))
    S2 = int(rng.integers(5, 16)); st2 = int(rng.integers(2, S2 + 1)); bs = int(rng.integers(1, 30))
    nw = H.window_counts(hw, hw, S2, st2)
    idx = int(rng.integers(0, max(1, -(-nw[0] * nw[1] // bs))))
    pos = cf.create_patches_per_map(data[0], mask[0], S2, st2, idx, bs)[2]
    check("coffee window positions", eq(np.asarray(pos).astype(np.int64), P.window_positions(hw, hw, S2, st2, idx, bs, "coffee")), "%d S%d st%d idx%d bs%d" % (hw, S2, st2, idx, bs))
    # ---- contest: one non-square tile, 7 classes + void (7)
    h, w = int(rng.integers(30, 70)), int(rng.integers(30, 70))
    img = rng.uniform(0, 1, size=(h, w, 3)).astype(np.float32)
    lab = rng.integers(0, 7, size=(h, w)).astype(np.int64)
    for _ in range(int(rng.integers(0, 4))):            # uniform regions and void regions: where the quirks live
        a, b = sorted(rng.integers(0, h, size=2)), sorted(rng.integers(0, w, size=2))
        lab[a[0]:a[1] + 1, b[0]:b[1] + 1] = int(rng.integers(0, 8))
    da = ct.create_distributions_over_classes(lab, cs, st)
    db = LI.create_distributions_over_classes_contest(lab, cs, st)
    fa = np.array(da, dtype=np.int64).reshape(-1, 2)
    check("contest create_distributions_over_classes", eq(fa, np.asarray(db, dtype=np.int64).reshape(-1, 3)[:, 1:]), "crop %d stride %d" % (cs, st))
    if len(da):
        ma = ct.create_mean_and_std(img, da, cs)
        mb = LI.create_mean_and_std_contest(img, db, cs)
        check("contest create_mean_and_std", np.allclose(ma[0], mb[0], rtol=2e-6, atol=0) and np.allclose(ma[1], mb[1], rtol=2e-6, atol=0))
        nd = len(da)
        shuffle = rng.integers(0, 3 * nd, size=int(rng.integers(1, 12)))
        S = int(rng.integers(cs, min(h, w, cs + 8) + 1))
        pa, ca, qa = ct.dynamically_create_patches(img, lab, S, da, shuffle)
        pb, cb, qb = H.indexed_create_patches(img[None], lab[None], S, [(0, int(i), int(j)) for i, j in fa], shuffle, void_label=7)
        check("contest dynamically_create_patches", eq(pa, pb) and eq(ca, cb) and eq(qa, qb), "S %d" % S)
    nh, nw2 = H.window_counts(h, w, S2, st2)
    idx = int(rng.integers(0, max(1, -(-nh * nw2 // bs))))
    try:
        res = ct.create_patches_per_map(img, lab, S2, st2, idx, bs)
    except Exception as e:                              # (contest:275's start index can run off a non-square tile)
        res = None
    if res is not None:
        check("contest window positions", eq(np.asarray(res[3]).astype(np.int64), P.window_positions(h, w, S2, st2, idx, bs, "contest")), "%dx%d S%d st%d idx%d bs%d" % (h, w, S2, st2, idx, bs))


def main(n=300, seed=0):
    rng = np.random.default_rng(seed)
    for i in range(n):
        one(rng, i)
        if i % 2 == 0:
            flavours(rng)
    names = sorted(set(f[0] for f in fails))
    print("%d rounds against the reference's own functions: %d mismatches%s" % (n, len(fails), (" in " + ", ".join(names)) if names else ""))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("n", 300)), int(kw.get("seed", 0)))
