"""CPU: the product's host-side sampling / augmentation logic against the reference-generated goldens, and the
C-ABI library's exported symbols against include/drs.h (no compute calls without a GPU)."""
import os
import random
import re

import numpy as np
import scipy.ndimage

from drs_amd import patches as P


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_library_exports_every_declared_symbol():
    from drs_amd import _lib
    lib = _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "drs.h")).read()
    pat = r"^(?:int|void|float|long long|size_t)\s+(drs_[a-z0-9_]+)\s*\("
    declared = set(re.findall(pat, hdr, flags=re.M))      # every function of the header
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    # the product library exports the drop-in boundary and nothing else; the development switches (include/drs_dev.h) exist in
    # libdrs_hip_dev.so only, which the package's product path never loads
    import subprocess
    exported = set(re.findall(r" T (drs_[a-z0-9_]+)$", subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True,
                                                                        check=True).stdout, flags=re.M))
    assert exported == declared, exported ^ declared
    dev_hdr = open(os.path.join(root, "include", "drs_dev.h")).read()
    dev_declared = set(re.findall(pat, dev_hdr, flags=re.M))
    assert dev_declared == set(_lib.DEV_SIGNATURES) and all(n.startswith("drs_debug_") for n in dev_declared)
    dev = _lib.dev()
    for name in declared | dev_declared:
        assert hasattr(dev, name)
    # pure size queries are host code and callable without a GPU
    assert _lib.query("drs_conv_mtile", 256) == 128 and _lib.query("drs_conv_mtile", 64) == 128 and _lib.query("drs_conv_mtile", 32) == 256
    assert _lib.query("drs_bn_backward_rows", 2, 16, 64, 0) == 16 and _lib.query("drs_bn_backward_rows", 128, 64, 64, 0) == 2048
    assert _lib.query("drs_bn_backward_rows", 128, 64, 256, 1) == 128 * 16 * 4  # 4 columns per workgroup, four strips of 16 rows (5120 workgroups aimed at)
    assert _lib.query("drs_conv_wgrad_splits", 64, 64, 3, 256, 256) >= 1


def test_select_batch_and_probs(golden_dir):
    g = _g(golden_dir, "select_batch.npz")
    random.seed(0)
    shuffle, it = np.arange(10), 8
    for row in g["rec"]:
        shuffle, batch, it = P.select_batch(shuffle, 4, it, 10)
        np.testing.assert_array_equal(np.concatenate([batch, [it], shuffle]), row)
    m = _g(golden_dir, "multinomial_probs.npz")
    np.testing.assert_allclose(P.define_multinomial_probs(list(m["values"])), m["probs"], rtol=0, atol=1e-18)
    np.testing.assert_allclose(P.define_multinomial_probs(list(m["values2"])), m["probs2"], rtol=0, atol=1e-18)


def test_patch_size_draws_follow_numpy_streams():
    vals = [25, 45, 65, 85]
    np.random.seed(3)
    a = [P.draw_patch_size("uniform", vals)[0] for _ in range(200)]
    np.random.seed(3)
    b = [int(np.random.uniform(25, 86, 1)[0]) for _ in range(200)]
    assert a == b and min(a) >= 25 and max(a) <= 85
    np.random.seed(4)
    s, i = P.draw_patch_size("multi_fixed", vals)
    assert s == vals[i]
    probs = P.define_multinomial_probs(vals)
    np.random.seed(5)
    s, i = P.draw_patch_size("multinomial", vals, probs)
    assert s == 25 + i and 0 <= i < 61
    assert P.draw_patch_size("single_fixed", vals) == (25, None)


def test_window_positions(golden_dir):
    g = _g(golden_dir, "windows.npz")
    h, w = g["tile"].shape[:2]
    for tag in "abcde":
        s, st, idx, bs = [int(v) for v in g["args_" + tag]]
        np.testing.assert_array_equal(P.window_positions(h, w, s, st, idx, bs), g["pos_" + tag].astype(np.int64))
    assert P.window_counts(6000, 6000, 64, 32) == (187, 187)        # 34 969 windows


def test_rotation_emulation_equals_scipy():
    rng = np.random.default_rng(0)
    for S in (7, 12, 25, 64, 85):
        img = rng.uniform(1, 2, size=(S, S))
        for angle in list(rng.integers(0, 360, size=12)) + [0, 90, 180, 270, 45, 359]:
            want = scipy.ndimage.rotate(img, angle, order=0, reshape=False)
            si, sj, valid = P.nearest_source_index(P.rotation_params(angle, S), S)
            got = np.where(valid, img[np.clip(si, 0, S - 1), np.clip(sj, 0, S - 1)], 0.0)
            np.testing.assert_array_equal(got, want, err_msg="S=%d angle=%d" % (S, angle))


def _emulate_crop(tiles, labs, inst, S, aug, mean=None, std=None):
    """numpy statement of drs_crop_normalize (same index algebra as the kernel)."""
    B, C = len(inst), tiles[0].shape[2]
    out = np.zeros((B, S, S, C))
    lab = np.zeros((B, S, S), dtype=np.int64)
    msk = np.zeros((B, S, S), dtype=bool)
    for b in range(B):
        m, x, y = int(inst[b][0]), int(inst[b][1]), int(inst[b][2])
        x, y = min(x, tiles[m].shape[0] - S), min(y, tiles[m].shape[1] - S)
        ii, jj = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
        fi = S - 1 - ii if aug.flip[b] == 1 else ii
        fj = S - 1 - jj if aug.flip[b] == 2 else jj
        if aug.rot_on[b]:
            si, sj, valid = P.nearest_source_index(aug.rot[b], S)
            si, sj, valid = si[fi, fj], sj[fi, fj], valid[fi, fj]
        else:
            si, sj, valid = fi, fj, np.ones((S, S), dtype=bool)
        sic, sjc = np.clip(si, 0, S - 1), np.clip(sj, 0, S - 1)
        v = np.where(valid[..., None], tiles[m][x + sic, y + sjc, :], 0.0)
        if aug.noise_on[b]:
            v = v + aug.noise[b][fi, fj, :]
        out[b] = v
        lab[b] = np.where(valid, labs[m][x + sic, y + sjc], 0)
        msk[b] = valid
    return out, lab, msk


def test_augmentation_draw_order_and_crop_algebra(golden_dir):
    g = _g(golden_dir, "patches.npz")
    tiles, labs, inst = [g["tile0"], g["tile1"]], [g["lab0"], g["lab1"]], g["inst"]
    for s in (9, 12, 25):
        np.random.seed(1234 + s)
        aug = P.draw_augmentation(inst, s, 5, noise="host")
        p, c, m = _emulate_crop(tiles, labs, inst, s, aug)
        np.testing.assert_array_equal(p, g["train_p_%d" % s])
        np.testing.assert_array_equal(c, g["train_c_%d" % s])
        np.testing.assert_array_equal(m, g["train_m_%d" % s])


def test_missing_library_fails_loudly(monkeypatch):
    """There is no CPU fallback: without libdrs_hip.so the binding raises, and so does everything built on it."""
    import pytest
    from drs_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdrs_hip.so")
    with pytest.raises(_lib.DrsError):
        _lib.load()
    with pytest.raises(_lib.DrsError):
        _lib.call("drs_conv_mtile", 64)


def test_entry_points_reject_bad_arguments_before_any_launch():
    """Argument validation is host code: every rejected call returns DRS_ERR_ARG (raised as DrsError) without touching the
    device, so it can be exercised without a GPU.  Pointers are dummies; a call that passed validation would launch."""
    import pytest
    from drs_amd import _lib
    _lib.load()
    p = 0x1000
    bad = [
        ("drs_conv_forward", (p, 2, 8, 2, 64, 0, p, p, 3, 2, 2, 48, 64, p, 64, 0, 0, None, None)),               # cin % 32
        ("drs_conv_forward", (p, 2, 8, 1, 64, 0, p, p, 3, 2, 2, 64, 64, p, 64, 0, 0, None, None)),               # halo < pad
        ("drs_conv_forward", (p, 4096, 4096, 2, 64, 0, p, p, 3, 2, 2, 64, 64, p, 64, 0, 0, None, None)),         # B*S*S >= 2^24
        ("drs_conv_forward_split", (p, 2, 8, 2, 64, 0, p, p, 3, 2, 2, 64, 96, p, 96, 0, 0, None, 2, None)),     # cout % 64
        ("drs_conv_forward_split", (p, 2, 8, 2, 64, 0, p, p, 3, 2, 2, 64, 64, p, 64, 0, 0, None, 4, None)),     # nterms
        ("drs_conv_forward_split", (p, 2, 8, 2, 72, 0, p, p, 3, 2, 2, 64, 64, p, 64, 0, 0, None, 2, None)),     # ld % 32
        ("drs_conv_wgrad_split", (p, 2, 8, 2, 64, 0, p, 2, 64, 8, 3, 2, 2, 64, 64, 64, p, p, 2, None)),         # coff_g % 32
        ("drs_split_terms", (p, 100, 2, p, None)),                                                              # n % 32
        ("drs_filter_split", (p, 3, 5, 24, 64, 2, p, None, None)),                                              # cin_pad % 32
        ("drs_bn_act_pool_forward_terms", (p, 2, 8, 64, p, 0.1, 1, None, 2, 64, 0, None, None, 2, None)),       # no terms
        ("drs_bn_backward_apply_terms", (p, p, 2, 8, 64, p, p, 128.0, None, 2, 72, 0, p, 2, None)),             # ld % 32
    ]
    for name, args in bad:
        with pytest.raises(_lib.DrsError):
            _lib.call(name, *args)
    assert _lib.query("drs_split_conv_mtile", 256) == 128 and _lib.query("drs_split_conv_mtile", 192) == 128
    assert _lib.query("drs_split_conv_mtile", 64) == 128
    assert _lib.query("drs_conv_wgrad_split_splits", 64, 64, 3, 256, 256, 8, 2) >= 1
