"""-m gpu: drs_crop_normalize and the stitch kernels against the reference-generated goldens and oracle/host_ref.py.
Bit-exact (the kernel works in fp64 and rounds once to the float32 feed)."""
import os

import numpy as np
import pytest
import torch

from oracle import host_ref as H

pytestmark = pytest.mark.gpu

from gpu_util import DEV, dev, stream   # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _net(ch, B, S):
    from drs_amd.net import DilatedNet
    return DilatedNet("dilated_grsl", ch, 6, 0.005, b_max=B, s_max=S, device=DEV)


def _slab(net, B, S):
    slab, P, ld = net.input_slab()
    a = slab[:B * (S + 2 * P) ** 2 * ld].cpu().numpy().reshape(B, S + 2 * P, S + 2 * P, ld)
    return a, P, ld


def test_crop_matches_reference_goldens(golden_dir):
    from drs_amd import patches as P
    g = np.load(os.path.join(golden_dir, "patches.npz"))
    tiles, labs, inst = [g["tile0"], g["tile1"]], [g["lab0"], g["lab1"]], g["inst"]
    mean = np.array([0.5, 0.4, 0.3, 0.0, 0.0])
    std = np.array([0.25, 0.2, 0.1, 1.0, 1.0])
    pool = P.TilePool(tiles, labs, DEV, dtype=np.float64)
    for s in (9, 12, 25):
        net = _net(5, len(inst), s)
        for train in (False, True):
            aug = None
            if train:
                np.random.seed(1234 + s)
                aug = P.draw_augmentation(inst, s, 5, noise="host")
            P.crop_to_net(net, pool, inst, s, mean, std, aug)
            torch.cuda.synchronize()
            key = "train" if train else "eval"
            want = g["%s_p_%d" % (key, s)].copy()
            H.normalize_images(want, mean, std)                      # isprs:1745: normalise after the crop
            a, Pd, ld = _slab(net, len(inst), s)
            np.testing.assert_array_equal(a[:, Pd:Pd + s, Pd:Pd + s, :5], want.astype(np.float32))
            assert np.all(a[:, :, :, 5:] == 0)                       # band padding
            halo = a.copy()
            halo[:, Pd:Pd + s, Pd:Pd + s] = 0
            assert np.all(halo == 0)
            M = len(inst) * s * s
            np.testing.assert_array_equal(net.labels[:M].cpu().numpy().reshape(-1, s, s), g["%s_c_%d" % (key, s)])
            np.testing.assert_array_equal(net.acc_mask[:M].cpu().numpy().reshape(-1, s, s).astype(bool), g["%s_m_%d" % (key, s)])


def test_crop_float32_pool_and_device_noise():
    from drs_amd import patches as P
    rng = np.random.default_rng(0)
    tiles = [rng.uniform(size=(70, 90, 4))]
    labs = [rng.integers(0, 6, size=(70, 90))]
    inst = np.array([[0, 3, 5, 10], [0, 60, 80, 77], [0, 20, 20, 180]])
    S = 16
    net = _net(4, 3, S)
    pool = P.TilePool(tiles, labs, DEV, dtype=np.float32)
    mean, std = [0.5, 0.5, 0.5], [0.2, 0.2, 0.2]
    P.crop_to_net(net, pool, inst, S, mean, std)
    a, Pd, ld = _slab(net, 3, S)
    want, _, _ = H.dynamically_create_patches([tiles[0].astype(np.float32).astype(np.float64)], labs, inst, S, is_train=False)
    H.normalize_images(want, mean + [0], std + [1])
    np.testing.assert_array_equal(a[:, Pd:Pd + S, Pd:Pd + S, :4], want.astype(np.float32))
    # device-side N(0, 0.01) noise: right moments, no exact reproduction of numpy's stream
    aug = P.Augmentation(3)
    aug.noise_on[:] = 1
    aug.seed = 12345
    P.crop_to_net(net, pool, inst, S, [0, 0, 0], [1, 1, 1], aug)
    b, _, _ = _slab(net, 3, S)
    raw, _, _ = H.dynamically_create_patches([tiles[0].astype(np.float32).astype(np.float64)], labs, inst, S, is_train=False)
    d = b[:, Pd:Pd + S, Pd:Pd + S, :4].astype(np.float64) - raw
    assert abs(d.mean()) < 1e-3 and abs(d.std() - 0.01) < 1e-3


@pytest.mark.parametrize("h,w,S,K,bs", [(70, 90, 25, 6, 16), (64, 64, 16, 2, 7), (50, 131, 20, 7, 64), (33, 40, 33, 6, 3)])
def test_stitch_matches_reference_order(h, w, S, K, bs):
    from drs_amd import _lib, patches as P
    rng = np.random.default_rng(h + w)
    st = H.stride_for(S)
    nh, nw = P.window_counts(h, w, S, st)
    assert (nh, nw) == H.window_counts(h, w, S, st)
    tile = np.zeros((h, w, 1))
    lab = np.zeros((h, w), dtype=np.uint8)
    prob = torch.zeros(h * w * K, dtype=torch.float32, device=DEV)
    occ = torch.zeros(h * w, dtype=torch.int32, device=DEV)
    batches = []
    nb = -(-nh * nw // bs)
    for i in range(nb):
        _, _, pos = H.create_patches_per_map(tile, lab, S, st, i, bs)
        lg = rng.normal(size=(len(pos), S, S, K)).astype(np.float32)
        batches.append((lg, pos))
        np.testing.assert_array_equal(P.window_positions(h, w, S, st, i, bs), np.asarray(pos).astype(np.int64))
        lgd = dev(lg)
        _lib.call("drs_stitch_accumulate", prob.data_ptr(), occ.data_ptr(), lgd.data_ptr(), h, w, K, S, st, i * bs, len(pos), stream())
        torch.cuda.synchronize()
    out = torch.zeros(h * w, dtype=torch.uint8, device=DEV)
    _lib.call("drs_stitch_finalize", prob.data_ptr(), occ.data_ptr(), h, w, K, out.data_ptr(), stream())
    torch.cuda.synchronize()
    p_ref, o_ref, am_ref = H.stitch_tile(h, w, K, S, batches)
    np.testing.assert_array_equal(prob.cpu().numpy().reshape(h, w, K), p_ref)      # same addition order -> bit-exact
    np.testing.assert_array_equal(occ.cpu().numpy().reshape(h, w), o_ref[:, :, 0])
    np.testing.assert_array_equal(out.cpu().numpy().reshape(h, w), am_ref)


# ---- the coffee / contest flavours of the sampler on the device, against goldens made by the reference's own functions
def _indexed_on_device(g, S, C, K, dist, data_tiles, lab_tiles, void_label=-1, quantize=False, mean=(0, 0, 0), std=(1, 1, 1)):
    from drs_amd import patches as P
    from drs_amd.net import DilatedNet
    n = len(dist)
    shuffle = g["shuffle"]
    B = len(shuffle)
    d = DilatedNet("dilated_icpr_rate6_small", C, K, 0.0, b_max=B, s_max=S, device=DEV)
    pool = P.TilePool(data_tiles, lab_tiles, DEV, dtype=np.float32)
    rows = np.asarray(dist, dtype=np.int64)[shuffle % n]
    aug = P.Augmentation(B)
    aug.flip = np.where(shuffle >= 2 * n, 1, np.where(shuffle >= n, 2, 0)).astype(np.int32)       # as loops_indexed.train
    P.crop_to_net(d, pool, rows, S, mean, std, aug, void_label=void_label, quantize_f16=quantize)
    torch.cuda.synchronize()
    slab, Pd, ld = d.input_slab()
    x = slab.view(B, S + 2 * Pd, S + 2 * Pd, ld)[:, Pd:Pd + S, Pd:Pd + S, :C].cpu().numpy()
    M = B * S * S
    return x, d.labels[:M].cpu().numpy().reshape(B, S, S), d.acc_mask[:M].cpu().numpy().reshape(B, S, S).astype(bool)


@pytest.mark.parametrize("S", [9, 13])
def test_coffee_sampler_flip_by_index_and_float16_on_device(S):
    g = np.load(os.path.join(GOLDEN, "coffee.npz"))
    dist = [(int(k), int(i), int(j)) for k, i, j in g["dist"]]
    tiles = [g["data"][0], g["data"][1]]
    labs = [g["mask"][0, :, :, 0].astype(np.uint8), g["mask"][1, :, :, 0].astype(np.uint8)]
    x, lab, _ = _indexed_on_device(g, S, 3, 2, dist, tiles, labs)
    np.testing.assert_array_equal(x, _ref_patches(g, S))                              # float32 path: the plain crop + flip
    np.testing.assert_array_equal(lab, g["classes_%d" % S][..., 0])
    xq, _, _ = _indexed_on_device(g, S, 3, 2, dist, tiles, labs, quantize=True)
    np.testing.assert_array_equal(xq, g["patches16_%d" % S].astype(np.float32))       # coffee:293
    xn, _, _ = _indexed_on_device(g, S, 3, 2, dist, tiles, labs, quantize=True, mean=g["mean"], std=g["std"])
    np.testing.assert_array_equal(xn, g["normalized16_%d" % S].astype(np.float32))    # coffee:67-74 on the float16 array, bit for bit
    # statistics that arrive as float64 (e.g. read back from a float64 .npy): NumPy >= 2 then evaluates float16 (op) float64 in float64
    m64, s64 = g["mean"].astype(np.float64) * 1.0000001, g["std"].astype(np.float64) * 0.9999999
    want = g["patches16_%d" % S].copy()
    for c in range(3):                                                                # coffee:67-74 with float64 scalars, by NumPy itself
        want[..., c] = np.subtract(want[..., c], m64[c])
        want[..., c] = np.divide(want[..., c], s64[c])
    assert want.dtype == np.float16
    xn64, _, _ = _indexed_on_device(g, S, 3, 2, dist, tiles, labs, quantize=True, mean=m64, std=s64)
    np.testing.assert_array_equal(xn64, want.astype(np.float32))


def _ref_patches(g, S):
    from oracle import host_ref as H
    dist = [(int(k), int(i), int(j)) for k, i, j in g["dist"]]
    p, _, _ = H.indexed_create_patches(g["data"], g["mask"][..., 0], S, dist, g["shuffle"])
    return p.astype(np.float32)


@pytest.mark.parametrize("S", [9, 14])
def test_contest_sampler_with_void_mask_on_device(S):
    g = np.load(os.path.join(GOLDEN, "contest.npz"))
    dist = [(0, int(i), int(j)) for i, j in g["dist"]]
    x, lab, mask = _indexed_on_device(g, S, 3, 7, dist, [g["data"]], [g["lab"].astype(np.uint8)], void_label=7)
    np.testing.assert_array_equal(x, g["patches_%d" % S])
    np.testing.assert_array_equal(lab, g["classes_%d" % S])
    np.testing.assert_array_equal(mask, g["masks_%d" % S])


@pytest.mark.parametrize("S", [15, 16, 19, 40])
def test_every_rotation_angle_matches_scipy(S):
    """isprs:294-296 rotates patch, labels and an all-ones mask with scipy.ndimage.rotate(angle, order=0, reshape=False), the angle any
    integer in [0, 360) (isprs:491).  At multiples of 15 / 45 degrees a source coordinate falls exactly between two pixels at some
    sides, and which neighbour wins depends on every product and sum being rounded on its own, as in ndimage's C: a fused
    multiply-add picked the other one (round 3: 23 of 3960 (side, angle) pairs, found by tests/fuzz/fuzz_patches.py; csrc/patches.hip now
    compiles with contraction off).  All 360 angles here, source pixel, label and mask of every output pixel."""
    from scipy import ndimage
    from drs_amd import patches as P
    h = w = S + 6
    tile = (np.arange(h * w, dtype=np.float64) + 1).reshape(h, w, 1).repeat(3, axis=2)
    lab = (np.arange(h * w) % 6).reshape(h, w)
    pool = P.TilePool([tile], [lab], DEV, dtype=np.float64)
    B = 90
    net = _net(3, B, S)
    patch, plab = tile[3:3 + S, 2:2 + S, 0], lab[3:3 + S, 2:2 + S]
    for a0 in range(0, 360, B):
        inst = np.array([[0, 3, 2, a0 + k] for k in range(B)])
        aug = P.Augmentation(B)
        aug.rot_on[:] = 1
        for k in range(B):
            aug.rot[k] = P.rotation_params(a0 + k, S)
        P.crop_to_net(net, pool, inst, S, [0, 0, 0], [1, 1, 1], aug)
        torch.cuda.synchronize()
        a, Pd, ld = _slab(net, B, S)
        got = a[:, Pd:Pd + S, Pd:Pd + S, 0]
        labs = net.labels[:B * S * S].cpu().numpy().reshape(B, S, S)
        mask = net.acc_mask[:B * S * S].cpu().numpy().reshape(B, S, S).astype(bool)
        for k in range(B):
            ang = a0 + k
            np.testing.assert_array_equal(got[k], ndimage.rotate(patch, ang, order=0, reshape=False).astype(np.float32), err_msg="angle %d" % ang)
            np.testing.assert_array_equal(labs[k], ndimage.rotate(plab, ang, order=0, reshape=False), err_msg="angle %d" % ang)
            np.testing.assert_array_equal(mask[k], ndimage.rotate(np.ones((S, S), dtype=bool), ang, order=0, reshape=False), err_msg="angle %d" % ang)
