"""Patch sampling (host) and patch materialisation (device).

Host mirror of /root/reference/isprs_dilated_random.py: `select_batch` :46-58, the patch-size draw
:1727-1737 with `define_multinomial_probs` :61-71, the per-patch augmentation draws inside
`dynamically_create_patches` :288-318 and the sliding-window enumeration of `create_patches_per_map`
:337-400.  Index work stays on the host (it is scalar work); every per-pixel operation -- crop,
rotation, noise, flip, normalisation of bands 0..2, zero halo and band padding for conv1 -- happens
in one HIP kernel (drs_crop_normalize) that writes the conv1 input slab directly.

RNG: like the reference, draws come from the global `random` / `numpy.random` streams in the
reference's call order, so seeding both reproduces the reference's sequence.
"""
import math
import random

import numpy as np
import torch
from scipy import special

from . import _lib


# ---------------------------------------------------------------------------------------- index sampling
def select_batch(shuffle, batch_size, it, total_size):
    """isprs:46-58: walk the permutation; at its end reshuffle and top up from the new permutation."""
    end = min(it + batch_size, total_size)
    batch = shuffle[it:end]
    if end != total_size:
        return shuffle, batch, it + batch_size
    shuffle = np.asarray(random.sample(range(total_size), total_size))
    short = batch_size - len(batch)
    if short > 0:
        batch = np.concatenate((batch, shuffle[:short]))
    return shuffle, batch, max(short, 0)


def define_multinomial_probs(values, dif_prob=2):
    """isprs:61-71."""
    n = values[-1] - values[0] + 1
    hi = dif_prob * (1.0 / float(n))
    probs = np.full(n, (1.0 - hi * len(values)) / float(n - len(values)))
    probs[np.asarray(values) - values[0]] = hi
    return probs


def draw_patch_size(distribution_type, values, probs=None):
    """isprs:1727-1737 -> (cur_patch_size, cur_size_int)."""
    if distribution_type == "multi_fixed":
        i = np.random.randint(len(values))
        return int(values[i]), i
    if distribution_type == "uniform":
        s = int(np.random.uniform(values[0], values[-1] + 1, 1)[0])
        return s, s - values[0]
    if distribution_type == "multinomial":
        i = int(np.random.multinomial(1, probs).argmax())
        return values[0] + i, i
    if distribution_type == "single_fixed":
        return int(values[0]), None
    raise ValueError("unknown distribution_type " + str(distribution_type))


def window_counts(h, w, crop_size, stride):
    """isprs:344-347."""
    def n(d):
        q, r = divmod(d - crop_size, stride)
        return q + 1 if r == 0 else q + 2
    return n(h), n(w)


def window_start(h, w, crop_size, stride, index, batch_size, flavour="isprs"):
    """flat (row-major) index of the first window of batch `index`.  isprs:353-354 starts batch i at window i*batch_size.
    contest:275-276 divides by the number of window ROWS where the number of columns belongs
    (`offset_h = int(index*batch_size / total_index_h)`), so on a non-square tile its batches start too early (tall tiles:
    windows evaluated twice, the bottom rows never) or too late: reproduced for flavour="contest" because the reference's label
    maps on such tiles are what they are.  coffee:304-309 takes both counts from the height; its tiles are square (500 x 500),
    where all three agree."""
    n_h, n_w = window_counts(h, w, crop_size, stride)
    f = index * batch_size
    if flavour == "contest":
        return (f // n_h) * n_w + f % n_w
    return f


def window_positions(h, w, crop_size, stride, index, batch_size, flavour="isprs"):
    """isprs:337-400 without the pixel copies: (x, y) of the windows of batch `index` (row-major from window_start), the last
    row / column shifted back to end at the border."""
    n_h, n_w = window_counts(h, w, crop_size, stride)
    f0 = window_start(h, w, crop_size, stride, index, batch_size, flavour)
    f = np.arange(f0, max(f0, min(f0 + batch_size, n_h * n_w)))
    x = np.minimum((f // n_w) * stride, h - crop_size)
    y = np.minimum((f % n_w) * stride, w - crop_size)
    return np.stack([x, y], axis=1).astype(np.int64)


# ---------------------------------------------------------------------------------------- augmentation draws
def rotation_params(angle_deg, S):
    """(m00, m01, m10, m11, off0, off1) that scipy.ndimage.rotate(reshape=False) hands to its
    geometric transform for an S x S plane (output -> input coordinates)."""
    c, s = special.cosdg(angle_deg), special.sindg(angle_deg)
    m = np.array([[c, s], [-s, c]])
    centre = (np.array([S, S]) - 1) / 2
    off = centre - m @ centre
    return np.array([m[0, 0], m[0, 1], m[1, 0], m[1, 1], off[0], off[1]], dtype=np.float64)


def nearest_source_index(params, S):
    """numpy statement of what the kernel evaluates per output pixel: order-0 geometric transform of
    ndimage (in = M.out + off accumulated left to right, nearest = floor(c + 0.5), zero fill outside
    [0, S-1]).  Returns (src_row, src_col, valid) arrays [S, S]."""
    i, j = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64), indexing="ij")
    c0 = ((0.0 + i * params[0]) + j * params[1]) + params[4]
    c1 = ((0.0 + i * params[2]) + j * params[3]) + params[5]
    valid = ~((c0 < 0) | (c0 > S - 1) | (c1 < 0) | (c1 > S - 1))
    return np.floor(c0 + 0.5).astype(np.int64), np.floor(c1 + 0.5).astype(np.int64), valid


class Augmentation(object):
    """Per-batch augmentation decisions, drawn in the reference's order (isprs:288-318)."""

    def __init__(self, B):
        self.rot_on = np.zeros(B, dtype=np.uint8)
        self.rot = np.zeros((B, 6), dtype=np.float64)
        self.noise_on = np.zeros(B, dtype=np.uint8)
        self.flip = np.zeros(B, dtype=np.int32)
        self.noise = None          # [B, S, S, C] float64 when host noise is used
        self.seed = 0
        self.index0 = 0            # place of the first patch in the global batch (device noise is keyed by the global index)


def draw_augmentation(instances, S, C, noise="device"):
    """For every instance, in order: randint(0,2) rotate?; randint(0,2) noise? [+ normal(0, .01, (S,S,C))
    when noise == 'host': bit-exact with the reference]; randint(0,3) flip."""
    B = len(instances)
    aug = Augmentation(B)
    if noise == "host":
        aug.noise = np.zeros((B, S, S, C), dtype=np.float64)
    for b in range(B):
        if np.random.randint(0, 2) == 1:
            aug.rot_on[b] = 1
            aug.rot[b] = rotation_params(instances[b][3], S)
        if np.random.randint(0, 2) == 1:
            aug.noise_on[b] = 1
            if noise == "host":
                aug.noise[b] = np.random.normal(0, 0.01, (S, S, C))
        aug.flip[b] = np.random.randint(0, 3)
    if noise != "host":
        aug.seed = int(np.random.randint(0, 2 ** 31 - 1))
    return aug


# ---------------------------------------------------------------------------------------- device tile pool
class TilePool(object):
    """All tiles (HWC) and label maps (HW) of a split, resident in HBM for the whole run.
    dtype float64 keeps the reference's `img_as_float` precision (bit-exact normalisation); float32
    halves the gather traffic."""

    def __init__(self, tiles, labels, device, dtype=np.float64):
        self.dev = torch.device(device)
        self.C = int(tiles[0].shape[2])
        self.n = len(tiles)
        self.f64 = np.dtype(dtype) == np.float64
        self.h = [int(t.shape[0]) for t in tiles]
        self.w = [int(t.shape[1]) for t in tiles]
        toff = np.cumsum([0] + [t.size for t in tiles])[:-1].astype(np.int64)
        loff = np.cumsum([0] + [t.shape[0] * t.shape[1] for t in tiles])[:-1].astype(np.int64)
        flat = np.concatenate([np.ascontiguousarray(t, dtype=dtype).reshape(-1) for t in tiles])
        if labels is None:
            labels = [np.zeros(t.shape[:2], dtype=np.uint8) for t in tiles]
        lflat = np.concatenate([np.ascontiguousarray(l).astype(np.uint8).reshape(-1) for l in labels])
        self.tiles = torch.from_numpy(flat).to(self.dev)
        self.labels = torch.from_numpy(lflat).to(self.dev)
        self.tile_off = torch.from_numpy(toff).to(self.dev)
        self.lab_off = torch.from_numpy(loff).to(self.dev)
        self.tile_h = torch.tensor(self.h, dtype=torch.int32, device=self.dev)
        self.tile_w = torch.tensor(self.w, dtype=torch.int32, device=self.dev)


def _shift_inside(inst_xy, pool, S):
    """isprs:260-269: a window clipped by the bottom/right border is moved back to end at the border."""
    inst = np.asarray(inst_xy, dtype=np.int64)[:, :3].copy()
    hh = np.asarray(pool.h)[inst[:, 0]]
    ww = np.asarray(pool.w)[inst[:, 0]]
    if np.any(hh < S) or np.any(ww < S):
        raise ValueError("Error: Current PATCH size exceeds the tile")       # reference prints and returns None
    inst[:, 1] = np.minimum(inst[:, 1], hh - S)
    inst[:, 2] = np.minimum(inst[:, 2], ww - S)
    return inst


class _Staging(object):
    """Per-step index / augmentation tables go to the device in ONE asynchronous copy from a pinned ring buffer
    (pageable uploads would block the host every step and keep it from running ahead of the GPU).
    Layout per slot, 8-byte aligned: rot f64 [B][6] | inst i32 [B][4] | rot_on u8 [B] | noise_on u8 [B]."""
    SLOTS = 4

    def __init__(self, dev, b_max):
        self.b_max = b_max
        self.o_rot, self.o_inst = 0, 48 * b_max
        self.o_ron, self.o_non = 64 * b_max, 65 * b_max
        self.nbytes = (66 * b_max + 7) // 8 * 8
        self.host = [torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=torch.cuda.is_available()) for _ in range(self.SLOTS)]
        self.dev = [torch.empty(self.nbytes, dtype=torch.uint8, device=dev) for _ in range(self.SLOTS)]
        self.events = [None] * self.SLOTS
        self.i = 0

    def upload(self, inst, aug):
        k = self.i
        self.i = (k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()              # the copy issued SLOTS steps ago has long finished
        B = len(inst)
        h = self.host[k].numpy()
        h[self.o_inst:self.o_inst + 16 * B].view(np.int32)[:] = inst.reshape(-1)
        if aug is not None:
            h[self.o_rot:self.o_rot + 48 * B].view(np.float64)[:] = aug.rot.reshape(-1)
            h[self.o_ron:self.o_ron + B] = aug.rot_on
            h[self.o_non:self.o_non + B] = aug.noise_on
        self.dev[k].copy_(self.host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        base = self.dev[k].data_ptr()
        return base + self.o_inst, base + self.o_rot, base + self.o_ron, base + self.o_non


def crop_to_net(net, pool, instances, S, mean, std, aug=None, void_label=-1, quantize_f16=False):
    """dynamically_create_patches + normalize_images (isprs:1742-1745 / 1579-1583) fused on the device:
    fills net's conv1 slab, net.labels and net.acc_mask for `instances` rows (map, x, y[, rot]).
    quantize_f16: the coffee script's training patches pass through float16 (coffee:293) and are normalised in place in that
    array (coffee:1290): value, difference and quotient are each rounded to float16; NumPy >= 2 evaluates the difference and the
    quotient in the type of the mean / std scalars when that is wider: float32 for coffee's own statistics (np.mean / np.std of
    float32 patches, coffee:78-79), float64 when `mean` arrives as float64 (drs_crop_normalize modes 1 / 2)."""
    import ctypes as C
    B = len(instances)
    net._check(B, S)
    inst = np.zeros((B, 4), dtype=np.int32)
    inst[:, :3] = _shift_inside(instances, pool, S)
    if aug is not None:
        inst[:, 3] = aug.flip
    stg = getattr(net, "_staging", None)
    if stg is None:
        stg = net._staging = _Staging(net.dev, net.b_max)
    p_inst, p_rot, p_ron, p_non = stg.upload(inst, aug)
    noise = None
    if aug is not None and aug.noise is not None:
        noise = torch.from_numpy(aug.noise).to(net.dev)          # reference-exact host noise (tests / parity runs)
    m = list(np.asarray(mean, dtype=np.float64)[:3]) + [0.0] * max(0, 3 - len(mean))
    sd = list(np.asarray(std, dtype=np.float64)[:3]) + [1.0] * max(0, 3 - len(std))
    m3c, s3c = (C.c_double * 3)(*m), (C.c_double * 3)(*sd)       # HOST pointers: copied into the kernel arguments
    slab, P, ld = net.input_slab()
    _lib.call("drs_crop_normalize", pool.tiles.data_ptr(), 1 if pool.f64 else 0, pool.labels.data_ptr(),
              pool.tile_off.data_ptr(), pool.lab_off.data_ptr(), pool.tile_h.data_ptr(), pool.tile_w.data_ptr(), pool.C,
              p_inst, p_rot if aug is not None else None, p_ron if aug is not None else None,
              None if noise is None else noise.data_ptr(), p_non if aug is not None else None,
              aug.seed if aug is not None else 0, aug.index0 if aug is not None else 0, C.cast(m3c, C.c_void_p), C.cast(s3c, C.c_void_p), B, S, P, ld,
              slab.data_ptr(), net.labels.data_ptr(), net.acc_mask.data_ptr(), int(void_label),
              (2 if getattr(mean, "dtype", None) == np.float64 else 1) if quantize_f16 else 0, net._stream())
    net._keep = noise                                            # alive until the stream has consumed it
    return inst[:, 1:3]


def pack_feed(net, batch_x, batch_y, crop_size, mask=None, acc_mask=None):
    """The reference's feed_dict form (isprs:1746-1752): x float32 [B, s*s*C], y [B, s*s] -> device slab.
    Implemented with the same gather kernel, each patch being its own float32 'tile'."""
    x = np.ascontiguousarray(np.asarray(batch_x, dtype=np.float32))
    B = x.shape[0]
    C_ = net.plan.channels
    S = int(crop_size) if crop_size is not None else int(round(math.sqrt(x.shape[1] // C_)))
    tiles = [x[b].reshape(S, S, C_) for b in range(B)]
    labs = None
    if batch_y is not None:
        labs = [np.asarray(batch_y[b]).reshape(S, S).astype(np.uint8) for b in range(B)]
    pool = TilePool(tiles, labs, net.dev, dtype=np.float32)
    inst = np.stack([np.arange(B), np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.int64)], axis=1)
    crop_to_net(net, pool, inst, S, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    M = B * S * S
    if mask is not None:
        net.loss_mask[:M].copy_(torch.from_numpy(np.asarray(mask).reshape(-1).astype(np.uint8)))
    if acc_mask is not None:
        net.acc_mask[:M].copy_(torch.from_numpy(np.asarray(acc_mask).reshape(-1).astype(np.uint8)))
    torch.cuda.current_stream(net.dev).synchronize()      # the temporary pool dies with this frame
    return B, S
