"""Synthetic multispectral tiles for benchmarks and tests (there is no dataset access): SURVEY.md 8(d).

Bands 0-2 are low-pass filtered uniform noise (image-like), band 3 plain uniform (IR), band 4 a damped
uniform (nDSM); labels are a Voronoi-style map so that every class is present and regions span several
patch sizes.
"""
import numpy as np
import scipy.ndimage
from scipy.spatial import cKDTree


def make_tile(h, w, channels=5, num_classes=6, seed=1234, n_seeds=400, dtype=np.float64, class_signal=0.0, signal_seed=99):
    """class_signal > 0 adds a per-class offset to every band (the same offsets for every tile that shares
    `signal_seed`), so that the labels can be learnt from the pixels: accuracy-parity runs need that."""
    rng = np.random.default_rng(seed)
    img = rng.uniform(0.0, 1.0, size=(h, w, channels))
    for c in range(min(3, channels)):
        img[:, :, c] = scipy.ndimage.uniform_filter(img[:, :, c], size=9, mode="reflect")
    if channels > 4:
        img[:, :, 4] *= 0.2
    # Voronoi labels on a coarse grid (<= 1024 a side), repeated up to full resolution
    f = max(1, int(np.ceil(max(h, w) / 1024.0)))
    hc, wc = -(-h // f), -(-w // f)
    pts = np.stack([rng.integers(0, hc, size=n_seeds), rng.integers(0, wc, size=n_seeds)], axis=1)
    cls = np.arange(n_seeds) % num_classes
    yy, xx = np.meshgrid(np.arange(hc), np.arange(wc), indexing="ij")
    _, nearest = cKDTree(pts).query(np.stack([yy.ravel(), xx.ravel()], axis=1))
    lab = cls[nearest].reshape(hc, wc).astype(np.uint8)
    lab = np.repeat(np.repeat(lab, f, axis=0), f, axis=1)[:h, :w]
    if class_signal:
        means = np.random.default_rng(signal_seed).uniform(-1.0, 1.0, size=(num_classes, channels))
        img = np.clip(img + class_signal * means[lab], 0.0, 1.0)
    return img.astype(dtype), np.ascontiguousarray(lab)


def grid_instances(h, w, crop, stride, n, seed=0, map_index=0):
    """n window origins on a stride grid (shift-back at the border), with random rotation angles."""
    rng = np.random.default_rng(seed)
    xs = np.minimum(np.arange(0, h, stride), h - crop)
    ys = np.minimum(np.arange(0, w, stride), w - crop)
    ix = rng.integers(0, len(xs), size=n)
    iy = rng.integers(0, len(ys), size=n)
    return np.stack([np.full(n, map_index), xs[ix], ys[iy], rng.integers(0, 360, size=n)], axis=1).astype(np.int64)
