"""One-off host preprocessing that feeds the hot path: class-balanced window lists, per-instance
rotation angles, the class-balanced super-batch and the dataset mean/std.

Host mirror of /root/reference/isprs_dilated_random.py: `create_distributions_over_classes` :448-483,
`create_rotation_distribution` :486-496, `select_super_batch_instances` :403-445,
`dynamically_calculate_mean_and_std` :151-184 with `compute_image_mean` :84-88.  Same outputs and the
same RNG call order as the reference (pinned by tests/golden/sampling.npz); the window scan is
vectorised with per-class integral images instead of a bincount per window.
"""
import random

import numpy as np


def create_distributions_over_classes(labels, crop_size, stride_crop, num_classes=6):
    """Per class, the list of (map, x, y) windows whose majority label is that class; windows start on
    a `stride_crop` grid and are shifted back to end at the border (isprs:461-470)."""
    classes = [[] for _ in range(num_classes)]
    for k, lab in enumerate(labels):
        h, w = lab.shape
        xs = np.minimum(np.arange(0, h, stride_crop), h - crop_size)
        ys = np.minimum(np.arange(0, w, stride_crop), w - crop_size)
        nmax = max(num_classes, int(lab.max()) + 1)
        counts = np.zeros((nmax, len(xs), len(ys)), dtype=np.int64)
        for c in range(nmax):
            ii = np.zeros((h + 1, w + 1), dtype=np.int64)
            ii[1:, 1:] = np.cumsum(np.cumsum(lab == c, axis=0), axis=1)
            x0, y0 = xs[:, None], ys[None, :]
            counts[c] = ii[x0 + crop_size, y0 + crop_size] - ii[x0, y0 + crop_size] - ii[x0 + crop_size, y0] + ii[x0, y0]
        major = np.argmax(counts, axis=0)            # first maximum, as np.argmax(np.bincount(...))
        for a, x in enumerate(xs):
            for b, y in enumerate(ys):
                classes[int(major[a, b])].append((k, int(x), int(y)))
    return classes


def create_rotation_distribution(class_distribution):
    """isprs:486-496: one integer angle in [0, 360) per training window."""
    return [np.random.randint(0, 360, size=len(c)) for c in class_distribution]


def select_super_batch_instances(class_distribution, rotation_distribution=None, batch_size=100, super_batch=500):
    """isprs:403-445: batch_size*super_batch instances, an equal share per class drawn without
    replacement, the shortfall filled with uniformly random (class, window) draws.  Rows (map, x, y, rot)."""
    want = batch_size * super_batch
    per_class = int(want / len(class_distribution))
    inst = []
    for i, wins in enumerate(class_distribution):
        for j in random.sample(range(len(wins)), min(per_class, len(wins))):
            rot = rotation_distribution[i][j] if rotation_distribution is not None else 0
            inst.append((wins[j][0], wins[j][1], wins[j][2], rot))
    while len(inst) < want:
        i = np.random.randint(len(class_distribution))
        j = np.random.randint(len(class_distribution[i]))
        w = class_distribution[i][j]
        rot = rotation_distribution[i][j] if rotation_distribution is not None else 0
        inst.append((w[0], w[1], w[2], rot))
    assert len(inst) == want, "Could not select ALL instances"
    return np.asarray(inst)


def dynamically_calculate_mean_and_std(data, indexes, crop_size):
    """isprs:151-184 / 84-88: chunks of ~5000 windows; per chunk the per-band mean over everything and
    the across-window standard deviation (ddof=1) AT PIXEL (0, 0); both averaged over the chunks."""
    total = [w for cls in indexes for w in cls]
    means, stds, chunk = [], [], []

    def flush():
        arr = np.asarray(chunk)
        means.append(np.mean(np.mean(np.mean(arr, axis=0), axis=0), axis=0) if arr.size else np.full(data[0].shape[2], np.nan))
        stds.append(np.std(arr[:, 0, 0, :], axis=0, ddof=1) if arr.size else np.full(data[0].shape[2], np.nan))

    for i, (m, x, y) in enumerate(total):
        chunk.append(data[m][x:x + crop_size, y:y + crop_size, :])
        if i > 0 and i % 5000 == 0:
            flush()
            chunk = []
    flush()
    return np.mean(means, axis=0), np.mean(stds, axis=0)
