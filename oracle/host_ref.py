"""Restatement of the reference's numpy host helpers on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status: PINNED against
tests/golden/*.npz, which were produced by importing the reference's own functions
(tests/golden/make_goldens.py) -- see tests/test_oracle_host.py -- and against the reference's functions
themselves on random inputs (tests/fuzz/fuzz_vs_reference.py, a differential fuzz that runs in the build container only;
tests/test_reference_differential.py is its short seeded form).

Citations are to /root/reference/isprs_dilated_random.py unless another file is named.
RNG: the reference draws from the *global* ``random`` / ``numpy.random`` streams; the
functions here do the same, in the same call order, so that seeding both reproduces it.
"""
import math
import random

import numpy as np
import scipy.ndimage


def select_batch(shuffle, batch_size, it, total_size):
    """isprs:46-58.  Walk a permutation; when the walk reaches the end, draw a NEW
    permutation and top the short batch up from its head (so an index can repeat)."""
    end = min(it + batch_size, total_size)
    batch = shuffle[it:end]
    if end == total_size:
        shuffle = np.asarray(random.sample(range(total_size), total_size))
        it = 0
        short = batch_size - len(batch)
        if short > 0:
            batch = np.concatenate((batch, shuffle[:short]))
            it = short
    else:
        it += batch_size
    return shuffle, batch, it


def define_multinomial_probs(values, dif_prob=2):
    """isprs:61-71.  Every integer size in [v0, v_last]; listed sizes get dif_prob/interval."""
    n = values[-1] - values[0] + 1
    hi = dif_prob / float(n)
    probs = np.full(n, (1.0 - hi * len(values)) / float(n - len(values)))
    for v in values:
        probs[v - values[0]] = hi
    return probs


def draw_patch_size(distribution_type, values, probs=None):
    """isprs:1727-1737.  Returns (cur_patch_size, cur_size_int)."""
    if distribution_type == "multi_fixed":
        i = np.random.randint(len(values))
        return int(values[i]), i
    if distribution_type == "uniform":
        s = int(np.random.uniform(values[0], values[-1] + 1, 1)[0])
        return s, s - values[0]
    if distribution_type == "multinomial":
        i = int(np.random.multinomial(1, probs).argmax())
        return values[0] + i, i
    return int(values[0]), None


def normalize_images(data, mean_full, std_full):
    """isprs:74-81: in place, channels 0,1,2 ONLY (subtract, then divide)."""
    for c in range(3):
        data[:, :, :, c] = np.subtract(data[:, :, :, c], mean_full[c])
    for c in range(3):
        data[:, :, :, c] = np.divide(data[:, :, :, c], std_full[c])


def _shift_inside(x, y, s, h, w):
    """isprs:260-269 / 366-375: a window clipped by the bottom/right border is moved
    back so that it ends at the border."""
    return min(x, h - s), min(y, w - s)


def dynamically_create_patches(data, mask_data, instances, crop_size, is_train=True):
    """isprs:245-334.  instances rows are (map, x, y[, rot]); x indexes rows.
    Train-time augmentation, in this RNG order per patch: randint(0,2) -> rotate by the
    instance angle (nearest, zero fill; patch, labels and an all-ones validity mask);
    randint(0,2) -> + N(0, 0.01) on every channel; randint(0,3) -> none/flipud/fliplr."""
    s = crop_size
    patches, classes, masks = [], [], []
    for inst in instances:
        m, x, y = int(inst[0]), int(inst[1]), int(inst[2])
        h, w = data[m].shape[0], data[m].shape[1]
        x, y = _shift_inside(x, y, s, h, w)
        patch = data[m][x:x + s, y:y + s, :]
        lab = mask_data[m][x:x + s, y:y + s]
        valid = np.ones((s, s), dtype=bool)
        if is_train:
            rot = inst[3]
            if np.random.randint(0, 2) == 1:
                patch = scipy.ndimage.rotate(patch, rot, order=0, reshape=False)
                lab = scipy.ndimage.rotate(lab, rot, order=0, reshape=False)
                valid = scipy.ndimage.rotate(valid, rot, order=0, reshape=False)
            if np.random.randint(0, 2) == 1:
                patch = patch + np.random.normal(0, 0.01, patch.shape)
            flip = np.random.randint(0, 3)
            if flip == 1:
                patch, lab, valid = np.flipud(patch), np.flipud(lab), np.flipud(valid)
            elif flip == 2:
                patch, lab, valid = np.fliplr(patch), np.fliplr(lab), np.fliplr(valid)
        patches.append(patch)
        classes.append(lab)
        masks.append(valid)
    return np.asarray(patches), np.asarray(classes, dtype=int), np.asarray(masks, dtype=bool)


def window_counts(h, w, crop_size, stride):
    """isprs:344-347 / 1253-1256."""
    def n(d):
        q, r = divmod(d - crop_size, stride)
        return q + 1 if r == 0 else q + 2
    return n(h), n(w)


def create_patches_per_map(data, mask_data, crop_size, stride_crop, index, batch_size):
    """isprs:337-400.  Windows in row-major order starting at flat index*batch_size;
    returns (patches, int8 label patches, list of float (x, y) positions)."""
    h, w = data.shape[0], data.shape[1]
    n_h, n_w = window_counts(h, w, crop_size, stride_crop)
    patches, classes, pos = [], [], []
    f = index * batch_size
    while f < n_h * n_w and len(patches) < batch_size:
        r, c = divmod(f, n_w)
        x, y = _shift_inside(r * stride_crop, c * stride_crop, crop_size, h, w)
        patches.append(data[x:x + crop_size, y:y + crop_size, :])
        classes.append(mask_data[x:x + crop_size, y:y + crop_size])
        pos.append(np.array([float(x), float(y)]))
        f += 1
    return np.asarray(patches), np.asarray(classes, dtype=np.int8), pos


def calc_accuracy_by_crop(true_crop, pred_crop, track_conf_matrix, masks=None, num_classes=6):
    """isprs:510-531 (vectorised; the per-pixel loop form is ``calc_accuracy_by_crop_loop``).
    Adds into track_conf_matrix in place; returns (acc, acc_norm, local uint32 matrix)."""
    t = np.asarray(true_crop).reshape(-1).astype(np.int64)
    p = np.asarray(pred_crop).reshape(-1).astype(np.int64)
    if masks is not None:
        keep = np.asarray(masks).reshape(-1).astype(bool)
        t, p = t[keep], p[keep]
    local = np.bincount(t * num_classes + p, minlength=num_classes * num_classes) \
        .reshape(num_classes, num_classes).astype(np.uint32)
    track_conf_matrix += local.astype(track_conf_matrix.dtype)
    acc = int(np.trace(local))
    rows = local.sum(axis=1).astype(np.float64)
    rec = np.where(rows != 0, np.diag(local) / np.where(rows != 0, rows, 1), 0.0)
    return acc, float(rec.sum() / num_classes), local


def calc_accuracy_by_crop_loop(true_crop, pred_crop, track_conf_matrix, masks=None, num_classes=6):
    """isprs:510-531 in its original per-pixel form (what the CPU baseline times)."""
    b, h, w = pred_crop.shape
    acc = 0
    local = np.zeros((num_classes, num_classes), dtype=np.uint32)
    for i in range(b):
        for j in range(h):
            for k in range(w):
                if masks is None or masks[i, j, k]:
                    t, p = true_crop[i, j, k], pred_crop[i, j, k]
                    if t == p:
                        acc += 1
                    track_conf_matrix[t][p] += 1
                    local[t][p] += 1
    tot = 0.0
    for i in range(num_classes):
        rs = np.sum(local[i])
        tot += local[i][i] / float(rs) if rs != 0 else 0
    return acc, tot / float(num_classes), local


def select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, is_loss_or_acc="acc",
                           patch_chosen_values=None):
    """isprs:549-608.  NB mutates patch_occur (zeros -> 1), so an unsampled size has mean
    0 and wins under 'loss'."""
    patch_occur[np.where(patch_occur == 0)] = 1
    mean = patch_acc_loss / patch_occur
    if is_loss_or_acc == "acc":
        i = int(np.argmax(mean))
    else:
        order = np.argsort(mean)
        i = int(next(j for j in order if patch_occur[j] > 0))
    if patch_chosen_values is not None:
        patch_chosen_values[i] += 1
    if distribution_type == "multi_fixed":
        return int(values[i])
    return values[0] + i


def stitch_tile(h, w, num_classes, crop_size, batches):
    """isprs:1261-1284.  ``batches`` yields (logits[b,s,s,K] float32, pos list); windows
    are accumulated in order; the averaged quantity is the raw logits."""
    prob = np.zeros([h, w, num_classes], dtype=np.float32)
    occur = np.zeros([h, w, num_classes], dtype=np.uint32)
    for logits, pos in batches:
        for j in range(len(logits)):
            x, y = int(pos[j][0]), int(pos[j][1])
            prob[x:x + crop_size, y:y + crop_size, :] += logits[j]
            occur[x:x + crop_size, y:y + crop_size, :] += 1
    occur[np.where(occur == 0)] = 1
    return prob, occur, np.argmax(prob / occur.astype(float), axis=2)


def stride_for(crop_size):
    """isprs:1243."""
    return int(math.floor(crop_size / 2.0))


def softmax_lastaxis(array):
    """isprs:38-43 (no max subtraction; dtype of the input)."""
    expa = np.exp(array)
    return expa / np.sum(expa, axis=-1, keepdims=True)


def multiscale_argmax(mean_logit_maps):
    """isprs:1416-1424: per-scale softmax of the averaged logits (float32), summed over scales, arg-max."""
    mean_prob = np.zeros((len(mean_logit_maps),) + mean_logit_maps[0].shape, dtype=np.float32)
    for i, m in enumerate(mean_logit_maps):
        mean_prob[i] = m
        mean_prob[i] = softmax_lastaxis(mean_prob[i])
    return np.argmax(np.sum(mean_prob, axis=0), axis=2)


def indexed_create_patches(data, mask_data, crop_size, class_distribution, shuffle, float16=False, void_label=None):
    """coffee_dilated_random.py:241-293 / contest_dilated_random.py:192-254: index i of the 3N-long permutation selects window
    i mod N and the flip ([0,N) as is, [N,2N) left-right, [2N,3N) up-down); a window clipped by the border moves back to end at
    it; coffee returns float16 patches (:293); contest also returns the void mask (label 7 -> False, :235-239).
    data [n,H,W,C], mask_data [n,H,W]; class_distribution rows (map, x, y)."""
    n = len(class_distribution)
    patches, classes, masks = [], [], []
    for i in shuffle:
        k, x, y = class_distribution[int(i) % n]
        h, w = data[k].shape[0], data[k].shape[1]
        x, y = min(x, h - crop_size), min(y, w - crop_size)
        p = data[k][x:x + crop_size, y:y + crop_size, :]
        c = mask_data[k][x:x + crop_size, y:y + crop_size]
        m = (c != void_label) if void_label is not None else np.ones(c.shape, dtype=bool)
        if n <= i < 2 * n:
            p, c, m = np.fliplr(p), np.fliplr(c), np.fliplr(m)
        elif i >= 2 * n:
            p, c, m = np.flipud(p), np.flipud(c), np.flipud(m)
        patches.append(p)
        classes.append(c)
        masks.append(m)
    return (np.asarray(patches, dtype=np.float16 if float16 else None), np.asarray(classes, dtype=np.int8), np.asarray(masks, dtype=bool))


def normalize_images_f16(patches16, mean_full, std_full):
    """coffee:67-74 applied to the float16 patches of coffee:293, as numpy >= 2 evaluates it (the goldens were made with 2.2):
    float16 array (op) float32 scalar runs in float32 and the assignment into the float16 array rounds -- so every value is
    rounded to float16 three times: by the cast, after the subtraction and after the division.  (numpy 1.x cast the scalar to
    float16 first; results can differ in the last float16 bit.)"""
    out = patches16.copy()
    for ch in range(3):
        out[..., ch] = np.subtract(out[..., ch], np.float32(mean_full[ch]))
        out[..., ch] = np.divide(out[..., ch], np.float32(std_full[ch]))
    return out
