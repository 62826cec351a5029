"""Test infrastructure: writes a Torch-ASCII tensor dump in the layout the reference's `load_imgs_torch` parses
(coffee_dilated_random.py:84-113: 17 header lines, line 7 = 'c h w', then all c*h*w values on ONE line), from a seed, so that
the golden generator and the tests read byte-identical files without a multi-megabyte fixture in the repository."""
import numpy as np


def values_for(seed, c, h, w, is_mask):
    rng = np.random.default_rng(seed)
    if is_mask:
        return np.round(rng.uniform(0.0, 1.0, size=(c, h, w)), 2)          # the reader rounds half up: floor(v + 0.5)
    return np.round(rng.uniform(0.0, 1.0, size=(c, h, w)), 4)


def write_torch_ascii(path, seed, c, h, w, is_mask=False):
    v = values_for(seed, c, h, w, is_mask)
    with open(path, "w") as fh:
        for i in range(17):
            fh.write("%d %d %d\n" % (c, h, w) if i == 7 else "header line %d\n" % i)
        fh.write(" ".join(("%.2f" if is_mask else "%.4f") % x for x in v.reshape(-1)))
    return v
