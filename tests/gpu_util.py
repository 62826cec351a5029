"""Helpers for the -m gpu parity tests (device views <-> numpy NHWC)."""
import numpy as np
import torch

DEV = "cuda:0"


def stream():
    return torch.cuda.current_stream(DEV).cuda_stream


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def padded(x, P, ld=None, coff=0, fill=0.0):
    """numpy [B,S,S,C] -> device slab [B,S+2P,S+2P,ld] with x in channels coff.. and `fill` elsewhere in the
    interior's other channels / halo (fill != 0 lets a test see a kernel that forgets to zero the halo)."""
    B, S, _, C = x.shape
    ld = ld or C
    buf = np.full((B, S + 2 * P, S + 2 * P, ld), fill, dtype=np.float32)
    buf[:, :, :, coff:coff + C] = 0.0
    buf[:, P:P + S, P:P + S, coff:coff + C] = x
    return dev(buf)


def unpad(t, B, S, P, ld, coff, C):
    a = t.detach().cpu().numpy().reshape(B, S + 2 * P, S + 2 * P, ld)
    return a[:, P:P + S, P:P + S, coff:coff + C], a


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
