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


def conv_stats_moments(lib, stats, M, mt, cout):
    """statistics slab of a convolution epilogue (per tile (sum, M2 about the tile mean)) -> numpy [cout, 2] = (sum z, sum z^2),
    through drs_conv_stats_reduce (Chan combination in fp64)."""
    sums = torch.zeros(2 * cout, dtype=torch.float64, device=DEV)
    lib.call("drs_conv_stats_reduce", stats.data_ptr(), M, mt, cout, sums.data_ptr(), None, stream())
    torch.cuda.synchronize()
    return sums.cpu().numpy().reshape(cout, 2)
