"""CPU: the error model of the split-bf16 arithmetic in plain numpy / torch-CPU, independent of the kernels:
term splitting (round-to-nearest-even bf16 of what the earlier terms left) and the partial products i + j < NS summed in
fp32, against fp64.  The -m gpu tests hold the HIP kernels to the fp64 oracle; this one documents why the bounds hold."""
import numpy as np
import torch


def split(x, ns):
    r = x.astype(np.float32).copy()
    out = []
    for _ in range(ns):
        t = torch.from_numpy(r).to(torch.bfloat16).to(torch.float32).numpy()
        out.append(t)
        r = r - t                      # exact in fp32
    return out


def test_terms_reconstruct_to_the_stated_residual():
    rng = np.random.default_rng(0)
    x = (rng.normal(size=20000) * np.exp(rng.normal(size=20000) * 3)).astype(np.float32)
    for ns, bound in ((2, 2.0 ** -17), (3, 2.0 ** -25)):
        t = split(x, ns)
        resid = np.abs(x.astype(np.float64) - sum(v.astype(np.float64) for v in t))
        assert np.all(resid <= bound * np.abs(x).astype(np.float64))
    # three 8-bit significands cover the 24 bits of an fp32 exactly in the typical case
    t = split(x, 3)
    assert np.mean(sum(v.astype(np.float64) for v in t) == x.astype(np.float64)) > 0.99


def test_partial_product_sums_against_fp64():
    rng = np.random.default_rng(1)
    K = 2304                                     # conv8's contraction length (3*3*256)
    a = rng.normal(size=(64, K)).astype(np.float32)
    b = (rng.normal(size=(K, 64)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(ref).max()
    errs = {}
    for ns in (2, 3):
        ta, tb = split(a, ns), split(b, ns)
        acc = np.zeros((64, 64), dtype=np.float32)
        for d in range(ns - 1, -1, -1):          # smallest partial products first, as the kernels issue them
            for i in range(d + 1):
                # each bf16 x bf16 product is exact in fp32; the sum over K is an fp32 accumulation (order differs from the MFMA's)
                acc = acc + (ta[i].astype(np.float32) @ tb[d - i].astype(np.float32))
        errs[ns] = np.abs(acc.astype(np.float64) - ref).max() / scale
    f32 = np.abs((a @ b).astype(np.float64) - ref).max() / scale
    assert errs[2] < 1e-5 and errs[3] < 2e-6     # the bounds test_gpu_split.py holds the kernels to are 1e-5 for both
    assert errs[3] < 4 * max(f32, 1e-7)          # three terms: the same order as a plain fp32 product-sum
