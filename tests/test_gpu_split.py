"""-m gpu: the split-bf16 arithmetic of the convolution (csrc/conv_split.hip: fp32 operands as 2 or 3 bf16 terms,
3 or 6 partial products on the bf16 MFMA pipe, fp32 accumulation) against the fp64 CPU oracle on the same seeded
inputs, through the C ABI.  Tolerances, relative to the tensor's max magnitude: 3 terms ("bf16x6") 1e-5, the bound
the exact-fp32 kernels are held to in test_gpu_ops.py, for both term counts (BASELINE north_star: logits
within 1e-3 relative end to end)."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from oracle import nets as onets

pytestmark = pytest.mark.gpu

from gpu_util import DEV, conv_stats_moments, dev, padded, rel_err, stream   # noqa: E402

# two terms (bf16x3) drop the t1*u1 partial product: 2^-16 = 1.5e-5 of a product.  Over sums of hundreds of products the errors average
# well below 1e-5 of the tensor's maximum; a sum of a FEW products (a 5 x 5 patch under a rate-6 filter: most taps meet the halo) keeps
# nearly the per-product bound -- tests/fuzz/fuzz_split.py seed 617, case (2, 6, 128, 256, 2, 5, 2): 1.10e-5 (profiles/r06/fuzz.txt).
# Three terms (bf16x6) stay under the exact-fp32 kernels' 1e-5.
TOL = {2: 1.6e-5, 3: 1e-5}


@pytest.fixture(scope="module")
def lib():
    from drs_amd import _lib
    assert torch.cuda.is_available()
    _lib.load()
    return _lib


def split_host(x, ns):
    """numpy restatement of the term split: term s = bf16_rne(x - earlier terms); returns float32 terms."""
    r = x.astype(np.float32).copy()
    out = []
    for _ in range(ns):
        t = torch.from_numpy(r).to(torch.bfloat16).to(torch.float32).numpy()
        out.append(t)
        r = r - t
    return out


def planes(lib, t, ns):
    """device terms of a float32 tensor in the kernels' layout: term s of element e at (e & ~31)*ns + 32*s + (e & 31)"""
    n = t.numel()
    assert n % 32 == 0
    p = torch.zeros(ns * n, dtype=torch.int16, device=DEV)
    lib.call("drs_split_terms", t.data_ptr(), n, ns, p.data_ptr(), stream())
    return p, n


@pytest.mark.parametrize("ns", [2, 3])
def test_split_planes_terms(lib, ns):
    rng = np.random.default_rng(ns)
    x = (rng.normal(size=4096) * np.exp(rng.normal(size=4096) * 4)).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 1e-30, 3.0e38, 1.0 + 2.0 ** -9, 2.0 ** -126]
    p, n = planes(lib, dev(x), ns)
    torch.cuda.synchronize()
    got = p.view(torch.bfloat16).to(torch.float32).cpu().numpy().reshape(n // 32, ns, 32).transpose(1, 0, 2).reshape(ns, n)
    want = split_host(x, ns)
    for s in range(ns):
        assert np.array_equal(got[s], want[s]), s
    resid = np.abs(x.astype(np.float64) - got.astype(np.float64).sum(axis=0))
    bound = np.abs(x.astype(np.float64)) * 2.0 ** (-8 * ns - 1) + 1e-40
    assert np.all(resid <= bound)


# (k, rate, cin, cout, B, S): every (k, rate) of the BASELINE nets, both tile shapes, ragged wgrad row tiles
CASES = [
    (5, 1, 32, 64, 3, 9), (5, 2, 64, 64, 2, 12), (4, 3, 64, 128, 2, 11), (4, 4, 128, 128, 2, 10),
    (3, 5, 128, 192, 2, 13), (3, 6, 192, 192, 1, 15), (3, 7, 192, 256, 1, 16), (3, 8, 256, 256, 2, 17),
    (4, 2, 64, 128, 2, 8), (3, 4, 128, 256, 1, 9), (3, 6, 320, 128, 1, 14), (4, 4, 128, 64, 1, 12), (3, 5, 32, 64, 1, 25),
]


@pytest.mark.parametrize("ns", [2, 3])
@pytest.mark.parametrize("k,rate,cin,cout,B,S", CASES)
def test_conv_split_forward_dgrad_wgrad(lib, k, rate, cin, cout, B, S, ns):
    rng = np.random.default_rng(k * 1000 + rate * 100 + cin + cout + S)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa) + 1
    M = B * S * S
    tol = TOL[ns]
    xd = padded(x, P, ld=cin + 32, coff=32, fill=7.0)
    xp, nx = planes(lib, xd, ns)
    wd, bd = dev(w), dev(bias)
    nw = k * k * cin * cout
    wf = torch.zeros(ns * nw, dtype=torch.int16, device=DEV)
    wg = torch.zeros(ns * nw, dtype=torch.int16, device=DEV)
    lib.call("drs_filter_split", wd.data_ptr(), k, cin, cin, cout, ns, wf.data_ptr(), wg.data_ptr(), stream())
    out = torch.full((M, cout + 32), -3.0, dtype=torch.float32, device=DEV)
    mt = lib.query("drs_split_conv_mtile", cout)
    rows = (M + mt - 1) // mt
    stats = torch.zeros(rows * cout * 2, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward_split", xp.data_ptr(), B, S, P, cin + 32, 32, wf.data_ptr(), bd.data_ptr(), k, rate, pb,
             cin, cout, out.data_ptr(), cout + 32, 32, 0, stats.data_ptr(), ns, stream())
    torch.cuda.synchronize()
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    ref = T.conv2d_same(x64, w64, rate) + bias.astype(np.float64)
    got = out.cpu().numpy()
    e_fwd = rel_err(got[:, 32:].reshape(B, S, S, cout), ref)
    assert e_fwd < tol
    assert np.all(got[:, :32] == -3.0)
    st = conv_stats_moments(lib, stats, M, mt, cout)
    r2 = ref.reshape(-1, cout)
    assert np.abs(st[:, 0] - r2.sum(axis=0)).max() < tol * np.abs(r2).sum(axis=0).max()
    assert rel_err(st[:, 1], (r2 ** 2).sum(axis=0)) < tol
    lib.call("drs_conv_forward_split", xp.data_ptr(), B, S, P, cin + 32, 32, wf.data_ptr(), None, k, rate, pb,
             cin, cout, out.data_ptr(), cout + 32, 32, 1, None, ns, stream())
    torch.cuda.synchronize()
    ref2 = ref + T.conv2d_same(x64, w64, rate)
    assert rel_err(out.cpu().numpy()[:, 32:].reshape(B, S, S, cout), ref2) < tol

    gx_ref, gw_ref = T.conv2d_same_bwd(x64, w64, rate, g.astype(np.float64))
    gd = padded(g, P, ld=cout, coff=0)
    gp, ng = planes(lib, gd, ns)
    gx = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
    if cin % 64 == 0:       # the input-gradient GEMM has N = cin; narrower layers stay on the exact-fp32 kernel
        lib.call("drs_conv_forward_split", gp.data_ptr(), B, S, P, cout, 0, wg.data_ptr(), None, k, rate, pa, cout, cin,
                 gx.data_ptr(), cin, 0, 0, None, ns, stream())
    nsplit = lib.query("drs_conv_wgrad_split_splits", B, S, k, cin, cout, P, ns)
    slab = torch.zeros(nsplit * nw, dtype=torch.float32, device=DEV)
    gw = torch.zeros(nw, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad_split", xp.data_ptr(), B, S, P, cin + 32, 32, gp.data_ptr(), P, cout, 0, k, rate, pb, cin,
             cin, cout, slab.data_ptr(), gw.data_ptr(), ns, stream())
    torch.cuda.synchronize()
    e_dg = rel_err(gx.cpu().numpy().reshape(B, S, S, cin), gx_ref) if cin % 64 == 0 else 0.0
    e_wg = rel_err(gw.cpu().numpy().reshape(k, k, cin, cout), gw_ref)
    print("k%d r%d %3d->%3d terms %d: fwd %.2e dgrad %.2e wgrad %.2e" % (k, rate, cin, cout, ns, e_fwd, e_dg, e_wg))
    assert e_dg < tol
    assert e_wg < tol


@pytest.mark.parametrize("ns", [2, 3])
def test_conv1_band_padding_split(lib, ns):
    """3..5 image bands ride the 32-channel K-step: the padded filter columns are zero, wgrad drops them again."""
    rng = np.random.default_rng(5)
    B, S, C, cout, k = 2, 11, 5, 64, 5
    x = rng.normal(size=(B, S, S, C)).astype(np.float32)
    w = rng.normal(size=(k, k, C, cout)).astype(np.float32) * 0.1
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    xp, nx = planes(lib, padded(x, 2, ld=32, coff=0), ns)
    nwp = k * k * 32 * cout
    wf = torch.zeros(ns * nwp, dtype=torch.int16, device=DEV)
    lib.call("drs_filter_split", dev(w).data_ptr(), k, C, 32, cout, ns, wf.data_ptr(), None, stream())
    out = torch.zeros(B * S * S * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward_split", xp.data_ptr(), B, S, 2, 32, 0, wf.data_ptr(), None, k, 1, 2, 32, cout,
             out.data_ptr(), cout, 0, 0, None, ns, stream())
    gp, ng = planes(lib, padded(g, 2), ns)
    nsplit = lib.query("drs_conv_wgrad_split_splits", B, S, k, 32, cout, 2, ns)
    slab = torch.zeros(nsplit * nwp, dtype=torch.float32, device=DEV)
    gw = torch.zeros(k * k * C * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad_split", xp.data_ptr(), B, S, 2, 32, 0, gp.data_ptr(), 2, cout, 0, k, 1, 2, 32, C, cout,
             slab.data_ptr(), gw.data_ptr(), ns, stream())
    torch.cuda.synchronize()
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), 1)
    _, gw_ref = T.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), 1, g.astype(np.float64))
    assert rel_err(out.cpu().numpy().reshape(B, S, S, cout), ref) < TOL[ns]
    assert rel_err(gw.cpu().numpy().reshape(k, k, C, cout), gw_ref) < TOL[ns]


# BASELINE shapes of Dilated8Pooling / Dilated6Pooling / DenseDilated6 (k, rate, cin, cout) at a size where the fp64 oracle is quick
BASELINE_SHAPES = [(5, 2, 64, 64), (4, 3, 64, 128), (4, 4, 128, 128), (3, 5, 128, 192), (3, 6, 192, 192), (3, 7, 192, 256), (3, 8, 256, 256),
                   (3, 5, 128, 256), (3, 6, 256, 256), (3, 6, 320, 128)]


def _rms_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


@pytest.mark.parametrize("k,rate,cin,cout", BASELINE_SHAPES)
def test_three_term_arithmetic_is_as_exact_as_the_fp32_mfma_kernels(lib, k, rate, cin, cout):
    """`bf16x6` (3 bf16 terms per operand, 6 products, fp32 accumulate) as an fp32-EQUIVALENT arithmetic: on every BASELINE layer
    shape its forward, input-gradient and filter-gradient errors against the fp64 oracle are no larger than those of the
    exact-fp32 MFMA kernels on the same inputs: both are dominated by the fp32 accumulation.  Compared as RMS errors (a
    maximum over ~10^5 elements scatters by +-40 % between two equally exact arithmetics), within 10 %.  This is a statement about accuracy only: the default arithmetic and the headline stay fp32."""
    B, S, ns = 2, 14, 3
    rng = np.random.default_rng(k * 1000 + rate * 100 + cin + cout)
    x = rng.normal(size=(B, S, S, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    g = rng.normal(size=(B, S, S, cout)).astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    M = B * S * S
    xd, gd, wd = padded(x, P), padded(g, P), dev(w)
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
    ref = T.conv2d_same(x64, w64, rate)
    gx_ref, gw_ref = T.conv2d_same_bwd(x64, w64, rate, g64)
    # exact-fp32 kernels
    out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    gx = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
    gw = torch.zeros(w.size, dtype=torch.float32, device=DEV)
    wt = torch.zeros(w.size, dtype=torch.float32, device=DEV)
    lib.call("drs_filter_flip_transpose", wd.data_ptr(), wt.data_ptr(), k, cin, cout, stream())
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), None, k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0, None, stream())
    lib.call("drs_conv_forward", gd.data_ptr(), B, S, P, cout, 0, wt.data_ptr(), None, k, rate, pa, cout, cin, gx.data_ptr(), cin, 0, 0, None, stream())
    nsp = lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
    slab = torch.zeros(nsp * w.size, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad", xd.data_ptr(), B, S, P, cin, 0, gd.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab.data_ptr(), gw.data_ptr(), stream())
    torch.cuda.synchronize()
    e32 = (_rms_err(out.cpu().numpy().reshape(ref.shape), ref), _rms_err(gx.cpu().numpy().reshape(gx_ref.shape), gx_ref),
           _rms_err(gw.cpu().numpy().reshape(gw_ref.shape), gw_ref))
    # three-term split kernels
    xt, _ = planes(lib, xd, ns)
    gt, _ = planes(lib, gd, ns)
    wf = torch.zeros(ns * w.size, dtype=torch.int16, device=DEV)
    wg = torch.zeros(ns * w.size, dtype=torch.int16, device=DEV)
    lib.call("drs_filter_split", wd.data_ptr(), k, cin, cin, cout, ns, wf.data_ptr(), wg.data_ptr() if cin % 64 == 0 else None, stream())
    out2 = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    gx2 = torch.zeros(M * cin, dtype=torch.float32, device=DEV)
    gw2 = torch.zeros(w.size, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward_split", xt.data_ptr(), B, S, P, cin, 0, wf.data_ptr(), None, k, rate, pb, cin, cout, out2.data_ptr(), cout, 0, 0, None, ns, stream())
    if cin % 64 == 0:
        lib.call("drs_conv_forward_split", gt.data_ptr(), B, S, P, cout, 0, wg.data_ptr(), None, k, rate, pa, cout, cin, gx2.data_ptr(), cin, 0, 0, None, ns,
                 stream())
    nsp2 = lib.query("drs_conv_wgrad_split_splits", B, S, k, cin, cout, P, ns)
    slab2 = torch.zeros(nsp2 * w.size, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_wgrad_split", xt.data_ptr(), B, S, P, cin, 0, gt.data_ptr(), P, cout, 0, k, rate, pb, cin, cin, cout, slab2.data_ptr(),
             gw2.data_ptr(), ns, stream())
    torch.cuda.synchronize()
    e6 = (_rms_err(out2.cpu().numpy().reshape(ref.shape), ref), _rms_err(gx2.cpu().numpy().reshape(gx_ref.shape), gx_ref) if cin % 64 == 0 else 0.0,
          _rms_err(gw2.cpu().numpy().reshape(gw_ref.shape), gw_ref))
    print("k%d r%d %3d->%3d  fp32 MFMA: fwd %.2e dgrad %.2e wgrad %.2e | bf16x6: fwd %.2e dgrad %.2e wgrad %.2e" % ((k, rate, cin, cout) + e32 + e6))
    # forward / input gradient: within 10 %.  Filter gradient: the exact-fp32 kernel cuts the pixel dimension into more, shorter splits
    # since r03 (workgroup count by a per-CU cost model, splits down to 8 chunks), i.e. shorter fp32 accumulation chains
    # and a smaller error of its own (2.1e-7 against the split kernel's 2.7e-7 on conv2's shape at this size): within 50 %.
    for a, b, lim in zip(e6, e32, (1.10, 1.10, 1.50)):
        assert a <= lim * b + 1e-9, (e6, e32)


ADVERSARIAL = ["gaussian", "large_channel_means", "wide_dynamic_range", "cancelling_pairs", "near_denormal"]


@pytest.mark.parametrize("case", ADVERSARIAL)
def test_three_term_arithmetic_on_adversarial_operands(lib, case):
    """Where `bf16x6` stands against the exact-fp32 MFMA kernels OUTSIDE Gaussian operands -- the reason it stays an opt-in
    arithmetic of the op-level path and earns no headline.  A product of 3-term operands keeps 6 of the 9 partial products: the
    three dropped ones are ~2^-24 .. 2^-32 of the product each, so a product carries ~2^-22.5 relative error where an fp32 FMA
    carries none (its only rounding is the accumulation's).  On sums of many comparable terms the accumulation rounding dominates
    both arithmetics (Gaussian operands: ratio ~0.85-1); when a few large terms dominate a sum, or the operands sit on a large
    common offset, the product error shows.  The forward product of conv8's shape, errors against the fp64 oracle relative to the
    largest sum |a| |b|; asserted: the bound the path is held to everywhere (1e-5), and the measured ratio to fp32 per case
    (profiles/r03/bf16x6_adversarial.log: RMS ratio 0.86 Gaussian / large means, 1.14 wide range (max 1.97), 2.36 cancelling pairs).
    VERDICT r02 asked for <= 1.1 x the fp32 kernels' RMS and MAX on such operands before the arithmetic may carry the headline: it
    does not meet that, by construction, so it is not promoted into the step engine."""
    k, rate, cin, cout, B, S, ns = 3, 8, 256, 256, 2, 14, 3
    rng = np.random.default_rng(17)
    x = rng.normal(size=(B, S, S, cin))
    w = rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)
    if case == "large_channel_means":             # activations on a per-channel offset 10^3 x their spread (what forced the two-pass BN statistics)
        x = x + 1000.0 * rng.normal(size=(1, 1, 1, cin))
    elif case == "wide_dynamic_range":            # magnitudes spanning 2^-20 .. 2^20 inside every dot product
        x = x * np.exp2(rng.uniform(-20, 20, size=x.shape))
        w = w * np.exp2(rng.uniform(-20, 20, size=w.shape))
    elif case == "cancelling_pairs":              # neighbouring channels carry +a, -a against the same filter value: the sum is rounding only
        x[..., 1::2] = -x[..., 0::2]
        w[:, :, 1::2, :] = w[:, :, 0::2, :]
        x = x + 1e-3 * rng.normal(size=x.shape)   # (plus a small signal, so that the reference is not identically zero)
    elif case == "near_denormal":                 # products around 2^-126: the low bf16 terms fall into the denormal range
        x = x * 2.0 ** -100
        w = w * 2.0 ** -20
    x, w = x.astype(np.float32), w.astype(np.float32)
    pb, pa = onets.same_pad(k, rate)
    P = max(pb, pa)
    M = B * S * S
    xd, wd = padded(x, P), dev(w)
    ref = T.conv2d_same(x.astype(np.float64), w.astype(np.float64), rate)
    out = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward", xd.data_ptr(), B, S, P, cin, 0, wd.data_ptr(), None, k, rate, pb, cin, cout, out.data_ptr(), cout, 0, 0, None, stream())
    xt, _ = planes(lib, xd, ns)
    wf = torch.zeros(ns * w.size, dtype=torch.int16, device=DEV)
    lib.call("drs_filter_split", wd.data_ptr(), k, cin, cin, cout, ns, wf.data_ptr(), None, stream())
    out2 = torch.zeros(M * cout, dtype=torch.float32, device=DEV)
    lib.call("drs_conv_forward_split", xt.data_ptr(), B, S, P, cin, 0, wf.data_ptr(), None, k, rate, pb, cin, cout, out2.data_ptr(), cout, 0, 0, None, ns, stream())
    torch.cuda.synchronize()
    scale = T.conv2d_same(np.abs(x).astype(np.float64), np.abs(w).astype(np.float64), rate).max()     # sum |a| |b|: the error scale of a dot product
    a32, a6 = out.cpu().numpy().reshape(ref.shape).astype(np.float64), out2.cpu().numpy().reshape(ref.shape).astype(np.float64)
    rms32, rms6 = np.sqrt(np.mean((a32 - ref) ** 2)) / scale, np.sqrt(np.mean((a6 - ref) ** 2)) / scale
    max32, max6 = np.abs(a32 - ref).max() / scale, np.abs(a6 - ref).max() / scale
    print("%-20s fp32 MFMA rms %.2e max %.2e | bf16x6 rms %.2e max %.2e | ratio rms %.2f max %.2f" % (case, rms32, max32, rms6, max6, rms6 / rms32, max6 / max32))
    assert max32 < 1e-5 and max6 < 1e-5                     # the parity bound both arithmetics are held to
    LIMIT = {"gaussian": (1.1, 1.1), "large_channel_means": (1.1, 1.1), "wide_dynamic_range": (1.5, 3.0), "cancelling_pairs": (3.5, 3.5),
             "near_denormal": (1.1, 1.1)}
    assert rms6 <= LIMIT[case][0] * rms32 + 1e-12 and max6 <= LIMIT[case][1] * max32 + 1e-12
