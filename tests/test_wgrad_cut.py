"""CPU: how drs_conv_wgrad cuts the pixel dimension of the filter gradient into workgroups (conv_mfma.hip, WgradPlan).

The library works the cut out on the host with the very code the kernels run (`wgrad_assign` is host + device); the development
entry point drs_debug_wgrad_cut lists the workgroups of a launch.  Checked here, without a GPU, before any kernel relies on it:
every (row tile, column tile) has its splits 0 .. n-1 exactly once, their chunk ranges tile [0, chunks) without gap or overlap,
n stays inside what drs_conv_wgrad_splits tells callers to size the slab by, and -- the point of the cut by live pixels -- the
workgroups of a launch all multiply (nearly) the same number of live chunks.
"""
import ctypes as C

import numpy as np
import pytest

from drs_amd import _lib


def _live_range(row_first, row_last, cin, k, rate, pad, S):
    """restatement of live_pixel_range (drs_common.hpp): image rows whose shifted rows meet image data for some tap row of the tile"""
    dy_min = (row_first // cin // k) * rate - pad
    dy_max = (row_last // cin // k) * rate - pad
    ylo, yhi = max(0, -dy_max), min(S, S - dy_min)
    if ylo >= yhi:
        ylo, yhi = 0, S
    return ylo * S, yhi * S


def _cut(B, S, k, rate, pad, cin, cout):
    lib = _lib.dev()          # libdrs_hip_dev.so: the product library does not export the development entry points
    cap = 1 << 16
    out = np.zeros((cap, 5), dtype=np.int32)
    nt = np.zeros(256, dtype=np.int32)
    tr = C.c_int()
    n = lib.drs_debug_wgrad_cut(B, S, k, rate, pad, cin, cout, out.ctypes.data, cap, nt.ctypes.data, C.addressof(tr))
    assert 0 < n <= cap
    return out[:n], nt, tr.value


def _live_chunks(cbeg, cend, lo, hi, S2):
    """32-pixel chunks of [cbeg, cend) that hold a pixel of some image's live range [b*S2+lo, b*S2+hi)"""
    c = np.arange(cbeg, cend, dtype=np.int64)
    first, last = 32 * c, 32 * c + 31
    live = np.zeros(len(c), dtype=bool)
    for b in range(int(first[0] // S2) if len(c) else 0, int(last[-1] // S2) + 1 if len(c) else 0):
        live |= (last >= b * S2 + lo) & (first < b * S2 + hi)
    return int(live.sum())


def _all_conv_shapes():
    """(k, rate, pad_before, cin as the kernel sees it, cout) of every convolution of every net table of the three scripts"""
    from drs_amd.nets import Plan, known_net_types
    shapes = set()
    for nt in known_net_types():
        for ch in (3, 4, 5):
            for L in Plan(nt, ch, 6, first_cin_pad=8).layers:
                shapes.add((L.k, L.rate, L.pad_b, L.cin_k, L.cout))
    return sorted(shapes)


SHAPES = _all_conv_shapes()


@pytest.mark.parametrize("B,S", [(16, 64), (32, 64), (128, 64), (128, 25), (128, 85), (16, 45), (3, 9), (1, 100)])
def test_cut_is_a_partition_and_balanced(B, S):
    lib = _lib.load()
    assert len(SHAPES) >= 40
    for (k, rate, pad, cin, cout) in SHAPES:
        wg, nt, tr = _cut(B, S, k, rate, pad, cin, cout)
        rows = k * k * cin
        ntr = -(-rows // tr)
        nchunks = -(-B * S * S // 32)
        bound = _lib.query("drs_conv_wgrad_splits", B, S, k, cin, cout)
        seen = {}
        for rt, ct, sp, cb, ce in wg:
            assert 0 <= rt < ntr and 0 <= sp < nt[rt] <= bound, (rt, sp, nt[rt], bound)
            assert (rt, ct, sp) not in seen
            assert 0 <= cb <= ce <= nchunks
            seen[(rt, ct, sp)] = (cb, ce)
        ncol = 1 + max(ct for _, ct, _, _, _ in wg)
        assert len(seen) == sum(int(nt[r]) for r in range(ntr)) * ncol == len(wg)
        lens = []
        for rt in range(ntr):
            lo, hi = _live_range(rt * tr, min((rt + 1) * tr, rows) - 1, cin, k, rate, pad, S)
            for ct in range(ncol):
                edge = 0
                for sp in range(int(nt[rt])):
                    cb, ce = seen[(rt, ct, sp)]
                    assert cb == edge, "gap or overlap between the splits of a tile"
                    edge = ce
                    if ct == 0:
                        lens.append(_live_chunks(cb, ce, lo, hi, S * S))
                assert edge == nchunks
        lens = np.asarray(lens)
        # workgroup lengths in live chunks: equal up to the rounding of n per class and the chunks that straddle a boundary
        if len(set(int(x) for x in nt[:ntr])) > 1 or S % 32 == 0:
            # what delays a launch is its LONGEST workgroup (a tile's last split may be short: the remainder of its live pixels)
            over = (lens.max() - lens.mean()) / max(1.0, lens.mean())
            assert over <= 0.2 or lens.max() - lens.mean() <= 3, (B, S, k, rate, cin, cout, lens.min(), lens.mean(), lens.max())

def test_equal_cut_still_available():
    lib = _lib.dev()
    old = lib.drs_debug_wgrad_balance(0)
    try:
        wg, nt, tr = _cut(128, 64, 3, 8, 8, 256, 256)
        assert len(set(int(x) for x in nt[:18])) == 1
        per = {}
        for rt, ct, sp, cb, ce in wg:
            per.setdefault((rt, ct), []).append((sp, cb, ce))
        for v in per.values():
            v.sort()
            assert v[0][1] == 0 and v[-1][2] == 128 * 64 * 64 // 32 and all(v[i][2] == v[i + 1][1] for i in range(len(v) - 1))
    finally:
        lib.drs_debug_wgrad_balance(old)


def test_split_path_slab_bound_is_monotone_in_batch_and_size():
    """the split-bf16 filter-gradient workspace is sized once, at (b_max, s_max), while the patch side changes every step
    (isprs:1727-1737) and the exact split count is not monotone in S.  drs_conv_wgrad_split_splits therefore returns a bound that
    is: every (b <= b_max, s <= s_max) needs no more than the allocation.  Swept over the nets / sizes the advisor found overflowing."""
    from drs_amd import _lib
    from drs_amd.nets import Plan
    _lib.load()
    for net_type in ("dilated_icpr_rate6_small", "dilated_icpr_rate6_squeeze", "dilated_grsl_rate8", "dilated_icpr_rate6_densely"):
        plan = Plan(net_type, 5, 6, first_cin_pad=8)
        for ns in (2, 3):
            for b_max, s_max in ((32, 85), (32, 100), (64, 65), (128, 45)):
                for L in plan.layers:
                    if L.cout % 64 or L.cin_k % 32:
                        continue
                    alloc = _lib.query("drs_conv_wgrad_split_splits", b_max, s_max, L.k, L.cin_k, L.cout, L.halo, ns)
                    for b in (1, b_max // 2, b_max):
                        for s in range(1, s_max + 1):
                            assert _lib.query("drs_conv_wgrad_split_splits", b, s, L.k, L.cin_k, L.cout, L.halo, ns) <= alloc, (net_type, L.name, b, s)


def test_slab_sized_at_the_largest_step_serves_every_smaller_one():
    """The step engine sizes the filter-gradient slab once, at (b_max, s_max) (engine.hip list_buffers), while the patch side changes
    every step (isprs:1727-1737) and the last batch of a validation pass is short: no (b <= b_max, s <= s_max) may cut a row tile
    into more splits than drs_conv_wgrad_splits(b_max, s_max) allows for.  (The exact count is not monotone in S.)"""
    from drs_amd.nets import Plan
    lib = _lib.dev()
    for net_type, ch, K in (("dilated_grsl_rate8", 5, 6), ("dilated_icpr_rate6_densely", 4, 2), ("dilated_grsl", 5, 6), ("dilated_icpr_original", 3, 6)):
        plan = Plan(net_type, ch, K, first_cin_pad=8)
        for b_max, s_max in ((16, 85), (32, 100), (128, 64), (64, 65), (4, 25)):
            for L in plan.layers:
                alloc = _lib.query("drs_conv_wgrad_splits", b_max, s_max, L.k, L.cin_k, L.cout)
                nt = np.zeros(256, dtype=np.int32)
                for b in sorted(v for v in {1, 2, 3, 5, 8, 13, b_max // 2, b_max - 1, b_max} if 1 <= v <= b_max):
                    for s in range(1, s_max + 1):
                        n = lib.drs_debug_wgrad_cut(b, s, L.k, L.rate, L.pad_b, L.cin_k, L.cout, None, 0, nt.ctypes.data, None)
                        assert n > 0 and int(nt.max()) <= alloc, (net_type, L.name, b, s, int(nt.max()), alloc, b_max, s_max)
                        nt[:] = 0


# ---------------------------------------------------------------------------------------------------------------------
# launch order of the plain forward / input-gradient launches: "full tiles first, the tiles that skip halo tap rows last"
# (conv_mfma.hip lpt_tile / conv_lpt_setup, read back through the development library on the host)
def _conv_order(B, S, k, rate, pad, cin, cout):
    import ctypes as C
    from drs_amd import _lib
    d = _lib.dev()
    n = d.drs_debug_conv_order(B, S, k, rate, pad, cin, cout, None, 0)
    if n <= 0:
        return n, None
    out = (C.c_int * n)()
    assert d.drs_debug_conv_order(B, S, k, rate, pad, cin, cout, out, n) == n
    return n, list(out)


@pytest.mark.parametrize("B,S,k,rate,cin,cout", [(128, 64, 3, 8, 256, 256), (128, 64, 3, 5, 128, 192), (128, 64, 5, 2, 64, 64), (128, 64, 4, 3, 64, 128),
                                                 (64, 64, 3, 6, 256, 256), (512, 32, 3, 4, 128, 256), (32, 128, 3, 8, 256, 256),
                                                 (16, 64, 3, 8, 256, 256), (32, 64, 3, 5, 128, 192)])      # one- and two-round launches too
def test_full_tiles_first_order_is_a_bijection_that_ends_with_the_short_tiles(B, S, k, rate, cin, cout):
    pad = (k - 1) * rate // 2
    n, order = _conv_order(B, S, k, rate, pad, cin, cout)
    M = B * S * S
    bn = 192 if cout % 192 == 0 and cout % 128 else (128 if cout % 128 == 0 else 64)
    nt = cout // bn
    assert n == (M // 128) * nt and sorted(order) == list(range(n))                 # every tile exactly once
    for w in range(0, n, nt):                                                       # the column tiles of an M tile stay adjacent
        assert [t % nt for t in order[w:w + nt]] == list(range(nt)) and len({t // nt for t in order[w:w + nt]}) == 1
    T, rows = S * S // 128, 128 // S

    def full(t):                                                                    # drs_common.hpp live_tap_rows for rows [t rows, (t+1) rows)
        y0, y1 = t * rows, t * rows + rows - 1
        lo = -(-(pad - y1) // rate) if pad - y1 > 0 else 0
        hi = min((S - 1 - y0 + pad) // rate + 1, k)
        return lo == 0 and hi == k
    chunk = n // 8
    for c in range(8):                                                              # per XCD chunk: whole patches, full tiles first
        part = order[c * chunk:(c + 1) * chunk]
        patches = sorted({t // nt // T for t in part})
        assert len(patches) * T * nt == chunk and patches == list(range(patches[0], patches[0] + len(patches)))
        kinds = [full((t // nt) % T) for t in part]
        first_short = kinds.index(False)
        assert all(kinds[:first_short]) and not any(kinds[first_short:]) and 0 < first_short < chunk
        m_full = [t // nt for t in part[:first_short:nt]]
        assert m_full == sorted(m_full)                                             # the full tiles keep their natural (patch-major) order


@pytest.mark.parametrize("B,S,k,rate,cin,cout", [(18, 64, 3, 8, 256, 256),       # 1152 tiles for 1024 places: a stream-K launch (equal K ranges, nothing skipped)
                                                 (128, 65, 3, 8, 256, 256),      # tiles do not hold whole image rows
                                                 (128, 64, 1, 1, 256, 256)])     # nothing to skip: every tile is full
def test_full_tiles_first_order_is_off_where_it_does_not_apply(B, S, k, rate, cin, cout):
    n, order = _conv_order(B, S, k, rate, (k - 1) * rate // 2, cin, cout)
    assert n == 0 and order is None
