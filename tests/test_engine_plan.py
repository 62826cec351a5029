"""CPU: the step-level library's net tables and variable layout (csrc/engine.hip, no GPU touched by drs_net_create) against
nets.Plan, the host mirror of the reference's net builders, for every net_type of the three scripts; and the boundary's error
behaviour."""
import ctypes as C

import numpy as np
import pytest

from drs_amd import _lib
from drs_amd.nets import Plan, known_net_types


def _create(net_type, ch, K, b=4, s=30):
    hp = C.c_void_p()
    rc = _lib.load().drs_net_create(net_type.encode(), ch, K, 0.005, b, s, 1, 0.5, C.byref(hp))
    return rc, hp


@pytest.mark.parametrize("net_type", known_net_types())
@pytest.mark.parametrize("ch,K", [(5, 6), (3, 2), (4, 7)])
def test_library_layout_equals_plan(net_type, ch, K):
    rc, h = _create(net_type, ch, K)
    assert rc == 0
    p = Plan(net_type, ch, K)
    name = C.create_string_buffer(96)
    off, cnt, inbn = C.c_size_t(), C.c_size_t(), C.c_int()
    shape = (C.c_int * 4)()
    seen = {}
    for i in range(_lib.query("drs_net_num_variables", h)):
        _lib.call("drs_net_variable_info", h, i, name, 96, C.byref(off), C.byref(cnt), shape, C.byref(inbn))
        seen[name.value.decode()] = (off.value, cnt.value, tuple(v for v in shape if v), inbn.value)
    for n, (o, shp) in p.offsets.items():
        assert seen[n][:2] == (o, int(np.prod(shp))) and seen[n][3] == 0, n
        assert seen[n][2] == tuple(shp), n
    for L in p.layers:
        o = p.bn_offsets[L.name]
        assert seen[L.name + "/moving_mean"] == (o, L.cout, (L.cout,), 1)
        assert seen[L.name + "/moving_variance"] == (o + L.cout, L.cout, (L.cout,), 1)
    assert len(seen) == len(p.offsets) + 2 * len(p.layers)
    npar, ndec, nbn, nl, c0, p0 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
    _lib.call("drs_net_layout", h, C.byref(npar), C.byref(ndec), C.byref(nbn), C.byref(nl), C.byref(c0), C.byref(p0))
    assert (npar.value, ndec.value, nbn.value, nl.value) == (p.n_params, p.n_decay, p.n_bn, len(p.layers))
    assert (c0.value, p0.value) == p.buffers["x0"]
    # the blocks themselves: geometry, wiring and pooling of every `_conv_layer` call, in execution order
    geom = (C.c_int * 8)()
    src, dst = C.create_string_buffer(32), C.create_string_buffer(32)
    coff, pool = C.c_int(), C.c_int()
    for i, L in enumerate(p.layers):
        _lib.call("drs_net_layer_info", h, i, name, 96, geom, src, dst, 32, C.byref(coff), C.byref(pool))
        assert name.value.decode() == L.name
        assert tuple(geom) == (L.k, L.rate, L.cin, L.cin_k, L.cout, L.pad_b, L.pad_a, L.halo), (L.name, tuple(geom))
        assert (src.value.decode(), dst.value.decode(), coff.value) == (L.src, L.dst, L.dst_coff)
        q = p.pools[i]
        assert pool.value == (0 if q is None else (1 if q[0] == "max" else 2 + 256 * q[1])), (L.name, pool.value, q)
    assert _lib.load().drs_net_layer_info(h, len(p.layers), name, 96, geom, src, dst, 32, None, None) == 1
    # buffers: every activation slab of the plan with its halo, sized for (b_max, s_max)
    nb, dt = C.c_size_t(), C.c_int()
    bufs = {}
    for i in range(_lib.query("drs_net_num_buffers", h)):
        _lib.call("drs_net_buffer_info", h, i, name, 96, C.byref(nb), C.byref(dt))
        bufs[name.value.decode()] = (nb.value, dt.value)
    for sname, (Cc, Pp) in p.buffers.items():
        assert bufs["act:" + sname] == (4 * 4 * (30 + 2 * Pp) ** 2 * Cc, 0), sname
    assert bufs["params"] == (4 * p.n_params, 0) and bufs["conf"] == (4 * K * K, 3) and bufs["labels"] == (4 * 30 * 30, 2)
    _lib.load().drs_net_destroy(h)


def test_boundary_rejects_bad_arguments():
    assert _create("no_such_net", 5, 6)[0] == 1                 # the reference prints "Net type not identified" (isprs:1679)
    assert _create("dilated_grsl", 0, 6)[0] == 1 and _create("dilated_grsl", 5, 9)[0] == 1
    assert _create("dilated_grsl", 5, 6, b=4096, s=100)[0] == 1   # B*S*S must stay below 2^24
    rc, h = _create("dilated_grsl", 5, 6)
    st = None
    # nothing bound yet: a step is refused, not executed on null pointers
    assert _lib.load().drs_train_step(h, 2, 20, 0.01, 0, 0.0, st) == 1
    assert _lib.load().drs_forward(h, 2, 20, 0, -1, st) == 1
    assert _lib.load().drs_net_bind(h, b"no_such_buffer", C.c_void_p(16), 1024) == 1
    assert _lib.load().drs_net_bind(h, b"params", C.c_void_p(16), 8) == 1          # too small
    assert _lib.query("drs_net_global_step", h, -1) == 0 and _lib.query("drs_net_global_step", h, 70000) == 70000
    assert abs(_lib.query("drs_net_learning_rate", h, 0.01) - 0.005) < 1e-9       # staircase decay, factor 0.5 (isprs:1686)
    _lib.load().drs_net_destroy(h)
