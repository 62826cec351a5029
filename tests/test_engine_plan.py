"""CPU: the step-level library's net tables and variable layout (csrc/engine.hip, no GPU touched by drs_net_create) -- the ONE copy
of the tables in the product -- against oracle/nets.py, the independent restatement of the reference's net builders, for every
net_type of the three scripts; nets.Plan (a read-back of the library, no table of its own); and the boundary's error behaviour."""
import ctypes as C

import numpy as np
import pytest

from drs_amd import _lib
from drs_amd.nets import Plan, known_net_types


def _create(net_type, ch, K, b=4, s=30):
    hp = C.c_void_p()
    rc = _lib.load().drs_net_create(net_type.encode(), ch, K, 0.005, b, s, 1, 0.5, C.byref(hp))
    return rc, hp


def _expected(net_type, ch, K):
    """what the library must hold for this net, worked out from the ORACLE's tables (oracle/nets.py: the independent restatement of the
    reference's builders) and the layout rules of DESIGN.md 2: blocks with their wiring, slabs, pools, SE blocks, variable layout"""
    from oracle import nets as O
    spec = O.NETS[O.resolve(net_type)]
    convs = O.conv_specs(net_type, ch)
    x0c = (ch + 7) // 8 * 8 if ch <= 8 else (ch + 31) // 32 * 32
    blocks = []                                           # (name, k, cin, cout, rate, src, dst, coff)
    if spec.get("squeezes"):                              # isprs:726-742: 1x1 squeeze, then 1x1 and k x k expands concatenated
        n0 = convs[0]
        blocks.append((n0[0], n0[1], n0[2], n0[3], n0[4], "x0", "c1", 0))
        for j, (n, k, ind, outd, r, kd) in enumerate(spec["squeezes"], start=2):
            blocks += [(n + "_s1", 1, ind, kd, r, "c%d" % (j - 1), "a%d" % j, 0), (n + "_s2_1", 1, kd, outd // 2, r, "a%d" % j, "c%d" % j, 0),
                       (n + "_s2_2", k, kd, outd // 2, r, "a%d" % j, "c%d" % j, outd // 2)]
        feat = "c%d" % (len(spec["squeezes"]) + 1)
    elif spec["dense"]:                                   # isprs:921-948: every block reads the concat so far, writes its slice
        off = 0
        for i, (n, k, ci, co, r) in enumerate(convs):
            blocks.append((n, k, ci, co, r, "x0" if i == 0 else "concat", "concat", off))
            off += co
        feat = "concat"
    else:
        for i, (n, k, ci, co, r) in enumerate(convs):
            blocks.append((n, k, ci, co, r, "x%d" % i, "x%d" % (i + 1) if i + 1 < len(convs) else "feat", 0))
        feat = "feat"
    pools = [("max", 3) if spec["pool"] else None] * len(blocks)
    for i, a in enumerate(spec.get("pools", [])):
        pools[i] = ("avg", a) if a else None
    chan, halo = {"x0": x0c}, {"x0": 0}
    layers = []
    for (n, k, ci, co, r, src, dst, coff) in blocks:
        pb, pa = O.same_pad(k, r)
        layers.append((n, k, r, ci, x0c if src == "x0" else (ci + 31) // 32 * 32, co, pb, pa, max(pb, pa), src, dst, coff))
        chan[dst] = max(chan.get(dst, 0), coff + co)
        halo[src] = max(halo.get(src, 0), pb, pa)
        halo.setdefault(dst, 0)
    variables, off = {}, 0
    for (n, k, r, ci, cik, co, *_rest) in layers:
        variables[n + "/weights"] = (off, (k, k, ci, co)); off += k * k * ci * co
    variables["conv_classifier/weights"] = (off, (1, 1, spec["c_last"], K)); off += spec["c_last"] * K
    se = spec.get("se", {})
    for i, scope in sorted(se.items()):
        Cc = layers[i][5]
        variables[scope + "_fc1/weights"] = (off, (Cc, Cc // 4)); off += Cc * (Cc // 4)
        variables[scope + "_fc2/weights"] = (off, (Cc // 4, Cc)); off += Cc * (Cc // 4)
    n_decay = off                                         # weight decay: kernels and FC weights only (isprs:640-652)
    for L in layers:
        variables[L[0] + "/biases"] = (off, (L[5],)); off += L[5]
    variables["conv_classifier/biases"] = (off, (K,)); off += K
    for i, scope in sorted(se.items()):
        Cc = layers[i][5]
        variables[scope + "_fc1/biases"] = (off, (Cc // 4,)); off += Cc // 4
        variables[scope + "_fc2/biases"] = (off, (Cc,)); off += Cc
    return dict(layers=layers, pools=pools, chan=chan, halo=halo, feat=feat, variables=variables, n_params=off, n_decay=n_decay,
                alpha=0.0 if spec["act"] == "relu" else 0.1, c_last=spec["c_last"], se=se, dense=spec["dense"])


def test_the_library_accepts_exactly_the_oracles_net_types():
    from oracle import nets as O
    assert sorted(known_net_types()) == sorted(list(O.NETS) + list(O.ALIASES))
    from drs_amd.nets import resolve
    for a, b in O.ALIASES.items():
        assert resolve(a) == b
    with pytest.raises(ValueError):
        resolve("no_such_net")


@pytest.mark.parametrize("net_type", known_net_types())
@pytest.mark.parametrize("ch,K", [(5, 6), (3, 2), (4, 7)])
def test_library_tables_equal_the_oracles(net_type, ch, K):
    """the ONE copy of the net tables in the product (csrc/engine.hip) against the oracle's independent restatement, read through the
    C ABI (drs_net_layer_info / _variable_info / _info / _se_info / _layout / _buffer_info) and through nets.Plan (which holds no table)"""
    rc, h = _create(net_type, ch, K)
    assert rc == 0
    e = _expected(net_type, ch, K)
    p = Plan(net_type, ch, K)
    name = C.create_string_buffer(96)
    off, cnt, inbn = C.c_size_t(), C.c_size_t(), C.c_int()
    shape = (C.c_int * 4)()
    seen = {}
    for i in range(_lib.query("drs_net_num_variables", h)):
        _lib.call("drs_net_variable_info", h, i, name, 96, C.byref(off), C.byref(cnt), shape, C.byref(inbn))
        seen[name.value.decode()] = (off.value, cnt.value, tuple(v for v in shape if v), inbn.value)
    for n, (o, shp) in e["variables"].items():
        assert seen[n] == (o, int(np.prod(shp)), tuple(shp), 0), n
    bo = 0
    for L in e["layers"]:
        assert seen[L[0] + "/moving_mean"] == (bo, L[5], (L[5],), 1) and seen[L[0] + "/moving_variance"] == (bo + L[5], L[5], (L[5],), 1)
        bo += 2 * L[5]
    assert len(seen) == len(e["variables"]) + 2 * len(e["layers"])
    npar, ndec, nbn, nl, c0, p0 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
    _lib.call("drs_net_layout", h, C.byref(npar), C.byref(ndec), C.byref(nbn), C.byref(nl), C.byref(c0), C.byref(p0))
    assert (npar.value, ndec.value, nbn.value, nl.value) == (e["n_params"], e["n_decay"], bo, len(e["layers"]))
    assert (c0.value, p0.value) == (e["chan"]["x0"], e["halo"]["x0"])
    # the blocks themselves: geometry, wiring and pooling of every `_conv_layer` call, in execution order
    geom = (C.c_int * 8)()
    src, dst = C.create_string_buffer(32), C.create_string_buffer(32)
    coff, pool = C.c_int(), C.c_int()
    for i, L in enumerate(e["layers"]):
        _lib.call("drs_net_layer_info", h, i, name, 96, geom, src, dst, 32, C.byref(coff), C.byref(pool))
        assert (name.value.decode(),) + tuple(geom) + (src.value.decode(), dst.value.decode(), coff.value) == \
            (L[0], L[1], L[2], L[3], L[4], L[5], L[6], L[7], L[8], L[9], L[10], L[11]), L[0]
        q = e["pools"][i]
        assert pool.value == (0 if q is None else (1 if q[0] == "max" else 2 + 256 * q[1])), (L[0], pool.value, q)
    assert _lib.load().drs_net_layer_info(h, len(e["layers"]), name, 96, geom, src, dst, 32, None, None) == 1
    alpha, c_last, topo, n_se = C.c_float(), C.c_int(), C.c_int(), C.c_int()
    _lib.call("drs_net_info", h, name, 96, C.byref(alpha), C.byref(c_last), dst, 32, C.byref(topo), C.byref(n_se))
    from oracle import nets as O
    assert name.value.decode() == O.resolve(net_type) and abs(alpha.value - e["alpha"]) < 1e-7 and c_last.value == e["c_last"]
    assert dst.value.decode() == e["feat"] and n_se.value == len(e["se"]) and (topo.value == 1) == e["dense"]
    li, cc, rr = C.c_int(), C.c_int(), C.c_int()
    for j, (i, scope) in enumerate(sorted(e["se"].items())):
        _lib.call("drs_net_se_info", h, j, name, 96, C.byref(li), C.byref(cc), C.byref(rr))
        assert (name.value.decode(), li.value, cc.value, rr.value) == (scope, i, e["layers"][i][5], e["layers"][i][5] // 4)
    assert _lib.load().drs_net_se_info(h, len(e["se"]), name, 96, None, None, None) == 1
    # buffers: every activation slab with its halo, sized for (b_max, s_max)
    nb, dt = C.c_size_t(), C.c_int()
    bufs = {}
    for i in range(_lib.query("drs_net_num_buffers", h)):
        _lib.call("drs_net_buffer_info", h, i, name, 96, C.byref(nb), C.byref(dt))
        bufs[name.value.decode()] = (nb.value, dt.value)
    for sname, Cc in e["chan"].items():
        assert bufs["act:" + sname] == (4 * 4 * (30 + 2 * e["halo"][sname]) ** 2 * Cc, 0), sname
    assert bufs["params"] == (4 * e["n_params"], 0) and bufs["conf"] == (4 * K * K, 3) and bufs["labels"] == (4 * 30 * 30, 2)
    _lib.load().drs_net_destroy(h)
    # nets.Plan is that read-back, nothing more
    assert [tuple(L) for L in p.layers] == [(L[0], L[1], L[3], L[4], L[5], L[2], L[6], L[7], L[8], L[9], L[10], L[11]) for L in e["layers"]]
    assert p.pools == e["pools"] and p.buffers == {n: (e["chan"][n], e["halo"][n]) for n in e["chan"]}
    assert p.offsets == e["variables"] and list(p.offsets) == list(e["variables"])
    assert (p.n_params, p.n_decay, p.n_bn, p.feat, p.c_last, p.alpha, p.dense, p.se) == \
        (e["n_params"], e["n_decay"], bo, e["feat"], e["c_last"], e["alpha"], e["dense"], e["se"])


def test_nets_module_holds_no_layer_table():
    import inspect
    import drs_amd.nets as N
    src = inspect.getsource(N)
    assert '("conv' not in src and '("main_conv' not in src and "_TABLES" not in src and "_ALIASES" not in src


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
