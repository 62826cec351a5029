"""The execution plan of a net, read back from the library.

The net tables of the three scripts (the reference's builders, /root/reference/isprs_dilated_random.py:761-1086,
coffee_dilated_random.py:665-841, contest_dilated_random.py:574-641; dispatch by `net_type`, isprs:1660-1680) live in ONE place
of the product: csrc/engine.hip, behind `drs_net_create`.  This module holds no layer literal: `Plan` creates the library-side net
(no kernel is launched and no device memory is touched; the library does ask the HIP runtime for the current device's CU count
when it sizes the stream-K workspace -- `drs_conv_workspace_floats` -- so on a GPU host set the device before the first `Plan`; a
host without a GPU gets the MI355X's 256), reads the blocks (`drs_net_layer_info`), the variable layout under TensorFlow's scope names
(`drs_net_variable_info`), the squeeze-and-excitation blocks and the net-wide facts (`drs_net_info`) and presents them to the Python
host code.  The independent statement of the same tables is oracle/nets.py (test infrastructure); tests/test_engine_plan.py holds
the library to it for every net_type.
"""
import ctypes as C
from collections import namedtuple

from . import _lib

# one conv block (conv + bias + BN + activation [+ pool]); it reads channels [0, cin) of activation buffer `src` and writes its
# output into channels [dst_coff, dst_coff + cout) of buffer `dst`
Layer = namedtuple("Layer", "name k cin cin_k cout rate pad_b pad_a halo src dst dst_coff")
SE_RATIO = 4


def _type_table():
    """name -> canonical name, for every net_type the library accepts (tables and aliases)"""
    names, canon = [], []
    name, ci = C.create_string_buffer(96), C.c_int()
    i = 0
    while _lib.load().drs_net_type_name(i, name, 96, C.byref(ci)) == 0:
        names.append(name.value.decode())
        canon.append(ci.value)
        i += 1
    return {n: names[c] for n, c in zip(names, canon)}


_TYPES = None


def _types():
    global _TYPES
    if _TYPES is None:
        _TYPES = _type_table()
    return _TYPES


def known_net_types():
    return sorted(_types())


def resolve(net_type):
    t = _types()
    if net_type not in t:
        # the reference prints a red message and returns None (isprs:1679-1680); the mirror raises
        raise ValueError("Error! Net type not identified: " + str(net_type))
    return t[net_type]


def same_pad(k, rate):
    """TF SAME padding at stride 1: total (k-1)*rate, the extra pixel goes after (bottom/right)."""
    total = (k - 1) * rate
    return total // 2, total - total // 2


def round_up(v, m):
    return (v + m - 1) // m * m


class Plan(object):
    """Static description of one net for given (bands, classes), as the library runs it."""

    def __init__(self, net_type, channels, num_classes, first_cin_pad=8):
        """first_cin_pad: channel count the image bands are padded to in the conv1 input slab: 8 (what the library's exact-fp32
        kernels take: several filter taps share a 32-deep K-step) or 32 (one tap per K-step; the op-level split-bf16 tools)."""
        self.net_type = resolve(net_type)
        self.channels, self.K = channels, num_classes
        hp = C.c_void_p()
        _lib.call("drs_net_create", self.net_type.encode(), channels, num_classes, 0.0, 1, 1, 1, 0.5, C.byref(hp))
        try:
            self._read(hp, first_cin_pad)
        finally:
            _lib.load().drs_net_destroy(hp)

    def _read(self, h, first_cin_pad):
        name, src, dst = C.create_string_buffer(96), C.create_string_buffer(32), C.create_string_buffer(32)
        alpha, c_last, topo, n_se = C.c_float(), C.c_int(), C.c_int(), C.c_int()
        _lib.call("drs_net_info", h, name, 96, C.byref(alpha), C.byref(c_last), dst, 32, C.byref(topo), C.byref(n_se))
        assert name.value.decode() == self.net_type
        self.alpha = round(float(alpha.value), 6)        # max(alpha*x, x): ReLU / leaky ReLU (isprs:620-621)
        self.c_last, self.feat, self.dense = c_last.value, dst.value.decode(), topo.value == 1
        npar, ndec, nbn, nl, c0, p0 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        _lib.call("drs_net_layout", h, C.byref(npar), C.byref(ndec), C.byref(nbn), C.byref(nl), C.byref(c0), C.byref(p0))
        self.n_params, self.n_decay, self.n_bn = npar.value, ndec.value, nbn.value
        # ---- blocks in execution order + the activation slabs they read / write (name -> (channels, halo))
        geom = (C.c_int * 8)()
        coff, pool = C.c_int(), C.c_int()
        self.layers, self.pools = [], []
        chan, halo = {"x0": c0.value}, {"x0": 0}
        for i in range(nl.value):
            _lib.call("drs_net_layer_info", h, i, name, 96, geom, src, dst, 32, C.byref(coff), C.byref(pool))
            k, rate, cin, cin_k, cout, pb, pa, hl = tuple(geom)
            s_, d_ = src.value.decode(), dst.value.decode()
            self.layers.append(Layer(name.value.decode(), k, cin, cin_k, cout, rate, pb, pa, hl, s_, d_, coff.value))
            q = pool.value
            self.pools.append(None if q == 0 else (("max", 3) if q == 1 else ("avg", q >> 8)))
            chan[d_] = max(chan.get(d_, 0), coff.value + cout)      # a slab is as wide as the slices written into it
            halo[s_] = max(halo.get(s_, 0), hl)                     # and its halo covers every conv that reads it
            halo.setdefault(d_, 0)
        if self.dense:
            chan[self.feat] = self.c_last
        if first_cin_pad != 8 and self.channels <= first_cin_pad:   # (tools on the op-level split-bf16 kernels: one tap per K-step)
            chan["x0"] = round_up(self.channels, first_cin_pad)
            self.layers = [L._replace(cin_k=chan["x0"]) if L.src == "x0" else L for L in self.layers]
        assert halo["x0"] == p0.value
        self.buffers = {n: (chan[n], halo[n]) for n in chan}
        self.pool = any(q is not None and q[0] == "max" for q in self.pools)
        # ---- variables under their TensorFlow scope names: flat layout (every kernel, then every bias; the classifier last in both
        # groups; SE layers after it) and the batch-norm moving statistics (per block mean[C] then variance[C])
        off, cnt, inbn = C.c_size_t(), C.c_size_t(), C.c_int()
        shape = (C.c_int * 4)()
        self.offsets, self.bn_offsets = {}, {}
        for i in range(_lib.query("drs_net_num_variables", h)):
            _lib.call("drs_net_variable_info", h, i, name, 96, C.byref(off), C.byref(cnt), shape, C.byref(inbn))
            n = name.value.decode()
            if inbn.value:
                if n.endswith("/moving_mean"):
                    self.bn_offsets[n.rsplit("/", 1)[0]] = off.value
            else:
                self.offsets[n] = (off.value, tuple(v for v in shape if v))
        # squeeze-and-excitation blocks: block index -> scope (isprs:1042, 1046, 1050)
        self.se = {}
        li = C.c_int()
        for j in range(n_se.value):
            _lib.call("drs_net_se_info", h, j, name, 96, C.byref(li), None, None)
            self.se[li.value] = name.value.decode()
        if self.dense:      # kept for callers that want the slice table (isprs:921-948)
            self.concat_off = [L.dst_coff for L in self.layers]
            self.concat_halo = self.buffers["concat"][1]

    def mac_per_pixel(self):
        return sum(L.k * L.k * L.cin * L.cout for L in self.layers) + self.c_last * self.K
