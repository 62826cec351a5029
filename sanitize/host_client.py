"""Runs under LD_PRELOAD=libclang_rt.asan (sanitize/test_host_asan.py): the HOST code of the library -- net tables, variable and
buffer layout of every net type, the filter-gradient cut (wgrad_assign / WgradPlan), the stream-K geometry queries, argument
validation -- through libdrs_hip_asan.so, the CPU-only AddressSanitizer + UBSan build (no device code, never used on a GPU box).
No torch import here: the sanitizer runtime and PyTorch's allocator hooks do not mix, and nothing below needs a device."""
import ctypes as C
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
lib = C.CDLL(os.environ["DRS_ASAN_LIB"])
lib.drs_net_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_void_p)]
lib.drs_net_destroy.argtypes = [C.c_void_p]
lib.drs_net_num_buffers.argtypes = [C.c_void_p]
lib.drs_net_buffer_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
lib.drs_net_num_variables.argtypes = [C.c_void_p]
lib.drs_net_variable_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.drs_net_layout.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 3 + [C.POINTER(C.c_int)] * 3
lib.drs_net_bind.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
lib.drs_train_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_double, C.c_void_p]
lib.drs_net_global_step.argtypes = [C.c_void_p, C.c_longlong]
lib.drs_net_global_step.restype = C.c_longlong
lib.drs_net_learning_rate.argtypes = [C.c_void_p, C.c_float]
lib.drs_net_learning_rate.restype = C.c_float
lib.drs_conv_workspace_floats.restype = C.c_size_t
lib.drs_debug_wgrad_cut.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]

NETS = ["dilated_icpr_original", "dilated_grsl", "dilated_grsl_rate8", "dilated8_grsl", "dilated_grsl_old", "dilated_icpr_rate6", "dilated_icpr_rate6_small",
        "dilated_icpr_rate6_nodilation", "dilated_icpr_rate1", "dilated_icpr_vary_rate", "dilated_icpr_old", "dilated_icpr_rate6_avgpool",
        "dilated_icpr_rate6_squeeze", "dilated_icpr_rate6_SE", "dilated_icpr_rate6_densely"]
checked = 0
for nt in NETS:
    for ch, K, b, s in ((3, 2, 4, 25), (5, 6, 16, 85), (4, 7, 2, 100)):
        h = C.c_void_p()
        assert lib.drs_net_create(nt.encode(), ch, K, 0.005, b, s, 1, 0.5, C.byref(h)) == 0, nt
        name = C.create_string_buffer(8)                 # deliberately short: the copies must truncate, not overrun
        long_name = C.create_string_buffer(64)
        nb, dt = C.c_size_t(), C.c_int()
        for i in range(lib.drs_net_num_buffers(h)):
            assert lib.drs_net_buffer_info(h, i, name, 8, C.byref(nb), C.byref(dt)) == 0 and len(name.value) <= 7
            assert lib.drs_net_buffer_info(h, i, long_name, 64, C.byref(nb), C.byref(dt)) == 0 and nb.value > 0
        assert lib.drs_net_buffer_info(h, -1, long_name, 64, None, None) == 1 and lib.drs_net_buffer_info(h, 10 ** 6, long_name, 64, None, None) == 1
        off, cnt, shape, inbn = C.c_size_t(), C.c_size_t(), (C.c_int * 4)(), C.c_int()
        total = 0
        for i in range(lib.drs_net_num_variables(h)):
            assert lib.drs_net_variable_info(h, i, long_name, 64, C.byref(off), C.byref(cnt), shape, C.byref(inbn)) == 0
            assert lib.drs_net_variable_info(h, i, name, 8, None, None, None, None) == 0
            total += cnt.value if not inbn.value else 0
        np_, nd, nbn, nl, c0, p0 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        assert lib.drs_net_layout(h, C.byref(np_), C.byref(nd), C.byref(nbn), C.byref(nl), C.byref(c0), C.byref(p0)) == 0
        assert total == np_.value and nd.value <= np_.value and nl.value >= 3
        assert lib.drs_net_bind(h, b"no such buffer", C.c_void_p(16), 16) == 1
        assert lib.drs_train_step(h, b, s, 0.01, 0, 0.0, None) == 1          # nothing bound: rejected before any launch
        assert lib.drs_net_global_step(h, 123456) == 123456 and abs(lib.drs_net_learning_rate(h, 0.01) - 0.0025) < 1e-9
        lib.drs_net_destroy(h)
        checked += 1
h = C.c_void_p()
assert lib.drs_net_create(b"no_such_net", 5, 6, 0.005, 4, 25, 1, 0.5, C.byref(h)) == 1 and not h.value
assert lib.drs_net_create(b"dilated_grsl", 5, 6, 0.005, 4096, 4096, 1, 0.5, C.byref(h)) == 1

# the filter-gradient cut, worked out on the host by the code the kernels run
import array
cuts = 0
for (B, S) in ((16, 25), (16, 64), (128, 64), (128, 85), (3, 9), (1, 100), (32, 33)):
    for (k, rate, pad, cin, cout) in ((3, 8, 8, 256, 256), (4, 3, 4, 64, 128), (5, 2, 4, 64, 64), (3, 6, 6, 192, 192), (5, 1, 2, 8, 64), (3, 6, 6, 320, 128),
                                      (5, 2, 4, 32, 32), (1, 1, 0, 64, 32)):
        cap = 1 << 14
        out = (C.c_int * (5 * cap))()
        nt_ = (C.c_int * 256)()
        tr = C.c_int()
        n = lib.drs_debug_wgrad_cut(B, S, k, rate, pad, cin, cout, out, cap, nt_, C.byref(tr))
        assert n > 0 and lib.drs_conv_wgrad_splits(B, S, k, cin, cout) >= 1
        n2 = lib.drs_debug_wgrad_cut(B, S, k, rate, pad, cin, cout, out, 7, nt_, C.byref(tr))      # a cap below the count: nothing past it
        assert n2 == n
        cuts += 1
assert lib.drs_debug_wgrad_cut(0, 0, 3, 1, 1, 64, 64, None, 0, None, None) < 0
for cout in (32, 64, 128, 192, 256, 448):
    assert lib.drs_conv_workspace_floats(cout) % (2 * 128) == 0
for B, S, C_, pool in ((1, 1, 64, 1), (128, 64, 256, 1), (16, 85, 192, 0), (2, 100, 448, 0)):
    assert lib.drs_bn_backward_rows(B, S, C_, pool) >= 1 and lib.drs_classifier_rows(B, S) >= 1
print("sanitized host run ok: %d nets, %d cuts" % (checked, cuts))
