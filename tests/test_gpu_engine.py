"""-m gpu: the step-level C ABI (drs_net_create / drs_train_step / drs_forward / drs_params_get|set / drs_grad_buffer,
csrc/engine.hip) against the op-level host mirror (net.DilatedNet with engine=False: the same launch sequence spelled out in
Python): BITWISE equal variables, gradients, moving statistics, logits, predictions and confusion matrices over several
training steps and an inference pass, for a net of every wiring (chain + max-pool, plain chain, dense concat, squeeze, SE,
average pool) -- and a pure-ctypes driver that never touches the Python net classes (what a non-Python host would write)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV   # noqa: E402

CASES = [("dilated_grsl_rate8", 5, 6, 3, 33), ("dilated_icpr_original", 3, 6, 2, 25), ("dilated_icpr_rate6_densely", 4, 2, 2, 21),
         ("dilated_icpr_rate6_squeeze", 3, 6, 2, 15), ("dilated_icpr_rate6_SE", 5, 2, 2, 18), ("dilated_icpr_rate6_avgpool", 3, 6, 2, 13),
         ("dilated_grsl", 3, 7, 2, 19)]


@pytest.mark.parametrize("net,ch,K,B,S", CASES)
def test_engine_equals_op_level_path_bitwise(net, ch, K, B, S):
    from drs_amd.net import DilatedNet
    from drs_amd.engine import EngineNet
    a = DilatedNet(net, ch, K, 0.005, b_max=B, s_max=S, device=DEV, seed=7)
    b = DilatedNet(net, ch, K, 0.005, b_max=B, s_max=S, device=DEV, seed=7, engine=False)
    assert isinstance(a, EngineNet) and not isinstance(b, EngineNet)
    assert torch.equal(a.params, b.params) and torch.equal(a.bn, b.bn)
    rng = np.random.default_rng(1)
    masked = K == 7
    for step in range(3):
        s = S if step != 1 else S - 4                     # the patch size changes between steps (isprs:1727-1737)
        x = rng.normal(size=(B, s * s * ch)).astype(np.float32)
        y = rng.integers(0, K, size=(B, s * s))
        m = rng.integers(0, 2, size=(B, s * s)).astype(bool) if masked else None
        outs = []
        for d in (a, b):
            d.feed(x, y, s, mask=m)
            outs.append(d.train_step(B, s, 0.01, use_loss_mask=masked, global_pixels=int(m.sum()) if masked else None, want_logits=True))
        torch.cuda.synchronize()
        M = B * s * s
        for name in ("params", "grads", "mom", "bn"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (name, step)
        assert torch.equal(outs[0]["loss_parts"], outs[1]["loss_parts"]) and torch.equal(outs[0]["conf"], outs[1]["conf"])
        assert torch.equal(outs[0]["pred"], outs[1]["pred"]) and torch.equal(a.logits[:M * K], b.logits[:M * K])
        assert a.loss_value(outs[0]["loss_parts"]) == b.loss_value(outs[1]["loss_parts"])
    assert a.global_step == b.global_step == 3
    x = rng.normal(size=(B, S * S * ch)).astype(np.float32)
    y = rng.integers(0, K, size=(B, S * S))
    for d in (a, b):
        d.feed(x, y, S)
        d.conf.zero_()
    pa, la = a.forward(B, S, labels=True)
    pb, lb = b.forward(B, S, labels=True)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(la, lb) and torch.equal(a.conf, b.conf) and int(a.conf.sum()) == B * S * S
    # variable access by TensorFlow scope name goes through the same buffers
    n0 = a.plan.layers[0].name
    np.testing.assert_array_equal(a.get_variable(n0 + "/weights"), b.get_variable(n0 + "/weights"))
    np.testing.assert_array_equal(a.get_variable("conv_classifier/weights", "Momentum"), b.get_variable("conv_classifier/weights", "Momentum"))


def test_step_level_abi_from_plain_ctypes():
    """A host that is not this package: create, size and bind the buffers, set the variables by name, crop, step, read the results
    -- through the C entry points only (torch supplies device memory, nothing else)."""
    from drs_amd import _lib
    from drs_amd.net import DilatedNet
    lib = _lib.load()
    net_type, ch, K, B, S, wd, lr = b"dilated8_grsl", 5, 6, 2, 24, 0.005, 0.01
    h = C.c_void_p()
    assert lib.drs_net_create(net_type, ch, K, wd, B, S, 1, 0.5, C.byref(h)) == 0
    name, nb, dt = C.create_string_buffer(64), C.c_size_t(), C.c_int()
    bufs = {}
    tdt = {0: torch.float32, 1: torch.float64, 2: torch.uint8, 3: torch.int32}
    for i in range(lib.drs_net_num_buffers(h)):
        assert lib.drs_net_buffer_info(h, i, name, 64, C.byref(nb), C.byref(dt)) == 0
        t = torch.zeros(nb.value // torch.empty(0, dtype=tdt[dt.value]).element_size(), dtype=tdt[dt.value], device=DEV)
        bufs[name.value.decode()] = t
        assert lib.drs_net_bind(h, name.value, t.data_ptr(), nb.value) == 0
    # reference net (op-level path) supplies the initial variables; they travel by TF scope name through drs_params_set
    ref = DilatedNet(net_type.decode(), ch, K, wd, b_max=B, s_max=S, device=DEV, seed=5, engine=False)
    off, cnt, inbn = C.c_size_t(), C.c_size_t(), C.c_int()
    shape = (C.c_int * 4)()
    names = []
    for i in range(lib.drs_net_num_variables(h)):
        lib.drs_net_variable_info(h, i, name, 64, C.byref(off), C.byref(cnt), shape, C.byref(inbn))
        v = np.ascontiguousarray(ref.get_variable(name.value.decode()).reshape(-1))
        assert v.size == cnt.value
        assert lib.drs_params_set(h, name.value, None, v.ctypes.data_as(C.c_void_p), v.size, None) == 0
        names.append(name.value.decode())
    assert "conv8/weights" in names and "conv1/moving_variance" in names and "conv_classifier/biases" in names
    rng = np.random.default_rng(3)
    x = rng.normal(size=(B, S, S, ch)).astype(np.float32)
    y = rng.integers(0, K, size=(B, S, S)).astype(np.uint8)
    # conv1's input slab: [B][S+2P][S+2P][ld] with P, ld from drs_net_layout (a host would fill it with drs_crop_normalize)
    ld, P = C.c_int(), C.c_int()
    lib.drs_net_layout(h, None, None, None, None, C.byref(ld), C.byref(P))
    slab = torch.zeros(B, S + 2 * P.value, S + 2 * P.value, ld.value, device=DEV)
    slab[:, P.value:P.value + S, P.value:P.value + S, :ch] = torch.from_numpy(x).to(DEV)
    bufs["act:x0"][:slab.numel()].copy_(slab.reshape(-1))
    bufs["labels"][:B * S * S].copy_(torch.from_numpy(y.reshape(-1)).to(DEV))
    bufs["acc_mask"].fill_(1)
    st = torch.cuda.current_stream(DEV).cuda_stream
    assert lib.drs_train_step(h, B, S, lr, _lib.USE_ACC_MASK, 0.0, st) == 0
    ref.feed(x.reshape(B, -1), y.reshape(B, -1), S)
    out = ref.train_step(B, S, lr)
    torch.cuda.synchronize()
    sc = bufs["scalars"].cpu().numpy()
    assert float(sc[0] + wd * sc[1]) == ref.loss_value(out["loss_parts"])
    got = np.empty(ref.get_variable("conv5/weights").size, dtype=np.float32)
    assert lib.drs_params_get(h, b"conv5/weights", None, got.ctypes.data_as(C.c_void_p), got.size, st) == 0
    np.testing.assert_array_equal(got.reshape(3, 3, 128, 192), ref.get_variable("conv5/weights"))
    assert lib.drs_params_get(h, b"conv5/weights", b"Momentum", got.ctypes.data_as(C.c_void_p), got.size, st) == 0
    np.testing.assert_array_equal(got.reshape(3, 3, 128, 192), ref.get_variable("conv5/weights", "Momentum"))
    gp, gn = C.c_void_p(), C.c_size_t()
    assert lib.drs_grad_buffer(h, C.byref(gp), C.byref(gn)) == 0 and gp.value == bufs["grads"].data_ptr() and gn.value == ref.plan.n_params
    assert torch.equal(bufs["grads"], ref.grads) and torch.equal(bufs["conf"].view(K, K), out["conf"])
    assert lib.drs_net_global_step(h, -1) == 1
    assert lib.drs_forward(h, B, S, _lib.WANT_LOGITS, -1, st) == 0
    _, lg = ref.forward(B, S)
    torch.cuda.synchronize()
    assert torch.equal(bufs["logits"][:B * S * S * K].view(B, S, S, K), lg)
    assert lib.drs_train_step(h, B + 1, S, lr, 0, 0.0, st) == 1           # outside the allocated (b_max, s_max): rejected
    lib.drs_net_destroy(h)
