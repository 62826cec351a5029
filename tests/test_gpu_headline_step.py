"""-m gpu: the step bench.py's headline times -- dilated_grsl_rate8 (Dilated8Pooling, isprs:996-1033), 128 patches of 64 x 64 x 5 of
a 2048 x 2048 tile, through the product library's step engine (drs_train_step, libdrs_hip.so) -- held to the oracle launch by launch;
and the same for BASELINE configs[1] (Dilated6Pooling, batch 64 of 64 x 64) and configs[3]'s net (DenseDilated6, batch 128 of 75 x 75 x 4).

The fp64 oracle of a whole step at this size does not fit a test (2.2 TFLOP forward, ~30 GB of fp64 activations), and train-mode
batch norm couples all 128 patches, so the comparison is TEACHER-FORCED: every launch of the step is checked on ITS OWN operands as
the engine left them in its bound buffers -- the convolutions, the normalise + activation + pool passes, the classifier, the
arg-max and the input gradient of the classifier on sampled patches against oracle/tf_ops.py (fp64); what sums over the whole batch
(batch-norm moments, loss, confusion matrix, the classifier's and conv1's filter gradients, the L2 term) against fp64 sums taken on
the device with plain torch ops.  Together with tests/test_gpu_configs.py (every Dilated8Pooling layer's forward / input gradient /
filter gradient at 128 x 64 x 64 on random operands, same binary) and the invariants of tests/test_gpu_fullsize.py (pool / batch-norm
backward at full size) this covers every kernel of the timed step at the timed shape."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

from gpu_util import DEV, rel_err   # noqa: E402

WD, LR = 0.005, 0.01
D8P_LAYERS = [(5, 1, 5, 64), (5, 2, 64, 64), (4, 3, 64, 128), (4, 4, 128, 128), (3, 5, 128, 192), (3, 6, 192, 192), (3, 7, 192, 256), (3, 8, 256, 256)]      # isprs:1000-1021

# the headline (BASELINE.json metric / configs[2]'s net at the metric's shape), configs[1] at its own shape (Dilated6Pooling, isprs:962-993),
# configs[3]'s net at one of the sides `multinomial` lists (DenseDilated6: ReLU, no pool, every block a slice of the 448-wide concat slab,
# conv1 / conv2 on the 256 x 32 register tile, input gradients ACCUMULATED into the concat gradient; isprs:914-959)
CASES = [("dilated_grsl_rate8", 5, 6, 128, 64, 2048), ("dilated_grsl", 5, 6, 64, 64, 2048), ("dilated_icpr_rate6_densely", 4, 2, 128, 75, 500)]


@pytest.mark.parametrize("NET,CH,K,B,S,TILE", CASES, ids=["headline-dilated8-128x64", "config2-dilated6pooling-64x64", "config4-dense-128x75"])
def test_every_launch_of_the_step_on_its_own_operands(NET, CH, K, B, S, TILE):
    from drs_amd import _lib, patches as P
    from drs_amd.net import DilatedNet
    from drs_amd.synthetic import grid_instances, make_tile
    _lib.load()
    SAMPLE = (0, B // 2 - 3, B - 1)

    def _interior(slab, C, P):
        return slab[:B * (S + 2 * P) ** 2 * C].view(B, S + 2 * P, S + 2 * P, C)[:, P:P + S, P:P + S, :]
    tile, lab = make_tile(TILE, TILE, CH, K, seed=1234)
    pool = P.TilePool([tile], [lab], DEV, dtype=np.float64)
    inst = grid_instances(TILE, TILE, S, 25, B * 4, seed=0)
    mean, std = tile[:, :, :3].mean(axis=(0, 1)).tolist(), tile[:, :, :3].std(axis=(0, 1)).tolist()
    net = DilatedNet(NET, CH, K, WD, b_max=B, s_max=S, device=DEV, seed=42)
    assert type(net).__name__ == "EngineNet" and net.plan.net_type == NET
    np.random.seed(3)
    for i in range(3):                                   # two real steps first: variables, momentum and moving statistics off their initial values
        rows = inst[i * B:(i + 1) * B]
        P.crop_to_net(net, pool, rows, S, mean, std, P.draw_augmentation(rows, S, CH, noise="device"))
        out = net.train_step(B, S, LR, apply_update=(i < 2), want_logits=True)
    torch.cuda.synchronize()
    M = B * S * S
    layers = net.plan.layers
    if NET == "dilated_grsl_rate8":
        assert [(L.k, L.rate, L.cin, L.cout) for L in layers] == D8P_LAYERS
    dense = net.plan.dense
    kind = "lrelu" if net.plan.alpha > 0 else "relu"
    assert (kind, dense) == {"dilated_grsl_rate8": ("lrelu", False), "dilated_grsl": ("lrelu", False), "dilated_icpr_rate6_densely": ("relu", True)}[NET]
    sample = list(SAMPLE)
    for i, L in enumerate(layers):
        Cin_slab, Pin = net.plan.buffers[L.src]
        Cout_slab, Pout = net.plan.buffers[L.dst]
        x_in = _interior(net.abuf[L.src], Cin_slab, Pin)
        # the halo of every slab a convolution reads is zero (the inner loops have no bounds checks: DESIGN 2)
        full = net.abuf[L.src][:B * (S + 2 * Pin) ** 2 * Cin_slab].view(B, S + 2 * Pin, S + 2 * Pin, Cin_slab)
        if Pin:
            assert float(full[:, :Pin].abs().max()) == 0 and float(full[:, -Pin:].abs().max()) == 0
            assert float(full[:, :, :Pin].abs().max()) == 0 and float(full[:, :, -Pin:].abs().max()) == 0
        w = net.get_variable(L.name + "/weights").astype(np.float64)
        bias = net.get_variable(L.name + "/biases").astype(np.float64)
        z = net.z[i][:M * L.cout].view(B, S, S, L.cout)
        # (1) the convolution (+ bias) on the sampled patches: conv1 = the packed-tap kernel on 5 of 8 padded bands, conv3 the 4|5 pad
        xs = x_in[sample][..., :L.cin].cpu().numpy().astype(np.float64)
        if L.cin_k > L.cin:
            assert float(x_in[..., L.cin:].abs().max()) == 0              # the padded bands are zero (drs_crop_normalize)
        ref = T.conv2d_same(xs, w, L.rate) + bias
        assert rel_err(z[sample].cpu().numpy(), ref) < 1e-5, L.name
        # (2) the batch's moments: the conv epilogue's per-tile (sum, M2) slabs + the Chan combination, against fp64 on the device
        mr = net.mean_rstd[i].view(L.cout, 2).double()
        z64 = z.reshape(M, L.cout).double()
        mu, var = z64.mean(0), z64.var(0, unbiased=False)
        assert float((mr[:, 0] - mu).abs().max()) <= 1e-6 * float(mu.abs().max()) + 1e-7, L.name
        assert float((mr[:, 1] / (var + 1e-3).rsqrt() - 1).abs().max()) < 2e-6, L.name                 # eps 0.001: contrib batch_norm (isprs:658)
        # (3) normalise + (leaky) ReLU [+ 3 x 3 / stride-1 max-pool] (isprs:715-721, 745-746) on the sampled patches, given the engine's moments
        zs = z[sample].cpu().numpy().astype(np.float64)
        xhat = (zs - mr[:, 0].cpu().numpy()) * mr[:, 1].cpu().numpy()
        act = T.act_fwd(xhat, kind)
        got = _interior(net.abuf[L.dst], Cout_slab, Pout)[sample][..., L.dst_coff:L.dst_coff + L.cout].cpu().numpy()
        if net.plan.pools[i] is None:
            assert rel_err(got, act) < 1e-5, L.name
            continue
        pooled, idx = T.max_pool_3x3(act)
        assert rel_err(got, pooled) < 1e-5, L.name
        # arg-max codes: equal wherever the winner is clear in fp64 (first maximum in scan order on ties, as TF's MaxPoolGrad)
        didx = net.idx[i][:M * L.cout].view(B, S, S, L.cout)[sample].cpu().numpy()
        forced, _ = T.max_pool_3x3(act, forced_idx=didx)
        assert np.abs(forced - pooled).max() <= 1e-5 * np.abs(pooled).max(), L.name            # the device's winner IS a maximum (to rounding)
        assert float((didx != idx).mean()) < 2e-3, L.name
    # ---- classifier, loss, arg-max, confusion matrix (isprs:1024-1031, 1089-1099, 1690, 510-531)
    Cf, Pf = net.plan.buffers[net.plan.feat]
    feat = _interior(net.abuf[net.plan.feat], Cf, Pf).reshape(M, Cf).double()
    wc = torch.from_numpy(net.get_variable("conv_classifier/weights").reshape(Cf, K)).to(DEV).double()
    bc = torch.from_numpy(net.get_variable("conv_classifier/biases")).to(DEV).double()
    lg = feat @ wc + bc
    logits = net.logits[:M * K].view(M, K)
    assert float((logits.double() - lg).abs().max()) <= 1e-5 * float(lg.abs().max())
    labels = net.labels[:M].long()
    ce = torch.nn.functional.cross_entropy(lg, labels, reduction="mean").item()
    l2 = 0.0
    for L in layers:
        l2 += 0.5 * float((net.get_variable(L.name + "/weights").astype(np.float64) ** 2).sum())
    l2 += 0.5 * float((net.get_variable("conv_classifier/weights").astype(np.float64) ** 2).sum())
    parts = out["loss_parts"].cpu().numpy()
    assert abs(parts[0] - ce) < 1e-6 * ce and abs(parts[1] - l2) < 1e-6 * l2
    assert abs(net.loss_value(out["loss_parts"]) - (ce + WD * l2)) < 1e-6 * (ce + WD * l2)            # isprs:646-651: wd * l2_loss per kernel
    srt = lg.sort(1).values
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4 * float(lg.abs().max())
    pred = out["pred"].reshape(-1).long()
    assert float(clear.double().mean()) > 0.99 and torch.equal(pred[clear], lg.argmax(1)[clear])
    mask = net.acc_mask[:M].bool()
    want = torch.zeros(K, K, dtype=torch.int64, device=DEV)
    want.index_put_((labels[mask], pred[mask]), torch.ones(int(mask.sum()), dtype=torch.int64, device=DEV), accumulate=True)
    assert torch.equal(out["conf"].long(), want)                                            # integer atomics: exact
    # ---- backward, what the buffers still hold after the step
    dl = (torch.softmax(lg, 1) - torch.nn.functional.one_hot(labels, K)) / M                # d(mean CE) / d(logits)
    if not dense:           # (the dense net's concat gradient has by now ACCUMULATED every later block's input gradient on top of the classifier's)
        gfeat = net.gbuf[net.plan.feat][:M * Cf].view(M, Cf)
        ref_g = dl @ wc.t()
        assert float((gfeat.double() - ref_g).abs().max()) <= 1e-5 * float(ref_g.abs().max())
    gw = torch.from_numpy(net.get_gradient("conv_classifier/weights").reshape(Cf, K)).to(DEV).double()
    ref_w = feat.t() @ dl
    assert float((gw - ref_w).abs().max()) <= 1e-5 * float(ref_w.abs().max())
    gb = torch.from_numpy(net.get_gradient("conv_classifier/biases")).to(DEV).double()
    assert float((gb - dl.sum(0)).abs().max()) <= 1e-5 * float(dl.sum(0).abs().max())
    for L in layers:        # biases in front of a mean-subtracting batch norm: gradient identically zero
        assert not net.get_gradient(L.name + "/biases").any()
    # conv1's filter gradient = sum over ALL pixels of x0[p + tap] (x) gz[p]: the last filter gradient of the step; `gz` still holds its
    # operand (one-stream backward pass at this size), haloed by conv1's 2 pixels
    L0 = layers[0]
    gz = net.gz[:B * (S + 2 * L0.halo) ** 2 * L0.cout].view(B, S + 2 * L0.halo, S + 2 * L0.halo, L0.cout)[:, L0.halo:L0.halo + S, L0.halo:L0.halo + S, :].double()
    C0, P0 = net.plan.buffers["x0"]
    x0 = net.abuf["x0"][:B * (S + 2 * P0) ** 2 * C0].view(B, S + 2 * P0, S + 2 * P0, C0).double()
    ref0 = torch.zeros(L0.k, L0.k, L0.cin, L0.cout, dtype=torch.float64, device=DEV)
    for u in range(L0.k):
        for v in range(L0.k):
            oy, ox = P0 - L0.pad_b + u * L0.rate, P0 - L0.pad_b + v * L0.rate
            ref0[u, v] = torch.einsum("byxc,byxo->co", x0[:, oy:oy + S, ox:ox + S, :L0.cin], gz)
    g0 = torch.from_numpy(net.get_gradient(L0.name + "/weights")).to(DEV).double()
    assert float((g0 - ref0).abs().max()) <= 1e-5 * float(ref0.abs().max())
    # the input gradient of conv2 (the last input-gradient launch: gact of conv1's output slab) on the sampled patches needs conv2's gz,
    # which `gz` no longer holds: covered on random operands at this shape by tests/test_gpu_configs.py
    assert M >= (1 << 18)         # (one-stream backward pass: `gz` is the only output-gradient slab, and conv1's was the last written)
