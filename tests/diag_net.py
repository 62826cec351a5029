"""Diagnostic (not a test): per-layer error of the HIP training pass against the fp64 oracle, to tell
rounding-induced ReLU / arg-max flips (isolated elements) from systematic error.  Run on the GPU box."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import tf_ops as T
from test_gpu_net import _mk


def main(net="dilated_icpr_original", ch=3, K=6, B=2, S=25):
    o, d, x, y = _mk(net, ch, K, B, S, 11)
    d.debug = {}
    d.feed(x.reshape(B, -1), y.reshape(B, -1), S)
    d.train_step(B, S, 0.01, apply_update=False)
    torch.cuda.synchronize()
    # oracle with per-layer gradients
    x64 = x.astype(np.float64)
    logits = o.forward(x64, True)
    ce, gl = T.softmax_ce(logits, y)
    cache, feat = o.cache
    gcur = gl @ o.p["conv_classifier/weights"][0, 0].T
    for li in reversed(range(len(o.convs))):
        name, k, ci, co, r = o.convs[li]
        inp, z, mean, var, xh, idx, _pos = cache[li]
        if o.spec["dense"]:
            gout, grest = gcur[..., gcur.shape[-1] - co:], gcur[..., :gcur.shape[-1] - co]
        else:
            gout = gcur
        ga = T.max_pool_3x3_bwd(idx, gout) if o.spec["pool"] else gout
        gxh = T.act_bwd(xh, o.spec["act"], ga)
        gz = T.batch_norm_train_bwd(z, mean, var, gxh)
        gin, gw = T.conv2d_same_bwd(inp, o.p[name + "/weights"], r, gz)
        L = d.plan.layers[li]
        got = d.debug["gxh%d" % li].cpu().numpy().reshape(B, S, S, co).astype(np.float64)
        err = np.abs(got - gxh)
        scale = np.abs(gxh).max()
        nbad = int((err > 1e-3 * scale).sum())
        zz = d.z[li][:B * S * S * co].cpu().numpy().reshape(B, S, S, co)
        gw_got = d.get_gradient(name + "/weights")
        print("%-12s gxh max-rel %.2e  elements>1e-3: %d of %d   min|xhat| at bad: %s   z rel %.2e   dW max-rel %.2e  dW l2-rel %.2e" % (
            name, err.max() / scale, nbad, err.size,
            np.array2string(np.abs(xh[err > 1e-3 * scale])[:4], precision=2) if nbad else "-",
            np.abs(zz - z).max() / np.abs(z).max(), np.abs(gw_got - gw).max() / np.abs(gw).max(),
            np.linalg.norm(gw_got - gw) / np.linalg.norm(gw)))
        if o.spec["dense"]:
            gcur = (grest + gin) if li > 0 else None
        else:
            gcur = gin


if __name__ == "__main__":
    main()
    main("dilated8_grsl", 5, 6, 2, 26)
