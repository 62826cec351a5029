#!/usr/bin/env python3
"""How much do the gradients of one training step move when the variables are perturbed by a relative eps?  (Context for
tools/fuzz_dp.py: two data-parallel ranks and one process differ by ~3e-5 in the variables after one step -- sums associated
differently flip a few ReLU signs / pool winners -- and by ~4e-2 in the next step's gradients; is that the net or a bug?)
    python tools/sensitivity.py [net=dilated8_grsl] [ch=3] [K=2] [B=3] [S=29]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd.net import DilatedNet


def main(net="dilated8_grsl", ch=3, K=2, B=3, S=29):
    rng = np.random.default_rng(B * 100 + S)
    x = rng.normal(size=(B, S * S * ch)).astype(np.float32)
    y = rng.integers(0, K, size=(B, S * S))
    def grads(eps, seed):
        d = DilatedNet(net, ch, K, 0.005, b_max=B, s_max=S, device="cuda:0", seed=3)
        if eps:
            g = torch.Generator(device="cuda:0").manual_seed(seed)
            d.params.add_(eps * d.params.abs().max() * torch.randn(d.params.shape, device="cuda:0", generator=g))
        d.feed(x, y, S)
        d.train_step(B, S, 0.01, apply_update=False)
        torch.cuda.synchronize()
        return d.grads.cpu().numpy().astype(np.float64)
    g0 = grads(0, 0)
    for eps in (1e-7, 1e-6, 1e-5, 3e-5, 1e-4):
        e = [float(np.abs(grads(eps, s) - g0).max() / np.abs(g0).max()) for s in (1, 2, 3)]
        print("%s B=%d S=%d: variables perturbed by %.0e (relative to the largest) -> gradients move by %s (relative to the largest)" % (net, B, S, eps, ["%.1e" % v for v in e]))


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(kw.get("net", "dilated8_grsl"), int(kw.get("ch", 3)), int(kw.get("K", 2)), int(kw.get("B", 3)), int(kw.get("S", 29)))
