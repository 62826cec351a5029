#!/usr/bin/env python3
"""The three elementwise passes of a block (batch-norm + activation + 3x3 pool forward; its backward in two passes) in isolation,
on the channel counts of Dilated8Pooling: time per launch and the fraction of 8 TB/s on the algorithmic bytes
(forward M*C*9, backward reduce M*C*13, backward apply M*C*12).  Development aid.

    python tools/bench_elementwise.py B=128 S=64 [C=64,128,192,256] [P=8] [blocks=4096 minrows=4]   (the last two: libdrs_hip_dev.so)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main(B, S, Cs, P, blocks=None, minrows=None, rowpad=0):
    global _lib
    if blocks is not None or minrows is not None or rowpad:
        _lib = _lib.dev()
        if blocks is not None:
            _lib.drs_debug_slide_blocks(blocks)
        if minrows is not None:
            _lib.drs_debug_slide_minrows(minrows)
        if rowpad:
            _lib.drs_debug_slide_rowpad(rowpad)      # timing experiment: the sliding kernels walk rows of S + rowpad pixels (apply stays linear)
        print("slide blocks target %s, min rows %s, row pad %d" % (blocks, minrows, rowpad))
    st = torch.cuda.current_stream(DEV).cuda_stream
    M = B * S * S
    MP = B * S * (S + rowpad)
    tot = {"fwd": 0.0, "reduce": 0.0, "apply": 0.0}
    for C in Cs:
        z = torch.randn(MP, C, device=DEV)
        mr = torch.stack([torch.zeros(C, device=DEV), torch.ones(C, device=DEV)], 1).contiguous()
        Sp = S + 2 * P
        out = torch.zeros(B * Sp * Sp * C, device=DEV)
        idx = torch.zeros(MP * C, dtype=torch.uint8, device=DEV)
        ga = torch.randn(MP, C, device=DEV)
        gxh = torch.empty(MP, C, device=DEV)
        rows = _lib.query("drs_bn_backward_rows", B, S, C, 1)
        partial = torch.zeros(rows * C * 2, device=DEV)
        sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
        gz = torch.zeros(B * Sp * Sp * C, device=DEV)
        f = lambda: _lib.call("drs_bn_act_pool_forward", z.data_ptr(), B, S, C, mr.data_ptr(), 0.1, 1, out.data_ptr(), P, C, 0, idx.data_ptr(), st)
        r = lambda: _lib.call("drs_bn_backward_reduce", ga.data_ptr(), C, 0, z.data_ptr(), idx.data_ptr(), B, S, C, mr.data_ptr(), 0.1, 1,
                              gxh.data_ptr(), partial.data_ptr(), st)
        a = lambda: _lib.call("drs_bn_backward_apply", gxh.data_ptr(), z.data_ptr(), B, S, C, mr.data_ptr(), sums.data_ptr(), float(M), gz.data_ptr(),
                              P, C, 0, st)
        res = {}
        for name, fn, bpe in (("fwd", f, 9), ("reduce", r, 13), ("apply", a, 12)):
            ms = timeit(fn)
            res[name] = (ms, M * C * bpe / ms / 1e6 / 8000.0)
            tot[name] += ms
        print("B=%d S=%d C=%3d  " % (B, S, C) + "  ".join("%s %.4f ms (%.3f)" % (k, v[0], v[1]) for k, v in res.items()), flush=True)
    print("sum over layers: " + "  ".join("%s %.3f ms" % kv for kv in tot.items()) + "   all %.3f ms" % sum(tot.values()), flush=True)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(int(kw.get("B", 128)), int(kw.get("S", 64)), [int(v) for v in kw.get("C", "64,64,128,128,192,192,256,256").split(",")], int(kw.get("P", 8)),
         int(kw["blocks"]) if "blocks" in kw else None, int(kw["minrows"]) if "minrows" in kw else None, int(kw.get("rowpad", 0)))
