#!/usr/bin/env python3
"""In-process A/B of the classifier block (drs_classifier_loss): MFMA form against the vector-ALU form (libdrs_hip_dev.so,
drs_debug_cls_variant), training / inference, with the algorithmic bytes (feature read + feature-gradient write) against the
8 TB/s of the HBM.    python tools/bench_classifier.py [B=128,16] [S=64,25] [C=256] [K=6]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drs_amd import _lib  # noqa: E402
_lib = _lib.dev()
DEV = "cuda:0"


def t(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def main(Bs, Ss, C, K):
    st = torch.cuda.current_stream(DEV).cuda_stream
    for B in Bs:
        for S in Ss:
            M = B * S * S
            feat = torch.randn(M * C, device=DEV)
            w = torch.randn(C * K, device=DEV) * 0.05
            b = torch.zeros(K, device=DEV)
            lab = torch.randint(0, K, (M,), dtype=torch.uint8, device=DEV)
            pred = torch.zeros(M, dtype=torch.uint8, device=DEV)
            g = torch.zeros(M * C, device=DEV)
            rows = _lib.query("drs_classifier_rows", B, S)
            dw = torch.zeros(rows * C * K, device=DEV)
            db = torch.zeros(rows * K, device=DEV)
            lp = torch.zeros(rows, dtype=torch.float64, device=DEV)
            conf = torch.zeros(K * K, dtype=torch.int32, device=DEV)

            def run(train):
                _lib.call("drs_classifier_loss", feat.data_ptr(), B, S, 0, C, 0, C, K, w.data_ptr(), b.data_ptr(), lab.data_ptr() if train else None, None, None,
                          1.0 / M, None, pred.data_ptr(), g.data_ptr() if train else None, C, 0, dw.data_ptr() if train else None,
                          db.data_ptr() if train else None, lp.data_ptr() if train else None, conf.data_ptr() if train else None, st)
            res = {}
            for rep in range(3):
                for v in (0, 2, 3):
                    _lib.drs_debug_cls_variant(v)
                    for train in (True, False):
                        res[(v, train)] = min(res.get((v, train), 1e9), t(lambda: run(train)))
            _lib.drs_debug_cls_variant(1)
            for train in (True, False):
                byt = M * C * 4 * (2 if train else 1)
                print("B=%3d S=%3d C=%d K=%d %-9s " % (B, S, C, K, "training" if train else "inference") +
                      "   ".join("%s %.4f ms (%.2f TB/s = %.3f of 8)" % (n, res[(v, train)], byt / res[(v, train)] / 1e9, byt / res[(v, train)] / 8e9)
                                 for v, n in ((0, "vector-ALU"), (2, "MFMA reg"), (3, "MFMA LDS-DMA"))), flush=True)


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main([int(v) for v in kw.get("B", "128,16").split(",")], [int(v) for v in kw.get("S", "64,25").split(",")], int(kw.get("C", 256)), int(kw.get("K", 6)))
