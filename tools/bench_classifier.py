import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drs_amd import _lib
_lib.load()
DEV="cuda:0"; B,S,C,K=128,64,256,6; M=B*S*S
st = torch.cuda.current_stream(DEV).cuda_stream
feat = torch.randn(M*C, device=DEV); w = torch.randn(C*K, device=DEV)*0.05; b = torch.zeros(K, device=DEV)
lab = torch.randint(0, K, (M,), dtype=torch.uint8, device=DEV)
pred = torch.zeros(M, dtype=torch.uint8, device=DEV); g = torch.zeros(M*C, device=DEV)
rows = _lib.query("drs_classifier_rows", B, S)
dw = torch.zeros(rows*C*K, device=DEV); db = torch.zeros(rows*K, device=DEV); lp = torch.zeros(rows, dtype=torch.float64, device=DEV)
conf = torch.zeros(K*K, dtype=torch.int32, device=DEV)
def run(cf, gf=True, lb=True):
    _lib.call("drs_classifier_loss", feat.data_ptr(), B, S, 0, C, 0, C, K, w.data_ptr(), b.data_ptr(), lab.data_ptr() if lb else None, None, None,
              1.0/M, None, pred.data_ptr(), g.data_ptr() if gf else None, C, 0, dw.data_ptr() if gf else None, db.data_ptr() if gf else None, lp.data_ptr() if lb else None,
              conf.data_ptr() if cf else None, st)
def t(fn):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
print("train with conf    %.3f ms" % t(lambda: run(True)))
print("train without conf %.3f ms" % t(lambda: run(False)))
print("loss only (no gfeat) %.3f ms" % t(lambda: run(False, gf=False)))
print("inference (no labels) %.3f ms" % t(lambda: run(False, gf=False, lb=False)))
