import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drs_amd.net import DilatedNet
from drs_amd import patches as P
from drs_amd.synthetic import make_tile, grid_instances
B, S = 16, 64
dev = "cuda:0"
tile, lab = make_tile(1024, 1024, 5, 6, seed=1234)
pool = P.TilePool([tile], [lab], dev)
inst = grid_instances(1024, 1024, S, 25, 4096, seed=0)
net = DilatedNet("dilated_grsl_rate8", 5, 6, 0.005, b_max=B, s_max=S, device=dev)
np.random.seed(0)
def step(i):
    rows = inst[(i * B) % 4000:(i * B) % 4000 + B]
    aug = P.draw_augmentation(rows, S, 5, noise="device")
    P.crop_to_net(net, pool, rows, S, [0.5] * 3, [0.2] * 3, aug)
    return net.train_step(B, S, 0.01)
for i in range(5): step(i)
ts = []
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(i); ts.append(time.perf_counter() - t0)
print("host enqueue per step with an empty queue: median %.2f ms" % (1e3 * np.median(ts)))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    torch.cuda.synchronize(); step(i)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
