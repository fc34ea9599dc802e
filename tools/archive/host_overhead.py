"""wall time of Model.trainingstep per call vs the device time of the step (HIP events):
what the synchronous reference API (loss returned every step) costs on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from elektronn2_amd import nets
np.random.seed(0)
m = nets.neuro3d_lite((None, 1, 23, 183, 183))
m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
x = torch.rand(1, 1, 23, 183, 183, device="cuda")
t = torch.randint(0, 2, (1, 1, 10, 37, 37), device="cuda").float()
for _ in range(10):
    m.trainingstep(x, t, optimiser='Adam')
n = 500
t0 = time.perf_counter(); dev = 0.0
for _ in range(n):
    loss, tsec, _ = m.trainingstep(x, t, optimiser='Adam')
    dev += tsec
wall = (time.perf_counter() - t0) / n
print("trainingstep: wall %.3f ms per call, device %.3f ms, host-serial %.3f ms" % (wall * 1e3, dev / n * 1e3, (wall - dev / n) * 1e3))
xn, tn = x.cpu().numpy(), t.cpu().numpy()
t0 = time.perf_counter()
for _ in range(n):
    m.trainingstep(xn, tn, optimiser='Adam')
print("with numpy inputs (3.1 MB upload per step): wall %.3f ms per call" % ((time.perf_counter() - t0) / n * 1e3))
