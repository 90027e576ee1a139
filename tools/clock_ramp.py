"""Per-step kernel times of the headline launch right after set-up (driver's K=20, W=5) and after sustained work."""
import sys, time, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench as B
import openairinterface5g_amd as pkg
pkg.LDPCinit()
_, llr = B.make_batch(pkg, torch, -12.0, 1000)
out = torch.zeros((B.BATCH, B.NUM_LLR // 8), dtype=torch.uint8, device="cuda")
nit = torch.zeros(B.BATCH, dtype=torch.int32, device="cuda")
def run(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev[0].record()
    for i in range(n):
        pkg.decode_batch_device(B.BG, B.Z, B.R, llr, out, nit, numMaxIter=B.MAX_ITER); ev[i + 1].record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt, [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
dt, k = run(25)
print("cold: 25 steps wall %.3f ms/step; per-step ms:" % (dt / 25 * 1e3), " ".join("%.3f" % x for x in k))
for rep in range(3):
    dt, k = run(500)
    k = np.array(k)
    print("sustained run %d: wall %.4f ms/step; chunks of 50: %s" % (rep, dt / 500 * 1e3, " ".join("%.4f" % k[i:i + 50].mean() for i in range(0, 500, 50))))
time.sleep(2.0)
dt, k = run(25)
print("after 2 s idle: wall %.3f; per-step:" % (dt / 25 * 1e3), " ".join("%.3f" % x for x in k))
