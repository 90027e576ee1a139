#!/usr/bin/env python3
"""Per-task durations of the fast decoder (needs the -DLDPC_TIMING build as NRLDPC_HIP_LIB): pass 2 of block 0."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import openairinterface5g_amd as pkg
m = pkg.ldpc
pkg.LDPCinit()
BG, Z, R, n = 1, 384, 13, 1024
llr = torch.randint(-128, 128, (n, 68 * Z), dtype=torch.int8, device="cuda")
llr[:, :2 * Z] = 0
out = torch.zeros((n, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
it = torch.zeros(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8)
torch.cuda.synchronize()
d = out[0].cpu().numpy().view(np.int32)
cn = d[:256].reshape(16, 16)
bn = d[256:256 + 128].reshape(16, 8)
by = {}
for w in range(16):
    row = [(int(v) >> 24, int(v) & 0xffffff) for v in cn[w] if v]
    print("wave %2d CN:" % w, " ".join("%d:%d" % x for x in row), " total", sum(x[1] for x in row),
          "| BN:", " ".join("%d:%d" % (int(v) >> 24, int(v) & 0xffffff) for v in bn[w] if v), " total", sum(int(v) & 0xffffff for v in bn[w]))
    for dg, t in row:
        by.setdefault(dg, []).append(t)
for dg in sorted(by):
    print("CN deg %2d: n=%2d mean %.0f  per edge %.1f" % (dg, len(by[dg]), np.mean(by[dg]), np.mean(by[dg]) / dg))
byb = {}
for w in range(16):
    for v in bn[w]:
        if v:
            byb.setdefault(int(v) >> 24, []).append(int(v) & 0xffffff)
for dg in sorted(byb):
    print("BN maxdeg %2d: n=%2d mean %.0f  per edge %.1f" % (dg, len(byb[dg]), np.mean(byb[dg]), np.mean(byb[dg]) / dg))
