#!/usr/bin/env python3
"""Timeline of the fast decoder's tasks (needs the -DLDPC_TIMING build, tools/task_timing.sh, as NRLDPC_HIP_LIB):
pass 2 of block 0, per wave: phase, degree, start and end in shader clocks since the kernel started."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import openairinterface5g_amd as pkg
m = pkg.ldpc
pkg.LDPCinit()
BG, Z, R = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 384, 13)))
n = 1024
llr = torch.randint(-128, 128, (n, m.NCOLS[(BG, R)] * Z), dtype=torch.int8, device="cuda")
llr[:, :2 * Z] = 0
out = torch.zeros((n, max(m.out_bytes(BG, Z, R), 16 * 12 * 16)), dtype=torch.uint8, device="cuda")
it = torch.zeros(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    out[0].zero_()
    pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8, kernel=3)
torch.cuda.synchronize()
d = out[0].cpu().numpy().view(np.int64)[:16 * 12 * 2].reshape(16, 12, 2)
ends = {0: [], 1: []}
first = None
for w in range(16):
    row = []
    for k in range(12):
        a, e = int(d[w, k, 0]), int(d[w, k, 1])
        if a == 0 and e == 0:
            continue
        start, ph, deg = a >> 20, (a >> 8) & 0xf, a & 0xff
        first = start if first is None else min(first, start)
        row.append((ph, deg, start, e))
    if row:
        print("wave %2d:" % w, "  ".join("%s%d[%d..%d]" % ("CB"[ph], deg, s - first, e - first) for ph, deg, s, e in row))
        for ph in (0, 1):
            es = [e for p_, _, _, e in row if p_ == ph]
            if es:
                ends[ph].append(max(es) - first)
d0 = out[0].cpu().numpy().view(np.int64)[16 * 12 * 2:16 * 12 * 2 + 3]
d1 = out[1000].cpu().numpy().view(np.int64)[:3] if n > 1000 else None
for name, v in (("block 0 (first round)", d0), ("block 1000 (last round)", d1)):
    if v is not None:
        print("%s: prologue %d clocks, %d passes %d clocks, whole kernel %d clocks -> prologue+epilogue = %.1f %%" % (
            name, v[0], int(it[0]), v[1], v[2], 100.0 * (v[2] - v[1]) / v[2]))
for ph, name in ((0, "CN"), (1, "BN")):
    if ends[ph]:
        print(name, "phase: waves finish between", min(ends[ph]), "and", max(ends[ph]), "clocks after the pass started")
