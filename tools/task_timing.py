#!/usr/bin/env python3
"""Timeline of the fast decoder's tasks (needs the -DLDPC_TIMING build, tools/task_timing.sh, as NRLDPC_HIP_LIB):
pass 2 of block 0, per wave: C<degree> = check-node task, B<loop bound>[xN] = bit-node task (N lanes per item), start and end
in shader clocks since the first task of the pass."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import openairinterface5g_amd as pkg
m = pkg.ldpc
pkg.LDPCinit()
BG, Z, R = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 384, 13)))
SLOTS, n = 28, 1024
llr = torch.randint(-128, 128, (n, m.NCOLS[(BG, R)] * Z), dtype=torch.int8, device="cuda")
llr[:, :2 * Z] = 0
out = torch.zeros((n, max(m.out_bytes(BG, Z, R), 16 * SLOTS * 16 + 64)), dtype=torch.uint8, device="cuda")
it = torch.zeros(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    out[0].zero_()
    pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8, kernel=3)
torch.cuda.synchronize()
d = out[0].cpu().numpy().view(np.int64)[:16 * SLOTS * 2].reshape(16, SLOTS, 2)
rows, first = [], None
for w in range(16):
    row = []
    for k in range(SLOTS):
        a, e = int(d[w, k, 0]), int(d[w, k, 1])
        if a == 0 and e == 0:
            continue
        start, ph, deg = a >> 20, (a >> 8) & 0xf, a & 0xff
        first = start if first is None else min(first, start)
        row.append((ph, deg, start, e))
    rows.append(row)
ends = {0: [], 1: []}
busy = {0: 0, 1: 0}
for w, row in enumerate(rows):
    if not row:
        continue
    def name(ph, deg):
        if ph == 2:
            return "w"
        if ph == 3:
            return "s%d" % deg
        return "C%d" % deg if ph == 0 else ("B%d" % (deg & 63) + ("x%d" % (deg // 64 + 1) if deg >= 64 else ""))
    print("wave %2d:" % w, "  ".join("%s[%d..%d]" % (name(ph, deg), s - first, e - first) for ph, deg, s, e in row))
    for ph in (0, 1):
        es = [e for p_, _, _, e in row if p_ == ph]
        busy[ph] += sum(e - s for p_, _, s, e in row if p_ == ph)
        if es:
            ends[ph].append(max(es) - first)
starts_bn = [s - first for row in rows for ph, _, s, _ in row if ph == 1]
for ph, nm in ((0, "CN"), (1, "BN")):
    if ends[ph]:
        print(nm, "phase: waves finish between", min(ends[ph]), "and", max(ends[ph]), "clocks after the pass started; sum of task "
              "durations over the 16 waves", busy[ph])
if starts_bn:
    print("BN phase: first task starts at", min(starts_bn), "-> phase length", max(ends[1]) - min(starts_bn))
