#!/usr/bin/env python3
"""Kernel time against the number of passes (fixed work: noise that never converges, iteration cap 0 .. 8): the slope is the
cost of a pass, the intercept what a block costs besides its passes (prologue, hard decisions, output; in CRC mode the
checks).  python tools/pass_slope.py [BG Z R blocks]"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import openairinterface5g_amd as pkg
m = pkg.ldpc
pkg.LDPCinit()
BG, Z, R, n = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1, 384, 89, 1664)))
g = torch.Generator(device="cuda").manual_seed(1)
llr = torch.randint(-100, 100, (n, m.NCOLS[(BG, R)] * Z), dtype=torch.int8, device="cuda", generator=g)
llr[:, :2 * Z] = 0
out = torch.zeros((n, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
it = torch.zeros(n, dtype=torch.int32, device="cuda")
K = (22 if BG == 1 else 10) * Z
for crc in (False, True):
    res = []
    for cap in (0, 1, 2, 4, 8):
        kw = dict(numMaxIter=cap, check_crc=crc, E=K if crc else 0)
        for _ in range(5):
            pkg.decode_batch_device(BG, Z, R, llr, out, it, **kw)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(20):
            pkg.decode_batch_device(BG, Z, R, llr, out, it, **kw)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))[10]
        res.append((cap + 1, ms))
    (p0, t0), (p1, t1) = res[0], res[-1]
    slope = (t1 - t0) / (p1 - p0)
    print("BG%d Zc=%d R=%d, %d blocks, %s stop: " % (BG, Z, R, n, "CRC" if crc else "parity-check") +
          "  ".join("%d passes %.1f us" % (p, t * 1e3) for p, t in res) + "  -> %.1f us per pass, %.1f us besides" % (slope * 1e3, (t0 - slope * p0) * 1e3))
