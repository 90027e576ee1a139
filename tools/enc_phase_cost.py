#!/usr/bin/env python3
"""Where the encoder kernel's time goes under its real occupancy: a -DENC_PHASE_STOP build returns after phase k (k in the
second byte of Kb); the kernel's duration for k = 1 .. 4 and for the whole kernel, 1024 x BG1 Zc=384, HIP events.
  hipcc ... -DENC_PHASE_STOP -c ldpc_encoder.hip ...; NRLDPC_HIP_LIB=<that library> python tools/enc_phase_cost.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import openairinterface5g_amd as pkg
pkg.LDPCinit()
BG, Z, n = 1, 384, 1024
info = torch.randint(0, 256, (n, 22 * Z // 8), dtype=torch.uint8, device="cuda")
out = torch.empty((n, 66 * Z), dtype=torch.uint8, device="cuda")
names = {9: "launch only (kernel returns at once)", 1: "loads -> LDS (info columns, tables)", 2: "+ lambda", 3: "+ core parity (one wave)", 4: "+ extension rows", 0: "+ byte-per-bit stores (whole kernel)"}
prev = 0.0
for k in (9, 1, 2, 3, 4, 0):
    os.environ["NRLDPC_HIP_ENC_STOP"] = str(k)
    Kb = 22
    for _ in range(5):
        pkg.encode_batch_device(BG, Z, info, out, Kb)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg.encode_batch_device(BG, Z, info, out, Kb)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[len(ts) // 4]
    print("%-42s %6.1f us  (+%5.1f)" % (names[k], t, t - prev))
    prev = t
