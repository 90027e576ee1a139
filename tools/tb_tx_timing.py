#!/usr/bin/env python3
"""Per-phase clocks of tb_tx_fused_kernel's workgroup 0 (needs a -DTB_TIMING build of tb_chain.hip as NRLDPC_HIP_LIB):
  hipcc ... -DTB_TIMING -c tb_chain.hip -o build/tb_chain_t.o; link as libldpc_hip_tbt.so
  NRLDPC_HIP_LIB=.../libldpc_hip_tbt.so python tools/tb_tx_timing.py [n_tb]"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import openairinterface5g_amd as pkg
m = pkg.ldpc
pkg.LDPCinit()
n_tb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6
tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(n_tb)]
po, co, ho, segs = m.tb_layout(tbs)
payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
coded = torch.zeros(int(co[-1]) + 4096, dtype=torch.uint8, device="cuda")
enc = m.PreparedTbBatch(tbs, payload, coded)
for _ in range(3):
    enc.encode()
torch.cuda.synchronize()
C = segs[0]
E_last = m.nr_get_E(G, C, 6, 1, C - 1)
off = (int(co[n_tb - 1]) + G + 15) & ~15
t = coded[off:off + 8 * 20].cpu().numpy().view(np.int64)
names = ["loads -> LDS", "CB CRC", "unpack", "extend info", "lambda", "extend lambda", "core parity (1 wave)", "extension rows",
         "select pack", "select store"]
for k in range(1, len(names) + 1):
    print("%-22s %7d clocks" % (names[k - 1], t[k] - t[k - 1]))
print("%-22s %7d clocks" % ("total", t[len(names)] - t[0]))

x = coded[off + 8 * 32:off + 8 * 32 + 8 * 4 * 14].cpu().numpy().view(np.int64).reshape(14, 4)
w0 = x[0, 2]
for k in range((sum(segs) + 127) // 128):
    print("workgroup %4d: runs %6d clocks; starts %6.2f us, ends %6.2f us after workgroup 0 started (wall clock)" % (
        128 * k, x[k, 1] - x[k, 0], (x[k, 2] - w0) / 100.0, (x[k, 3] - w0) / 100.0))
