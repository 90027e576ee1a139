#!/usr/bin/env python3
"""Fixed-work decoder throughput per code (random LLRs: the parity check never passes, all numMaxIter+1 passes run).

  python tools/sweep_codes.py [n_blocks] > gpurun_out/sweep_codes.txt
"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import openairinterface5g_amd as pkg  # noqa: E402

m = pkg.ldpc
pkg.LDPCinit()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
KERNEL = int(os.environ.get("SWEEP_KERNEL", "0"))   # nrLDPC_hip_dec_batch_t.kernel: 0 best, 3 one block per workgroup, 5 several
only = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]  # optional: BG,Z,R triples
codes = [(1, 384, 13), (1, 384, 23), (1, 384, 89), (1, 352, 13), (1, 320, 13), (1, 256, 13), (1, 192, 13), (1, 128, 13), (1, 96, 13),
         (1, 64, 13), (1, 32, 13), (1, 16, 13), (1, 8, 13), (1, 30, 13), (1, 7, 13),
         (2, 384, 15), (2, 208, 15), (2, 208, 13), (2, 208, 23), (2, 128, 15), (2, 64, 15), (2, 64, 13), (2, 32, 15), (2, 16, 15), (2, 15, 15)]
print("BG   Z   R  threads  lds_KiB  kernel     ms   blocks/us  coded_Gb/s  ns/(edge*Z*pass)")
for BG, Z, R in (only or codes):
    g = torch.Generator(device="cuda").manual_seed(Z + R)
    stride = (m.NCOLS[(BG, R)] * Z + 15) // 16 * 16
    llr = torch.randint(-128, 128, (n, stride), dtype=torch.int8, device="cuda", generator=g)
    llr[:, :2 * Z] = 0
    out = torch.zeros((n, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
    it = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(3):
        pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8, kernel=KERNEL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8, kernel=KERNEL)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert int(it.min()) == 9, (BG, Z, R, int(it.min()))
    ntx = (66 if BG == 1 else 50) * Z
    info = m.code_info(BG, Z, R) if hasattr(m, "code_info") else {}
    edges = info.get("nedges", 0)
    print("%2d %4d  %2d  %7s  %7s  %-7s %7.3f  %9.2f  %10.2f  %s" % (
        BG, Z, R, info.get("threads", "?"), info.get("lds_kib", "?"), info.get("kernel", "?"), dt * 1e3, n / dt / 1e6,
        n * ntx / dt / 1e9, ("%.4f" % (dt * 1e9 / (n * edges * Z * 9)) if edges else "?")))
