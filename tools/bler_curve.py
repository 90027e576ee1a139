#!/usr/bin/env python3
"""BLER and mean pass count vs Es/N0 of the HIP decoder (numMaxIter = 8, parity-check stop), 4096 blocks per point, with
every block of every point also decoded by the oracle's vectorisable restatement on the host: the two must agree in
every bit and pass count, so the curve is the reference algorithm's curve.

  python tools/bler_curve.py > gpurun_out/bler_curve.txt
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

m = pkg.ldpc
pkg.LDPCinit()
n = 4096
threads = min(os.cpu_count() or 1, 64)
for BG, Z, R, grid in ((1, 384, 13, np.arange(-3.0, 1.51, 0.5)), (2, 64, 15, np.arange(-5.0, 1.01, 1.0))):
    K = (22 if BG == 1 else 10) * Z
    N = (66 if BG == 1 else 50) * Z
    ncols = m.NCOLS[(BG, R)]
    print(f"BG{BG} Zc={Z} R-mode {R} (K={K}, N={N}), numMaxIter=8, {n} blocks per point")
    print("  Es/N0 dB      BLER   mean passes   identical to the oracle (bits, passes)")
    for snr in grid:
        g = torch.Generator(device="cuda").manual_seed(int(1000 + 10 * snr) + Z)
        info = torch.randint(0, 256, (n, K // 8), dtype=torch.uint8, device="cuda", generator=g)
        coded = torch.empty((n, N), dtype=torch.uint8, device="cuda")
        pkg.encode_batch_device(BG, Z, info, coded)
        sigma = 1.0 / np.sqrt(2.0 * 10.0 ** (snr / 10.0))
        y = 1.0 - 2.0 * coded.float() + sigma * torch.randn((n, N), device="cuda", generator=g)
        llr = torch.zeros((n, ncols * Z), dtype=torch.int8, device="cuda")
        llr[:, 2 * Z:] = torch.clamp(torch.floor(y / (sigma / 16.0)), -128, 127).to(torch.int8)
        out = torch.zeros((n, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
        it = torch.zeros(n, dtype=torch.int32, device="cuda")
        pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8)
        torch.cuda.synchronize()
        it_h, out_h = it.cpu().numpy(), out.cpu().numpy()
        bad = (it_h > 8) | (out[:, :K // 8] != info).any(dim=1).cpu().numpy()
        it_o, out_o = O.decode_mt(threads, BG, Z, R, llr.cpu().numpy(), 8, vec=True)
        same = np.array_equal(it_o, it_h) and np.array_equal(out_o, out_h)
        print("  %8.1f  %8.5f   %11.3f   %s" % (snr, bad.mean(), it_h.mean(), same))
        assert same
