#!/usr/bin/env python3
"""PCIe-inclusive decode of BASELINE configs[1] (1024 x BG1 Zc=384 R=1/3 at 1 dB) from page-locked and from pageable host
memory through LDPCdecoder_batch; run once per NRLDPC_HIP_HOST_CHUNK setting (the library reads it at first use)."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import openairinterface5g_amd as pkg
pkg.LDPCinit()
BG, Z, R, n = 1, 384, 13, 1024
g = torch.Generator(device="cuda").manual_seed(5)
info = torch.randint(0, 256, (n, 22 * Z // 8), dtype=torch.uint8, device="cuda", generator=g)
coded = torch.empty((n, 66 * Z), dtype=torch.uint8, device="cuda")
pkg.encode_batch_device(BG, Z, info, coded)
sigma = 1.0 / np.sqrt(2.0 * 10.0 ** 0.1)
y = 1.0 - 2.0 * coded.float() + sigma * torch.randn((n, 66 * Z), device="cuda", generator=g)
llr = torch.zeros((n, 68 * Z), dtype=torch.int8, device="cuda")
llr[:, 2 * Z:] = torch.clamp(torch.floor(y / (sigma / 16.0)), -128, 127).to(torch.int8)
pin = torch.empty(llr.shape, dtype=torch.int8, pin_memory=True); pin.copy_(llr); pinn = pin.numpy()
page = llr.cpu().numpy().copy()
out = np.zeros((n, 3264), np.uint8)
out_pin = torch.zeros((n, 3264), dtype=torch.uint8, pin_memory=True).numpy()
ref_it, ref_out = pkg.decode_batch_host(BG, Z, R, page, numMaxIter=8)
for name, src, dst in (("pinned LLRs, pinned out", pinn, out_pin), ("pinned", pinn, out), ("pageable", page, out)):
    for _ in range(3):
        it, o = pkg.decode_batch_host(BG, Z, R, src, numMaxIter=8, out=dst)
    assert np.array_equal(it, ref_it) and np.array_equal(o, ref_out), name
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); pkg.decode_batch_host(BG, Z, R, src, numMaxIter=8, out=dst); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"chunk={os.environ.get('NRLDPC_HIP_HOST_CHUNK','default')} {name}: median {ts[7]*1e3:.3f} ms min {ts[0]*1e3:.3f} ms -> {n*66*Z/ts[7]/1e9:.1f} Gb/s coded")
