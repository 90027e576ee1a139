#!/usr/bin/env python3
"""Where the time of one LDPCdecoder() call goes (resident server path): BG1 Zc=384 R=1/3 at ldpctest's -s 10 (3 passes),
N calls from one thread; the library's own stamps (nrLDPC_hip_server_stats) split the host call into
GPU doorbell-seen -> LLRs staged in LDS, staged -> decoded + result written, and what is left on the host side.

  python tools/srv_breakdown.py [calls] [BG Z R snr_dB [crc]]      crc: CRC stop mode (the check fails on random bits: all passes run)
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
BG, Z, R = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 384, 13)
snr = float(sys.argv[5]) if len(sys.argv) > 5 else 10.0
pkg.LDPCinit()
m = pkg.ldpc
rng = np.random.default_rng(7)
K = (22 if BG == 1 else 10) * Z
info = rng.integers(0, 256, K // 8, dtype=np.uint8)
coded = O.encode(BG, Z, info)
llr = np.zeros(m.num_llr(BG, Z, R) + 64, np.int8)
l = O.awgn_llr(rng, coded, Z, snr)
llr[:min(l.size, m.num_llr(BG, Z, R))] = l[:m.num_llr(BG, Z, R)]
crc = len(sys.argv) > 6 and sys.argv[6] == "crc"
p = pkg.make_dec_params(BG, Z, R, 8, check_crc=crc, E=K if crc else 0)
out = np.zeros(m.out_bytes(BG, Z, R) + 64, np.uint8)
call, keep = m.raw_decoder_call(p)
for _ in range(200):
    it = call(llr.ctypes.data, out.ctypes.data)
import os
gap = float(os.environ.get("SRV_GAP_US", "0")) * 1e-6  # idle time between calls (ldpctest leaves ~1 ms: encoder, channel, compare)
s0 = m.server_stats()
t_calls = 0.0
for _ in range(n):
    if gap:
        time.sleep(gap)
    t0 = time.perf_counter()
    call(llr.ctypes.data, out.ctypes.data)
    t_calls += time.perf_counter() - t0
t = t_calls / n
s1 = m.server_stats()
d = {k: (s1[k] - s0[k]) for k in s1}
c = max(d["calls"], 1)
print({"gap_us": gap * 1e6, "code": (BG, Z, R), "crc_mode": crc, "passes": it, "calls": d["calls"], "us_per_call_python": round(t * 1e6, 2),
       "host_call_us": round(d["host_call_ns"] / c / 1e3, 2), "host_wait_us": round(d["host_wait_ns"] / c / 1e3, 2),
       "gpu_stage_us": round(d["gpu_stage_ns"] / c / 1e3, 2), "gpu_decode_us": round(d["gpu_decode_ns"] / c / 1e3, 2)})
