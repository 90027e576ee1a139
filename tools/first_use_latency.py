#!/usr/bin/env python3
"""First LDPCdecoder() call of a code never used before while the resident server is already running (descriptor build +
upload happen under it): must be a fraction of a millisecond, not the server's idle time-out.

  python tools/first_use_latency.py
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import openairinterface5g_amd as pkg
pkg.LDPCinit()
m = pkg.ldpc
rng = np.random.default_rng(1)
def call(BG, Z, R):
    p = pkg.make_dec_params(BG, Z, R, 8)
    llr = rng.integers(-20, 20, m.num_llr(BG, Z, R)).astype(np.int8)
    t0 = time.perf_counter(); n, _ = pkg.LDPCdecoder(p, llr); t1 = time.perf_counter()
    t2 = time.perf_counter(); n, _ = pkg.LDPCdecoder(p, llr); t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t3 - t2) * 1e6
print("first code (starts the server): first call %.0f us, second %.0f us" % call(1, 384, 13))
for code in ((1, 352, 13), (2, 208, 15), (1, 96, 23), (2, 30, 13), (1, 7, 89)):
    print(code, "first call %.0f us, second %.0f us" % call(*code))
