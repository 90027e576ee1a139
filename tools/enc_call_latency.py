#!/usr/bin/env python3
"""LDPCencoder() per call (8 segments of BG1 Zc=384), timed around the C call: resident encoder server vs one launch per call.

  python tools/enc_call_latency.py [calls]
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import openairinterface5g_amd as pkg  # noqa: E402

m = pkg.ldpc
pkg.LDPCinit()
L = m.load_library()
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for BG, Zc, nseg in ((1, 384, 8), (1, 384, 1), (2, 64, 8)):
    K = (22 if BG == 1 else 10) * Zc
    rng = np.random.default_rng(Zc)
    ins = [np.ascontiguousarray(np.concatenate([rng.integers(0, 256, K // 8, dtype=np.uint8), np.zeros(8, np.uint8)])) for _ in range(nseg)]
    outs = [np.zeros(68 * 384, np.uint8) for _ in range(nseg)]
    ip = (C.c_void_p * nseg)(*[a.ctypes.data for a in ins])
    op = (C.c_void_p * nseg)(*[a.ctypes.data for a in outs])
    impp = m.encoder_implemparams_t(n_segments=nseg, macro_num=0, gen_code=0, Kr=K, Kb=22 if BG == 1 else 10, Zc=Zc, BG=BG, K=K, E=K)
    N = (66 if BG == 1 else 50) * Zc
    # two different inputs, alternated call by call, each checked against the batch encoder (device path, parity-tested)
    alt = [[rng.integers(0, 256, K // 8, dtype=np.uint8) for _ in range(nseg)] for _ in range(2)]
    refs = [pkg.encode_batch_host(BG, Zc, np.stack(a)) for a in alt]
    ok, dt = True, 0.0
    for i in range(n_calls + 50):
        which = i & 1
        for j in range(nseg):
            ins[j][:K // 8] = alt[which][j]
        t0 = time.perf_counter()
        L.LDPCencoder(ip, op, C.byref(impp))
        if i >= 50:
            dt += time.perf_counter() - t0
        if i % 97 == 0 or i < 4:
            ok = ok and all(np.array_equal(outs[j][:N], refs[which][j][:N]) for j in range(nseg))
    print(f"BG{BG} Zc={Zc} {nseg} segment(s) per call: {dt / n_calls * 1e6:.1f} us per call, code words equal to the batch encoder's: {ok}")
