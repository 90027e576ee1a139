#!/usr/bin/env python3
"""Randomised soak of the decoder through the C ABI against the oracle's vectorisable restatement: random code, batch size,
iteration cap, stop mode, output mode and input statistics, every block of every batch compared (bits and pass counts).

  python tools/soak.py [seconds] [seed]
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402
from common import ALL_RATES, kbits, make_llr, random_info  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

pkg.LDPCinit()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
threads = min(os.cpu_count() or 1, 64)
t0 = time.time()
batches = blocks = 0
while time.time() - t0 < budget:
    BG = int(rng.integers(1, 3))
    Z = int(os.environ["SOAK_Z"]) if os.environ.get("SOAK_Z") else int(rng.choice(O.LIFT_SIZES))   # (SOAK_Z=384: the lifting size with kernels of its own)
    R = int(rng.choice(ALL_RATES[BG]))
    n = int(rng.choice([1, 2, 3, 17, 64, 255, 256, 257, 300, 513]))
    it = int(rng.choice([0, 1, 2, 3, 5, 8, 12]))
    K = kbits(BG, Z)
    use_crc = bool(rng.integers(0, 2)) and K % 8 == 0 and K >= 48
    mode = 0 if use_crc else int(rng.integers(0, 3))
    if Z % 4 == 0 and Z >= 8:      # 5: several blocks per workgroup (Zc <= 64) / four blocks interleaved (other small Zc)
        kern = int(rng.choice([0, 0, 1, 3, 4, 5] if Z <= 64 else [0, 0, 1, 3, 4]))
    else:
        kern = int(rng.choice([0, 1, 5] if Z <= 30 else [0, 1]))
    base = [make_llr(rng, BG, Z, R, k, random_info(rng, BG, Z, with_crc24b=True)) for k in (-2.0, 0.0, 1.5, "rand", "sat")]
    llr = np.stack([base[int(rng.integers(0, len(base)))] for _ in range(n)])
    pre = np.full((n, (pkg.ldpc.out_bytes(BG, Z, R, mode) + 3) // 4 * 4), 0x33, np.uint8)
    n_gpu, out_gpu = pkg.decode_batch_host(BG, Z, R, llr, numMaxIter=it, outMode=mode, check_crc=use_crc, E=K, crc_type=1,
                                           out=pre.copy(), kernel=kern)
    if use_crc or mode != 0:
        for i in range(0, n, max(1, n // 24)):        # the threaded oracle driver does parity-check mode / packed bits only
            nr, outr = O.decode(BG, Z, R, llr[i], it, mode, use_crc, K, 1, out_init=0x33, vec=True)
            assert nr == n_gpu[i] and np.array_equal(outr, out_gpu[i]), (BG, Z, R, n, it, mode, use_crc, kern, i)
            blocks += 1
    else:
        n_ref, out_ref = O.decode_mt(threads, BG, Z, R, llr, it, vec=True)
        assert np.array_equal(n_ref, n_gpu) and np.array_equal(out_ref, out_gpu), (BG, Z, R, n, it, kern)
        blocks += n
    batches += 1
print("soak: %d batches, %d blocks compared, 0 mismatches in %.0f s" % (batches, blocks, time.time() - t0))
