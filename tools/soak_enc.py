#!/usr/bin/env python3
"""Randomised soak of the encoder through the C ABI against the oracle: random code (every lifting size; Zc % 32 == 0 drawn
half of the time: the path of ldpc_enc_packed32.h), Kb, batch size, row pitches and alignments (device buffers), every block
compared; every 8th batch also through the by-name LDPCencoder (resident kernel, 8-segment groups).

  python tools/soak_enc.py [seconds] [seed]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

pkg.LDPCinit()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
W32 = [z for z in O.LIFT_SIZES if z % 32 == 0]
t0 = time.time()
batches = blocks = fast = 0
while time.time() - t0 < budget:
    BG = int(rng.integers(1, 3))
    Z = int(rng.choice(W32 if rng.integers(0, 2) else O.LIFT_SIZES))
    Kb = 22 if BG == 1 else int(rng.choice([10, 10, 9, 8, 6]))
    n = int(rng.choice([1, 2, 7, 8, 9, 33, 200]))
    K, N = (22 if BG == 1 else 10) * Z, (66 if BG == 1 else 50) * Z
    bits = rng.integers(0, 2, (n, K), dtype=np.uint8)
    bits[:, Kb * Z:] = 0
    info = np.packbits(bits, axis=1)
    in_off, out_off = int(rng.choice([0, 0, 1, 2, 4])), int(rng.choice([0, 0, 3, 4, 16]))
    in_pad, out_pad = int(rng.choice([0, 4, 5])), int(rng.choice([0, 16, 7]))
    buf_in = torch.zeros((n, info.shape[1] + in_off + in_pad + 8), dtype=torch.uint8, device="cuda")
    buf_in[:, in_off:in_off + info.shape[1]] = torch.from_numpy(info).cuda()
    buf_out = torch.full((n, N + out_off + out_pad + 16), 9, dtype=torch.uint8, device="cuda")
    pkg.encode_batch_device(BG, Z, buf_in[:, in_off:], buf_out[:, out_off:], Kb)
    torch.cuda.synchronize()
    got = buf_out.cpu().numpy()
    for i in range(n):
        ref = O.encode(BG, Z, info[i], Kb)
        assert np.array_equal(got[i, out_off:out_off + N], ref), (BG, Z, Kb, n, in_off, out_off, i)
    assert (got[:, :out_off] == 9).all() and (got[:, out_off + N:] == 9).all(), (BG, Z, "wrote outside its rows")
    if batches % 8 == 0 and Kb == (22 if BG == 1 else 10):
        segs = [info[i] for i in range(min(n, 11))]
        outs = pkg.LDPCencoder(segs, BG, Z, n_segments=len(segs), macro_num=0)
        for i in range(min(len(segs), 8)):
            assert np.array_equal(outs[i], O.encode(BG, Z, segs[i])), (BG, Z, "LDPCencoder", i)
    batches += 1
    blocks += n
    fast += n if Z % 32 == 0 else 0
print("soak_enc: %d batches, %d blocks compared (%d on word-aligned codes), 0 mismatches in %.0f s" % (batches, blocks, fast, time.time() - t0))
