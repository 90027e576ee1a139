#!/usr/bin/env python3
"""The DL half of the path as two workloads for rocprofv3 / event timing (round 5):
  enc   1024 x BG1 Zc=384 through LDPCencoder_batch on device buffers (ldpc_enc_packed_kernel)
  slot  the 64-TB slot of configs[3]/[4] through nrLDPC_hip_dlsch_encode (tb_tx_crc_partial_kernel + tb_tx_fused_kernel)
  python tools/tx_enc_driver.py [reps] [enc|slot|both]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
what = sys.argv[2] if len(sys.argv) > 2 else "both"
m = pkg.ldpc
pkg.LDPCinit()


def timed(fn, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("%-28s median %7.1f us   min %7.1f us  (event pair around one call, %d calls)" % (name, ts[len(ts) // 2], ts[0], reps))


if what in ("enc", "both"):
    BG, Z, n = 1, 384, 1024
    info = torch.randint(0, 256, (n, 22 * Z // 8), dtype=torch.uint8, device="cuda")
    out = torch.empty((n, 66 * Z), dtype=torch.uint8, device="cuda")
    timed(lambda: pkg.encode_batch_device(BG, Z, info, out), "encoder 1024 x BG1 Zc=384")
if what in ("slot", "both"):
    A = 213176
    while m.nr_segmentation(A + 24, 1) is None:
        A += 8
    G = (12 * 13 - 6) * 273 * 6
    tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(64)]
    po, co, ho, segs = m.tb_layout(tbs)
    payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
    coded = torch.zeros(int(co[-1]) + 4096, dtype=torch.uint8, device="cuda")
    enc = m.PreparedTbBatch(tbs, payload, coded)
    timed(enc.encode, "DL-SCH slot, 64 TBs")
