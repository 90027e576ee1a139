#!/usr/bin/env python3
"""All 93 small transport block sizes of 38.214 Table 5.1.3.2-1 (one code segment each), `copies` of each, through
nrLDPC_hip_ulsch_decode on device buffers, in four processes: the default plan (decoder launches by workgroup shape; small
segments share workgroups only when there are more of them than fit on the GPU side by side), sharing forced
(NRLDPC_HIP_TB_MULTI=2), sharing off (=0), and one decoder launch for the whole batch (NRLDPC_HIP_TB_CLASSES=0: every
workgroup with the threads and LDS of the largest segment = the round-2 plan).  ACKs, pass counts, payloads and soft
buffers must agree.

  python tools/small_tbs.py [copies] [reps] [sigma]
"""
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
TBS = [24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192, 208, 224, 240, 256,
       272, 288, 304, 320, 336, 352, 368, 384, 408, 432, 456, 480, 504, 528, 552, 576, 608, 640, 672, 704, 736, 768, 808, 848,
       888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672, 1736, 1800,
       1864, 1928, 2024, 2088, 2152, 2216, 2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104, 3240, 3368, 3496,
       3624, 3752, 3824]


def worker(copies, reps, sigma):
    sys.path.insert(0, str(ROOT))
    import torch
    import openairinterface5g_amd as pkg
    m = pkg.ldpc
    pkg.LDPCinit()
    torch.manual_seed(93)
    tbs = []
    for A in TBS:
        for k in range(copies):
            rate = (0.33, 0.5, 0.6, 0.75)[k % 4]
            BG = 2 if (A <= 292 or (A <= 3824 and rate <= 0.67)) else 1
            tbs.append(dict(A=A, G=max(int((A + 24) / rate) // 4, 4) * 4, BG=BG, Qm=4, Nl=1, rv=0, tbslbrm=0, round=0))
    po, co, ho, segs = m.tb_layout(tbs)
    payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
    coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
    enc = m.PreparedTbBatch(tbs, payload, coded)
    enc.encode()
    llr = ((1.0 - 2.0 * coded.float()) * 10 + sigma * 10 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
    harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
    pay_out = torch.zeros_like(payload)
    ack = torch.zeros(len(tbs), dtype=torch.uint8, device="cuda")
    itm = torch.zeros(len(tbs), dtype=torch.int32, device="cuda")
    dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm)
    for _ in range(3):
        dec.decode()
        enc.encode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode()
    torch.cuda.synchronize()
    t_dec = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode()
    torch.cuda.synchronize()
    t_enc = (time.perf_counter() - t0) / reps
    h = hashlib.sha256()
    for t in (ack, itm, pay_out, harq):
        h.update(t.cpu().numpy().tobytes())
    print(json.dumps({"multi": os.environ.get("NRLDPC_HIP_TB_MULTI", "1"), "classes": os.environ.get("NRLDPC_HIP_TB_CLASSES", "1"), "tbs": len(tbs), "segments": int(sum(segs)),
                      "ulsch_ms": round(t_dec * 1e3, 4), "dlsch_ms": round(t_enc * 1e3, 4), "acks": int(ack.sum().item()),
                      "mean_passes": round(float(itm.float().mean().item()), 3), "digest": h.hexdigest()[:16]}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
        sys.exit(0)
    copies = sys.argv[1] if len(sys.argv) > 1 else "64"
    reps = sys.argv[2] if len(sys.argv) > 2 else "20"
    sigma = sys.argv[3] if len(sys.argv) > 3 else "0.45"
    res = []
    for multi, classes in (("1", "1"), ("2", "1"), ("0", "1"), ("0", "0")):
        env = dict(os.environ, NRLDPC_HIP_TB_MULTI=multi, NRLDPC_HIP_TB_CLASSES=classes)
        out = subprocess.run([sys.executable, __file__, "--worker", copies, reps, sigma], env=env, capture_output=True, text=True)
        if out.returncode:
            print(out.stdout, out.stderr)
            sys.exit(1)
        res.append(json.loads(out.stdout.strip().splitlines()[-1]))
        print(res[-1])
    same = all(r["digest"] == res[0]["digest"] for r in res)
    print({"identical_results": same, "speedup_default": round(res[3]["ulsch_ms"] / res[0]["ulsch_ms"], 3),
           "speedup_classes_only": round(res[3]["ulsch_ms"] / res[2]["ulsch_ms"], 3),
           "speedup_classes_and_forced_sharing": round(res[3]["ulsch_ms"] / res[1]["ulsch_ms"], 3)})
    sys.exit(0 if same else 2)
