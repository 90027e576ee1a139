#!/usr/bin/env python3
"""UL-SCH / DL-SCH chain time against the number of transport blocks per call (273 PRB x 13 symbols, 64QAM TBs of 26
segments, device buffers, descriptors repeated from call to call): what one slot costs when it carries 1, 4, 16 or 64 TBs.

  python tools/tb_latency.py
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

m = pkg.ldpc
pkg.LDPCinit()
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6
for n_tb in (1, 2, 4, 8, 16, 32, 64):
    tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(n_tb)]
    po, co, ho, segs = m.tb_layout(tbs)
    payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
    coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
    enc = m.PreparedTbBatch(tbs, payload, coded)
    enc.encode()
    llr = ((1.0 - 2.0 * coded.float()) * 10 + 1.8 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
    harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
    pay_out = torch.zeros_like(payload)
    ack = torch.zeros(n_tb, dtype=torch.uint8, device="cuda")
    itm = torch.zeros(n_tb, dtype=torch.int32, device="cuda")
    dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm)
    res = {}
    for name, fn in (("ulsch", dec.decode), ("dlsch", enc.encode)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(40):                      # one call at a time, completion awaited: latency, not pipelined rate
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        res[name] = ts[len(ts) // 2] * 1e6
    print(f"{n_tb:3d} TB(s) = {sum(segs):5d} segments: UL-SCH call + completion {res['ulsch']:7.1f} us, DL-SCH {res['dlsch']:6.1f} us"
          f"   all ACK: {bool(ack.all().item())}")
