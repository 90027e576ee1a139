#!/usr/bin/env python3
"""One slot's 64 transport blocks (273 PRB x 13 symbols, 64QAM: 1664 code segments; BASELINE configs[3]/[4]) through the
DL-SCH and UL-SCH chains on device buffers, N times -- the workload rocprofv3 is pointed at for the chain kernels.

  python tools/slot_chain.py [reps] [sigma] [retx]

retx: every first transmission (rv 0, round 0) is followed by a retransmission (rv 2, round 1) that is combined into the
same soft buffers -- the UL-SCH figure is then the mean of the two calls.
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.18
m = pkg.ldpc
pkg.LDPCinit()
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6
tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(64)]
po, co, ho, segs = m.tb_layout(tbs)
payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
enc = m.PreparedTbBatch(tbs, payload, coded)
enc.encode()
llr = ((1.0 - 2.0 * coded.float()) * 10 + sigma * 10 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
pay_out = torch.zeros_like(payload)
ack = torch.zeros(64, dtype=torch.uint8, device="cuda")
itm = torch.zeros(64, dtype=torch.int32, device="cuda")
dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm)
retx = len(sys.argv) > 3 and sys.argv[3] == "retx"
dec1 = None
if retx:
    dec.decode()
    torch.cuda.synchronize()
    tbs1 = [dict(t, rv=2, round=1, llrLen=int(dec.arr[i].llrLen)) for i, t in enumerate(tbs)]
    coded1 = torch.zeros_like(coded)
    m.PreparedTbBatch(tbs1, payload, coded1).encode()
    llr1 = ((1.0 - 2.0 * coded1.float()) * 10 + sigma * 10 * torch.randn(coded1.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
    dec1 = m.PreparedTbBatch(tbs1, pay_out, llr1, harq, ack, itm)
for _ in range(3):
    enc.encode()
    dec.decode()
    if dec1:
        dec1.decode()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    dec.decode()
    if dec1:
        dec1.decode()
torch.cuda.synchronize()
t_dec = (time.perf_counter() - t0) / reps / (2 if dec1 else 1)
t0 = time.perf_counter()
for _ in range(reps):
    enc.encode()
torch.cuda.synchronize()
t_enc = (time.perf_counter() - t0) / reps
print({"retransmissions": retx, "segments": int(sum(segs)), "ulsch_ms": t_dec * 1e3, "dlsch_ms": t_enc * 1e3, "all_ack": bool(ack.all().item()),
       "max_passes": int(itm.max().item()), "info_gbps_rx": 64 * A / t_dec / 1e9})
