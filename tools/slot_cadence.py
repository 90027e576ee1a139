#!/usr/bin/env python3
"""The 64-TB slot (BASELINE configs[3]/[4]) at the cadence a gNB runs it: one UL-SCH decode + one DL-SCH encode per slot period,
the GPU idle in between -- against the same calls back to back.  After an idle period the GPU's clocks ramp for ~30 ms
(tools/clock_ramp.py); this tool shows what that costs a slot that arrives every 0.5 ms.

  python tools/slot_cadence.py [slots per point]

Per point: median / p90 of (a) the host's call -> results-ready time of the UL-SCH decode, (b) the GPU time between a pair of
events around it, (c) the same for the DL-SCH encode."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 400
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
llr = ((1.0 - 2.0 * coded.float()) * 10 + 1.8 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
pay_out = torch.zeros_like(payload)
ack = torch.zeros(64, dtype=torch.uint8, device="cuda")
itm = torch.zeros(64, dtype=torch.int32, device="cuda")
dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm)
for _ in range(5):
    dec.decode()
    enc.encode()
torch.cuda.synchronize()


def point(period_s, label):
    host, gpu_ul, gpu_dl = [], [], []
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_slots)]
    t_next = time.perf_counter()
    for i in range(n_slots):
        if period_s:
            while time.perf_counter() < t_next:      # (busy wait: the RT thread of a modem does not sleep either)
                pass
            t_next += period_s
        t0 = time.perf_counter()
        ev[i][0].record()
        dec.decode()
        ev[i][1].record()
        ev[i][1].synchronize()
        host.append(time.perf_counter() - t0)
        enc.encode()
        ev[i][2].record()
        if not period_s:
            continue
        ev[i][2].synchronize()
    torch.cuda.synchronize()
    for i in range(n_slots):
        gpu_ul.append(ev[i][0].elapsed_time(ev[i][1]))
        gpu_dl.append(ev[i][1].elapsed_time(ev[i][2]))
    h, u, d = np.array(host) * 1e3, np.array(gpu_ul), np.array(gpu_dl)
    k = n_slots // 4                                  # (the first quarter is the transition from the previous point)
    print("%-22s UL-SCH host call -> ready %.4f / %.4f ms | GPU events %.4f / %.4f ms | DL-SCH GPU events %.4f / %.4f ms   (median / p90 of %d slots; all ACK: %s)"
          % (label, np.median(h[k:]), np.percentile(h[k:], 90), np.median(u[k:]), np.percentile(u[k:], 90), np.median(d[k:]),
             np.percentile(d[k:], 90), n_slots - k, bool(ack.all().item())))


point(0.0, "back to back")
for per_ms in (0.25, 0.5, 1.0, 10.0):
    point(per_ms * 1e-3, "one slot per %.2f ms" % per_ms)
point(0.0, "back to back again")
