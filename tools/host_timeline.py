#!/usr/bin/env python3
"""One host-LLR UL-SCH slot (configs[4], page-locked arrays, library-kept soft buffers) a few times: the workload
rocprofv3 --kernel-trace --memory-copy-trace is pointed at to see how the copies and the kernels of a call fall in time.
  python tools/host_timeline.py [calls]      (then tools/host_timeline_parse.py <trace dir>)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402
m = pkg.ldpc
pkg.LDPCinit()
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6
n = 64
tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(n)]
po, co, ho, segs = m.tb_layout(tbs)
payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
m.PreparedTbBatch(tbs, payload, coded).encode()
llr = ((1.0 - 2.0 * coded.float()) * 10 + 1.8 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
src = m.PinnedArray(int(co[n]) + 16, np.int16)
src.a[:] = llr.cpu().numpy()
pay, ack, itm = m.PinnedArray(int(po[n]) + 16, np.uint8), m.PinnedArray(n, np.uint8), m.PinnedArray(n, np.int32)
b = m.PreparedTbBatch(tbs, pay, src, None, ack, itm, mem=m.MEM_HOST | m.MEM_HARQ_LIBRARY, harq_ids=list(range(n)))
import time
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    t0 = time.perf_counter()
    b.decode()
    print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms, all ack {bool(ack.a.all())}")
