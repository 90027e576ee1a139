#!/usr/bin/env python3
"""PCIe-inclusive cost of one slot's UL-SCH chain when the LLRs arrive in HOST memory (BASELINE configs[4]: 64 transport
blocks of 26 segments, 273 PRB x 13 symbols, 64QAM = 31.4 MB of int16 LLRs per slot), by where the HARQ soft buffers live:

  host      legacy layout: everything in host arrays (soft buffers cross the link both ways on retransmissions, one way on
            first transmissions)
  device    NRLDPC_HIP_MEM_HOST | NRLDPC_HIP_MEM_HARQ_DEVICE: soft buffers in the caller's device memory
  library   NRLDPC_HIP_MEM_HOST | NRLDPC_HIP_MEM_HARQ_LIBRARY: soft buffers kept by the library
  resident  everything in device memory (the HBM-resident figure bench.py reports), for reference

each with pageable host arrays and with page-locked ones (LLRs: copied in chunks that overlap with the decoding, or read in
place by the segments' workgroups for small calls; payload and verdicts: written by the kernels in place),
for a first transmission (round 0) and a retransmission (round 1, rv 2), plus the latency of a 1-TB call.  Prints one
JSON object; the link's own rate is measured with a plain pinned hipMemcpy of the same size beside it.

  python tools/slot_chain_host.py [reps] [n_tb]
"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_tb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = pkg.ldpc
pkg.LDPCinit()
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6


def make(n):
    tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(n)]
    po, co, ho, segs = m.tb_layout(tbs)
    payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
    coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
    m.PreparedTbBatch(tbs, payload, coded).encode()
    llr0 = ((1.0 - 2.0 * coded.float()) * 10 + 1.8 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
    tbs1 = [dict(t, rv=2, round=1) for t in tbs]
    coded1 = torch.zeros_like(coded)
    m.PreparedTbBatch(tbs1, payload, coded1).encode()
    llr1 = ((1.0 - 2.0 * coded1.float()) * 10 + 1.8 * torch.randn(coded1.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
    torch.cuda.synchronize()
    return tbs, tbs1, (po, co, ho, segs), payload, llr0, llr1


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def run(n, out):
    tbs, tbs1, (po, co, ho, segs), payload, llr0_d, llr1_d = make(n)
    llr_bytes = int(co[n]) * 2
    harq_bytes = int(ho[n]) * 2
    out["segments"] = int(sum(segs))
    out["llr_MB"] = llr_bytes / 1e6
    out["harq_MB"] = harq_bytes / 1e6
    # the link: one pinned copy of the LLR bytes
    pin = m.PinnedArray(int(co[n]) + 16, np.int16)
    dst = torch.empty(int(co[n]) + 16, dtype=torch.int16, device="cuda")
    pin_t = torch.from_numpy(pin.a)
    out["link_h2d_ms"] = timed(lambda: dst.copy_(pin_t, non_blocking=True), max(reps, 5))
    out["link_GBps"] = llr_bytes / out["link_h2d_ms"] / 1e6
    # first round to get llrLen for round 1
    harq_d = torch.zeros(int(ho[n]) + 16, dtype=torch.int16, device="cuda")
    pay_d = torch.zeros_like(payload)
    ack_d = torch.zeros(n, dtype=torch.uint8, device="cuda")
    itm_d = torch.zeros(n, dtype=torch.int32, device="cuda")
    d0 = m.PreparedTbBatch(tbs, pay_d, llr0_d, harq_d, ack_d, itm_d)
    d0.decode()
    torch.cuda.synchronize()
    for i, t in enumerate(tbs1):
        t["llrLen"] = int(d0.arr[i].llrLen)
    d1 = m.PreparedTbBatch(tbs1, pay_d, llr1_d, harq_d, ack_d, itm_d)
    out["resident"] = {"round0_ms": timed(d0.decode, reps), "round1_ms": timed(d1.decode, reps), "all_ack": bool(ack_d.all().item())}
    ids = list(range(0x100, 0x100 + n))
    for pinned in (False, True):
        # pageable run: every host array pageable; page-locked run: LLRs, payload and verdict arrays page-locked
        if pinned:
            keep_out = (m.PinnedArray(int(po[n]) + 16, np.uint8), m.PinnedArray(max(n, 1), np.uint8), m.PinnedArray(max(n, 1), np.int32))
            pay_h, ack_h, itm_h = (k.a for k in keep_out)
        else:
            pay_h, ack_h, itm_h = np.zeros(int(po[n]) + 16, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.int32)
        srcs = []
        for llr_d in (llr0_d, llr1_d):
            if pinned:
                k = m.PinnedArray(int(co[n]) + 16, np.int16)
                k.a[:] = llr_d.cpu().numpy()
                srcs.append(k)
            else:
                srcs.append(llr_d.cpu().numpy().copy())
        for where in ("host", "host_registered", "device", "library"):
            if where == "host":
                harq, mem, kw = np.zeros(int(ho[n]) + 16, np.int16), m.MEM_HOST, {}
            elif where == "host_registered":   # the caller page-locks its persistent d[r] once (nrLDPC_hip_host_register / _alloc)
                keep_harq = m.PinnedArray(int(ho[n]) + 16, np.int16)
                keep_harq.a[:] = 0
                harq, mem, kw = keep_harq.a, m.MEM_HOST, {}
            elif where == "device":
                harq, mem, kw = torch.zeros(int(ho[n]) + 16, dtype=torch.int16, device="cuda"), m.MEM_HOST | m.MEM_HARQ_DEVICE, {}
            else:
                harq, mem, kw = None, m.MEM_HOST | m.MEM_HARQ_LIBRARY, {"harq_ids": ids}
            b0 = m.PreparedTbBatch(tbs, pay_h, srcs[0], harq, ack_h, itm_h, mem=mem, **kw)
            b1 = m.PreparedTbBatch(tbs1, pay_h, srcs[1], harq, ack_h, itm_h, mem=mem, **kw)
            r = {"round0_ms": timed(b0.decode, reps)}
            ok0 = bool(ack_h.all())
            r["round1_ms"] = timed(b1.decode, reps)
            r["all_ack"] = ok0 and bool(ack_h.all())
            moved0 = llr_bytes + (harq_bytes if where.startswith("host") else 0)        # round 0: soft buffers come back only
            moved1 = llr_bytes + (2 * harq_bytes if where.startswith("host") else 0)
            r["link_MB_round0"], r["link_MB_round1"] = moved0 / 1e6, moved1 / 1e6
            r["llr_rate_GBps_round0"] = llr_bytes / r["round0_ms"] / 1e6
            r["frac_of_link_round0"] = (moved0 / r["round0_ms"] / 1e6) / out["link_GBps"]
            r["frac_of_link_round1"] = (moved1 / r["round1_ms"] / 1e6) / out["link_GBps"]
            out[f"{'pinned' if pinned else 'pageable'}_llr/{where}_harq"] = r
            if where == "library":
                m.harq_release()
    return out


res = {"config": f"{n_tb} TB x A={A} G={G} Qm=6 (BASELINE configs[4]); median of {reps} calls, completion awaited"}
run(n_tb, res)
one = {}
run(1, one)
res["one_tb"] = {k: v for k, v in one.items() if isinstance(v, dict) or k in ("llr_MB", "link_h2d_ms")}
print(json.dumps(res, indent=1))
