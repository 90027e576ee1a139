#!/usr/bin/env python3
"""Per-workgroup timeline of the homogeneous fast decoder launch (NRLDPC_HIP_DEC_TRACE, diagnostic instantiation
ldpc_dec_fast_kernel<false, false, true>): BASELINE configs[1] (1024 x BG1 Zc=384 R=1/3, numMaxIter 8) at the operating
point (Es/N0 = 1 dB, early stop) and in the fixed-work regime.  Prints, per regime: where a workgroup's time goes
(prologue, every pass' check-node / bit-node phase, parity sweeps, epilogue), the gaps between consecutive workgroups of a
CU, per-CU busy time, the tail, and what list scheduling of the measured durations on the CUs would give.
  python tools/dec_trace.py [out_prefix]"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
TRACE = tempfile.mktemp(suffix=".dectrace")
os.environ["NRLDPC_HIP_DEC_TRACE"] = TRACE


def analyse(d, label, ms_untraced):
    n = len(d)
    hw, xcc = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64)
    t0 = d[:, 2].astype(np.int64).min()
    us = lambda x: (x.astype(np.int64) - t0) / 100.0
    start, end, n_iter = us(d[:, 2]), us(d[:, 3]), d[:, 4].astype(np.int64)
    w = d[:, 5:].copy().view(np.uint32).reshape(n, -1)            # the stamps row (low 32 bits of the 100 MHz clock)
    lo = lambda x: ((x.astype(np.int64) - (t0 & 0xffffffff)) & 0xffffffff) / 100.0
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    ids = {c: i for i, c in enumerate(sorted(set(cu.tolist())))}
    cui = np.array([ids[c] for c in cu.tolist()])
    ncu = len(ids)
    dur = end - start
    span = end.max()
    print(f"== {label}: {n} workgroups on {ncu} CUs ({len(set(xcc.tolist()))} XCDs); kernel span {span:.1f} us by the trace"
          + (f", {ms_untraced * 1e3:.1f} us per launch un-traced (HIP events)" if ms_untraced else "") + "; passes (index = count) " + str(np.bincount(n_iter).tolist()))
    pro = lo(w[:, 0]) - start
    epi = end - lo(w[:, 1])
    print(f"   workgroup: total {dur.mean():.2f} us (min {dur.min():.2f}, max {dur.max():.2f}); prologue {pro.mean():.2f}; "
          f"hard decision + output (after the last pass) {epi.mean():.2f}")
    for k in sorted(set(n_iter.tolist())):
        sel = n_iter == k
        row = []
        prev = lo(w[sel, 0])
        for p in range(1, min(k + 1, 9) + 1):
            a, b = lo(w[sel, 2 * p]), lo(w[sel, 2 * p + 1])
            have_a, have_b = w[sel, 2 * p] != 0, w[sel, 2 * p + 1] != 0
            if not have_a.any():
                break
            cn = (a - prev)[have_a].mean()
            bn = (b - a)[have_b].mean() if have_b.any() else float("nan")
            row.append(f"p{p}: {cn:.1f}+{bn:.1f}" + ("" if have_b.all() else f" ({int(have_b.sum())}/{int(sel.sum())} ran the bit nodes)"))
            # next pass' check-node phase starts behind this pass' bit-node barrier (+ a parity sweep when one ran)
            prev = np.where(have_b, b, a)
        eager = [(int(((w[sel, 20] >> p) & 1).sum())) for p in range(1, 10)]
        bad = [int(np.median(w[sel, 22 + p])) for p in range(1, min(k + 1, 9) + 1)]
        print(f"   n_iter = {k}: {int(sel.sum())} blocks, total {dur[sel].mean():.2f} us | check-node + bit-node phase per pass (us; a phase that "
              f"follows a parity sweep includes it): " + ", ".join(row))
        print(f"      parity sweeps run behind pass 1..9 by this many blocks: {eager}; median unsatisfied lanes counted in pass 1..: {bad}")
    # per CU: order, gaps, busy
    gaps, busy, last_end, per_cu_n = [], np.zeros(ncu), np.zeros(ncu), np.zeros(ncu, dtype=int)
    for c in range(ncu):
        idx = np.where(cui == c)[0]
        idx = idx[np.argsort(start[idx])]
        per_cu_n[c] = len(idx)
        busy[c] = dur[idx].sum()
        last_end[c] = end[idx].max()
        gaps += [start[idx[i + 1]] - end[idx[i]] for i in range(len(idx) - 1)]
    gaps = np.array(gaps)
    first = np.array([start[cui == c].min() for c in range(ncu)])
    print(f"   first workgroup of a CU starts at {first.mean():.2f} us (max {first.max():.2f}); gap between consecutive workgroups of a CU: "
          f"mean {gaps.mean():.2f} us, median {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f}, max {gaps.max():.2f}"
          f" (negative = overlap: {int((gaps < 0).sum())} of {len(gaps)})")
    print(f"   workgroups per CU: {np.bincount(per_cu_n).tolist()} (index = count); busy per CU mean {busy.mean():.1f} us = {busy.mean() / span:.2f} of the span; "
          f"CU finishes at mean {last_end.mean():.1f} us, earliest {last_end.min():.1f}, latest {last_end.max():.1f}")
    ideal = dur.sum() / ncu
    print(f"   sum of workgroup durations / CUs = {ideal:.1f} us  |  + mean gaps {ideal + gaps.mean() * (n / ncu - 1) + first.mean():.1f} us  |  span {span:.1f} us")
    # list scheduling of the measured durations (+ the measured mean gap) in launch order: on all CUs, and per XCD
    def greedy(durs, servers, gap):
        free = np.zeros(servers)
        for x in durs:
            i = int(np.argmin(free))
            free[i] += x + gap
        return free.max() - gap
    order = np.argsort(d[:, 2].astype(np.int64), kind="stable")
    g = max(0.0, float(np.median(gaps)))
    print(f"   list scheduling of these durations, blocks in index order, gap {g:.2f} us: {greedy(dur, ncu, g):.1f} us on {ncu} CUs; "
          f"sorted longest first: {greedy(np.sort(dur)[::-1], ncu, g):.1f} us; with no gap: {greedy(dur, ncu, 0.0):.1f} us")
    return dict(span=span, dur=dur, gaps=gaps, n_iter=n_iter)


def main():
    import torch
    import bench as B
    import openairinterface5g_amd as pkg
    pkg.LDPCinit()
    m = pkg.ldpc
    BG, Z, R = [int(x) for x in os.environ.get("DEC_TRACE_CODE", "1,384,13").split(",")]
    n = B.BATCH
    ncol = m.NCOLS[(BG, R)]
    K = (22 if BG == 1 else 10) * Z
    out = torch.zeros((n, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
    it = torch.zeros(n, dtype=torch.int32, device="cuda")
    print(f"# NRLDPC_HIP_PERSIST={os.environ.get('NRLDPC_HIP_PERSIST', '(default)')}  BG{BG} Zc={Z} R={R}, {n} blocks, numMaxIter {B.MAX_ITER}; "
          "every launch of this process is the diagnostic instantiation (stamps by thread 0); the sixth one is analysed")
    for label, snr, seed in (("operating point", 1.0 if R in (13, 15) else 4.0, 2000), ("fixed work, -12 dB", -12.0, 1000)):
        g = torch.Generator(device="cuda").manual_seed(seed)
        info = torch.randint(0, 256, (n, K // 8), dtype=torch.uint8, device="cuda", generator=g)
        coded = torch.empty((n, (66 if BG == 1 else 50) * Z), dtype=torch.uint8, device="cuda")
        pkg.encode_batch_device(BG, Z, info, coded)
        coded = coded[:, :(ncol - 2) * Z]
        sigma = 1.0 / np.sqrt(2.0 * 10.0 ** (snr / 10.0))
        y = 1.0 - 2.0 * coded.float() + sigma * torch.randn(coded.shape, device="cuda", generator=g)
        llr = torch.zeros((n, ncol * Z), dtype=torch.int8, device="cuda")
        llr[:, 2 * Z:] = torch.clamp(torch.floor(y / (sigma / 16.0)), -128, 127).to(torch.int8)
        for _ in range(6):
            pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=B.MAX_ITER)      # traced (and synchronised) launches
        d = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 32)
        if len(sys.argv) > 1:
            d.tofile(f"{sys.argv[1]}_{'op' if snr > 0 else 'fixed'}.bin")
        analyse(d, f"{label} (Es/N0 = {snr} dB)", 0.0)
    os.unlink(TRACE)


if __name__ == "__main__":
    main()
