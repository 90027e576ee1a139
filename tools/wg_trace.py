#!/usr/bin/env python3
"""Reads a per-workgroup trace of the fused segment kernel (NRLDPC_HIP_TB_TRACE=<file>, last launch of the process) and
prints where the workgroups ran and how their phases fell: CU placement by launch order, prologue / decode / epilogue
durations by round, and how many of a CU's workgroups were in their prologue at the same time.
  python tools/wg_trace.py <file> [n_cus]"""
import sys
import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
hw, xcc = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64)
t = d[:, 2:6].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0                      # us (wall_clock64 ticks at 100 MHz)
cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
ids = {c: i for i, c in enumerate(sorted(set(cu.tolist())))}
cui = np.array([ids[c] for c in cu.tolist()])
n = len(d)
print(f"{n} workgroups on {len(ids)} distinct CUs; kernel span {t[:, 3].max():.1f} us")
print("first 24 workgroups -> (xcc, se, sh, cu):", [(int(xcc[i]), int((hw[i] >> 13) & 7), int((hw[i] >> 12) & 1), int((hw[i] >> 8) & 15)) for i in range(24)])
print("CU index of workgroups 0..31:", cui[:32].tolist())
first = {}
for i in range(n):
    first.setdefault(cui[i], []).append(i)
pairs = [v[:2] for v in first.values() if len(v) >= 2]
print("first two workgroups of a CU differ in launch index by:", np.bincount(np.minimum(np.array([b - a for a, b in pairs]), 600))[:8].tolist(), "...",
      "median", int(np.median([b - a for a, b in pairs])))
pro, dec, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
if d[:, 7].any():      # finer stamps of the prologue
    ps = (d[:, 7:11].astype(np.int64) - t0) / 100.0
    steps = np.stack([ps[:, 0] - t[:, 0], ps[:, 1] - ps[:, 0], ps[:, 2] - ps[:, 1], ps[:, 3] - ps[:, 2], t[:, 1] - ps[:, 3]], axis=1)
    print("prologue steps (mean us over all workgroups): clear LDS image %.2f | scatter LLRs %.2f | stream soft buffer + decoder input %.2f | "
          "stores visible (fence + barrier) %.2f | (kernel wrapper -> block body) %.2f" % tuple(steps.mean(axis=0)))
order = np.argsort(t[:, 0])
q = max(1, n // 4)
for k in range(0, n, q):
    sel = order[k:k + q]
    print(f"start-order {k:5d}..{k + len(sel):5d}: start {t[sel, 0].mean():7.1f}  prologue {pro[sel].mean():6.2f} (max {pro[sel].max():6.2f})  "
          f"decode {dec[sel].mean():6.2f}  epilogue {epi[sel].mean():5.2f}  total {(t[sel, 3] - t[sel, 0]).mean():6.2f} us")
# overlap: for every workgroup, was the other resident workgroup of its CU in its prologue during (most of) this one's prologue?
both = 0
for c, v in first.items():
    for i in v:
        for j in v:
            if i < j:
                lo, hi = max(t[i, 0], t[j, 0]), min(t[i, 1], t[j, 1])
                if hi - lo > 0.5 * min(pro[i], pro[j]):
                    both += 1
print(f"pairs of workgroups of one CU whose prologues overlap by more than half: {both} of {n} workgroups")
