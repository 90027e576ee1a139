#!/usr/bin/env python3
"""How much bit-node work could start under the check-node tail if the phase barrier of the fast decoder were replaced by
per-column readiness counters?  Model on the code's real graph: the check-node queue hands out the
rows most expensive first (degree 19 rows, then by degree); a row's finish time = its position in the cumulative edge work
spread over the workgroup's waves; a column is ready when the LAST row that touches it has finished; the next pass' rows
are ready when the last of their columns has been updated.  Prints, for BG1 R = 1/3, which fraction of the bit-node work is
ready at which fraction of the check-node phase, and the same in the other direction.
  python tools/bn_overlap_model.py [BG] [R]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle_lib as O

BG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = int(sys.argv[2]) if len(sys.argv) > 2 else 13
g = O.graph(BG, 384, R)
nrows, ncols, ncore = g.nrows, g.ncols, g.ncore
rows = [[g.col[e] for e in range(g.row_ptr[r], g.row_ptr[r + 1])] for r in range(nrows)]
deg = np.array([len(r) for r in rows])
order = sorted(range(nrows), key=lambda r: -deg[r])            # the queue: most expensive rows first
work = deg[order].astype(float)                                  # a row's check-node work ~ its edges
finish = np.cumsum(work) / work.sum()                            # fraction of the CN phase at which the row is done (ideal queue)
row_done = {r: finish[i] for i, r in enumerate(order)}
col_rows = {c: [r for r in range(nrows) if c in rows[r]] for c in range(ncore)}
col_ready = {c: max(row_done[r] for r in col_rows[c]) for c in range(ncore)}
col_work = {c: len(col_rows[c]) for c in range(ncore)}
tot = sum(col_work.values())
print(f"BG{BG} R{R}: {nrows} rows, {ncore} core columns, {deg.sum()} edges; bit-node work by column readiness")
for x in (0.5, 0.7, 0.8, 0.9, 0.95, 0.99):
    ready = sum(w for c, w in col_work.items() if col_ready[c] <= x)
    print(f"  at {x:4.0%} of the check-node phase: {ready / tot:5.1%} of the bit-node work is ready "
          f"({sum(1 for c in col_ready if col_ready[c] <= x)} of {ncore} columns)")
first = min(col_ready.values())
print(f"  the earliest column is ready at {first:.1%}; columns 0 and 1 (degree {col_work[0]}, {col_work[1]}: the two longest bit-node tasks) at "
      f"{col_ready[0]:.1%} / {col_ready[1]:.1%}")
# other direction: bit-node tasks longest first; a row of the next pass is ready when all its core columns are updated
cord = sorted(range(ncore), key=lambda c: -col_work[c])
cfin = np.cumsum([col_work[c] for c in cord]) / tot
cdone = {c: cfin[i] for i, c in enumerate(cord)}
row_ready = {r: max(cdone[c] for c in rows[r] if c < ncore) for r in range(nrows)}
for x in (0.5, 0.8, 0.9, 0.95):
    ready = sum(deg[r] for r in range(nrows) if row_ready[r] <= x)
    print(f"  at {x:4.0%} of the bit-node phase: {ready / deg.sum():5.1%} of the next pass' check-node work is ready")
print(f"  the four degree-19 rows (first in the queue) are ready at {max(row_ready[r] for r in order[:4]):.1%}")
