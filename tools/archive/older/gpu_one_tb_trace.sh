#!/bin/bash
# kernel timeline of one transport block through both chain calls (start / end relative to the call's first kernel)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python "$GRAFT_REPO_ROOT/tools/one_tb_trace.py" > /dev/null 2>&1)
f=$(find /tmp/tr1 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/one_tb_timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("tb_rx_dematch")]
a = idx[20]; t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:idx[21]]:
    print("%7.1f %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:60]))
PY
