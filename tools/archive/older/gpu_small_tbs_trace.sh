#!/bin/bash
# timeline of one UL-SCH call over the small transport blocks: kernel start/end relative to the de-matching kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
C=${1:-64}
for mode in "1 1 1" "1 1 0" "0 0 0"; do
  set -- $mode
  (cd /tmp && NRLDPC_HIP_TB_MULTI=$1 NRLDPC_HIP_TB_CLASSES=$2 NRLDPC_HIP_TB_OVERLAP=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tr" -- python "$GRAFT_REPO_ROOT/tools/small_tbs.py" --worker $C 20 0.35 > /dev/null 2>&1)
  f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo "== NRLDPC_HIP_TB_MULTI=$1 NRLDPC_HIP_TB_CLASSES=$2 NRLDPC_HIP_TB_OVERLAP=$3" | tee -a $O/small_tbs_timeline.txt
  python - "$f" <<'PY' | tee -a $O/small_tbs_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the 15th de-matching launch and everything up to the next one
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("tb_rx_dematch")]
a, b = idx[14], idx[15]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f %8.1f us  grid %7s wg %5s lds %6s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
          r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:60]))
PY
  rm -rf $O/tr
done
