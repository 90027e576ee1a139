#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/q; mkdir -p $O; export TMPDIR=/tmp
echo "== de-matching launch (four-launch path, first transmissions), symbols requested ahead: variants on one box"
for L in tools/ab/libldpc_hip_r03.so tools/ab/libldpc_hip_dm_u0.so tools/ab/libldpc_hip_dm_u1.so tools/ab/libldpc_hip_dm_u2.so openairinterface5g_amd/lib/libldpc_hip.so tools/ab/libldpc_hip_r03.so; do
    cd /tmp && NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L NRLDPC_HIP_TB_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_ab" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 40 > /dev/null 2>&1
    cd "$GRAFT_REPO_ROOT"; python - <<PY
import csv, glob
for f in glob.glob("$O/prof_ab/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dematch" in r["Name"]:
            print("$(basename $L) tb_rx_dematch_kernel: calls %s avg %.1f us min %.1f max %.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
    rm -rf $O/prof_ab
done 2>&1 | tee $O/ab_dematch_ahead.txt
