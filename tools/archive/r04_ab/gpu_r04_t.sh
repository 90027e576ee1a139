#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/t; mkdir -p $O; export TMPDIR=/tmp
echo "== de-matching launch (four-launch path): r03 library vs HEAD with the streaming loop that never waits for its own stores"
for L in tools/ab/libldpc_hip_r03.so openairinterface5g_amd/lib/libldpc_hip.so tools/ab/libldpc_hip_r03.so openairinterface5g_amd/lib/libldpc_hip.so; do
  for MODE in "" "0.18 retx"; do
    cd /tmp && NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L NRLDPC_HIP_TB_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_ab" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 40 $MODE > /dev/null 2>&1
    cd "$GRAFT_REPO_ROOT"; python - <<PY
import csv, glob
for f in glob.glob("$O/prof_ab/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dematch" in r["Name"]:
            print("$(basename $L) [$MODE] tb_rx_dematch_kernel: calls %s avg %.1f us min %.1f max %.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
    rm -rf $O/prof_ab
  done
done 2>&1 | tee $O/ab_dematch_store_waits.txt
echo "fused slot (HEAD): $(timeout 300 python tools/slot_chain.py 50 | cut -c40-90)  retx: $(timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-90)" | tee -a $O/ab_dematch_store_waits.txt
echo "unfused slot (HEAD): $(NRLDPC_HIP_TB_FUSED=0 timeout 300 python tools/slot_chain.py 50 | cut -c40-90)" | tee -a $O/ab_dematch_store_waits.txt
timeout 300 python tools/tb_latency.py 2>&1 | grep " 1 TB\| 16 TB\| 64 TB" | tee -a $O/ab_dematch_store_waits.txt
NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | grep "prologue steps\|start-order\|kernel span" | tee -a $O/ab_dematch_store_waits.txt; rm -f $O/trace.bin
timeout 600 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_gpu_offload.py -m gpu -q 2>&1 | tail -3
