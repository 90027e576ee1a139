#!/bin/bash
# round 5, VERDICT r04 item 4: what binds the encoder and the fused TX kernel -- kernel trace, SQ counters, per-phase clocks
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05/$1; mkdir -p $O; export TMPDIR=/tmp
LIB=${2:-}
[ -n "$LIB" ] && export NRLDPC_HIP_LIB="$GRAFT_REPO_ROOT/$LIB"
python tools/tx_enc_driver.py 30 both > $O/tx_enc_events.txt 2>&1; cat $O/tx_enc_events.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 30 both > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
python tools/prof_kernels.py $O/trace --csv $O/tx_enc_kernel_stats.csv | grep -E "enc_packed|tb_tx" | tee $O/tx_enc_kernel_stats.txt
run() {
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$1" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 5 both > "$GRAFT_REPO_ROOT/$O/pmc_$1.log" 2>&1 )
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_$1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        if "enc_packed" in k or "tb_tx" in k:
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-40s %-28s n=%d mean=%.0f" % (k, c, len(v), sum(v) / len(v)))
PY
}
{
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
run grbm "GRBM_GUI_ACTIVE"
} | tee $O/tx_enc_pmc.txt
rm -rf $O/trace $O/pmc_sq1 $O/pmc_sq2 $O/pmc_grbm
if [ -f tools/ab/libldpc_hip_tbt.so ]; then
  NRLDPC_HIP_LIB="$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_tbt.so" python tools/tb_tx_timing.py 64 | tee $O/tb_tx_phases.txt
fi
