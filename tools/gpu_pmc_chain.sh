#!/bin/bash
# SQ counters of the chain kernels (own pass per group, --kernel-trace only): tools/gpu_pmc_chain.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
run() {
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_chain_$1" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 5 > "$GRAFT_REPO_ROOT/$O/pmc_chain_$1.log" 2>&1
  cd "$GRAFT_REPO_ROOT"; python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_chain_$1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        if "dematch" in k or "assemble" in k or "fused" in k:
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-40s %-28s n=%d mean=%.0f" % (k, c, len(v), sum(v) / len(v)))
PY
}
{
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
run grbm "GRBM_GUI_ACTIVE"
} | tee $O/chain_pmc_sq.txt
