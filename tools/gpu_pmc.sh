#!/bin/bash
# PMC counter passes for the decoder kernel (separate runs per counter group; --kernel-trace only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
KERN=${KERN:-0}
run() { # name, counters
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$1" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --kernel $KERN > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$1.log" 2>&1
  tail -2 "$GRAFT_REPO_ROOT/gpurun_out/pmc_$1.log" | cut -c1-200
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run grbm "GRBM_GUI_ACTIVE"
find "$GRAFT_REPO_ROOT/gpurun_out" -name '*counter_collection.csv' | head
