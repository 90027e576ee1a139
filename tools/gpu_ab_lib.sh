#!/bin/bash
# A/B of two builds of the library on the headline bench: tools/gpu_ab_lib.sh <libA> <libB> [reps]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
for i in $(seq 1 ${3:-3}); do
  for L in $1 $2; do
    echo "$L: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/$L timeout 300 python bench.py --steps 50 --no-cpu-baseline --no-operating-point 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["value"],2), round(d["roofline"]["kernel_avg_ms"],4))')"
  done
done | tee $O/ab_lib.txt
