#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for L in libldpc_hip.so libldpc_hip_noBN.so libldpc_hip_noCN.so; do
  NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/$L timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L kernel %.4f ms' % d['roofline']['kernel_avg_ms'])"
done
