#!/bin/bash
# A/B two builds of the library with bench.py (kernel time of the fixed-work launch, operating-point rate)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for L in ${LIBS:-libldpc_hip.so libldpc_hip_mode1.so}; do
  NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/$L timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L kernel %.4f ms  op %.1f Gb/s' % (d['roofline']['kernel_avg_ms'], d['operating_point']['gbps']))"
done; done
