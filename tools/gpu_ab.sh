#!/bin/bash
# A/B runs of bench.py with different waves-per-workgroup of the fast kernel (and the parity tests once)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for W in ${WAVES:-16 12 8}; do
  echo "== waves $W"
  NRLDPC_HIP_FAST_WAVES=$W timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W=$W value %.2f Gb/s  kernel %.4f ms  op %.2f Gb/s' % (d['value'], d['roofline']['kernel_avg_ms'], d['operating_point']['gbps']))" | tee -a gpurun_out/ab.log
done
if [ -n "$TESTS" ]; then NRLDPC_HIP_FAST_WAVES=${TESTW:-12} timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5; fi
