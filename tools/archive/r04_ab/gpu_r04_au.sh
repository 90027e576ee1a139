#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/au; mkdir -p $O; export TMPDIR=/tmp
{
echo "== per-segment call latency against the idle time between calls (BG1 Zc=384 R=1/3, 2 passes; the GPU's own wall-clock stamps)"
for G in 0 50 200 1000 5000; do SRV_GAP_US=$G timeout 120 python tools/srv_breakdown.py 600 2>&1 | tail -1; done
echo "sclk under a tight loop of calls: $( (timeout 20 python tools/srv_breakdown.py 200000 > /dev/null 2>&1 &) ; sleep 6; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -1 | tr -s ' ')"
} 2>&1 | grep -v amdgpu.ids | tee $O/call_latency_vs_idle_gap.txt
