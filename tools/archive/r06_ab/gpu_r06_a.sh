#!/bin/bash
# round 6, call A: where the operating point's time goes
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/dec_trace.py gpurun_out/r06/dec_trace > gpurun_out/r06/dec_trace.txt 2>&1
timeout 600 python tools/ab_fast.py head: head2: > gpurun_out/r06/ab_head.txt 2>&1
timeout 600 python bench.py --steps 50 > gpurun_out/r06/bench_head.json 2> gpurun_out/r06/bench_head.err
tail -3 gpurun_out/r06/dec_trace.txt; cat gpurun_out/r06/ab_head.txt | head -5
