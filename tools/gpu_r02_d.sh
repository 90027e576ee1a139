#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
bash tools/gpu_ab_lib.sh libldpc_hip_old.so libldpc_hip.so 3
timeout 120 tools/ubench/hbm_rw.bin | tee $O/hbm_rw.txt
timeout 900 python tools/bench_extra.py > $O/bench_extra.json 2> $O/bench_extra.err; echo "bench_extra rc=$?"; python -c "
import json; d=json.load(open('$O/bench_extra.json'))
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
