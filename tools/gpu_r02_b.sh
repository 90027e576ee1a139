#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step')}); print(d['strong_scaling_slot']); print(d['cpu_baseline'])"; tail -3 $O/bench.err
bash tools/gpu_ab_lib.sh libldpc_hip_old.so libldpc_hip.so 2
