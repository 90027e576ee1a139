#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 > $O/ldpctest_hip_8448.txt 2>&1; tail -3 $O/ldpctest_hip_8448.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt
timeout 300 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
