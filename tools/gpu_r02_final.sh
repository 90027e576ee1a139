#!/bin/bash
# whole GPU suite, then the round's evidence (tools/gpu_r02_profiles.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02/final; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02/final/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_r02_profiles.sh
