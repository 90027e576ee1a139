#!/bin/bash
# first GPU contact: smoke, parity tests, bench, kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E 'Marketing Name|Compute Unit|gfx' | head -6 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15 | tee gpurun_out/smoke.log
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -5 | tee gpurun_out/bench.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof1" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-operating-point > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1; tail -3 "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log"
find "$GRAFT_REPO_ROOT/gpurun_out/prof1" -name '*stats*' | head
