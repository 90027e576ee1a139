#!/bin/bash
# round 5, first GPU call: whole -m gpu suite (timed), the reference-compiled pin on the GPU box, the bench line
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r05/pytest_gpu.txt
python -m pytest tests/test_ref_pin.py tests/test_ref_fixtures.py -q > gpurun_out/r05/pytest_ref_pin_on_gpu_box.txt 2>&1
tail -3 gpurun_out/r05/pytest_ref_pin_on_gpu_box.txt
python bench.py > gpurun_out/r05/bench_baseline.json 2> gpurun_out/r05/bench_baseline.err
cat gpurun_out/r05/bench_baseline.json
