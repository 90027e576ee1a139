#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -k "every_lifting or several_blocks or iteration_caps or crc_early or survey" 2>&1 | tail -6
CODES="1,30,13 1,26,13 1,22,13 1,15,13 1,7,13 1,2,13 2,30,15 2,15,15 2,7,15 2,3,15"
{
echo "# generic kernel (kernel 1), 32768 blocks"; SWEEP_KERNEL=1 timeout 300 python tools/sweep_codes.py 32768 $CODES 2>&1 | grep -v amdgpu.ids
echo "# four blocks interleaved byte-wise, several groups per workgroup (kernel 0 = automatic), 32768 blocks"; SWEEP_KERNEL=0 timeout 300 python tools/sweep_codes.py 32768 $CODES 2>&1 | grep -v amdgpu.ids
} | tee $O/small_z_interleaved.txt
