#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -k "every_lifting or several_blocks or iteration_caps or crc_early" 2>&1 | tail -6
CODES="1,64,13 1,32,13 1,16,13 1,8,13 1,48,13 1,24,13 2,64,15 2,32,15 2,16,15 2,8,15 2,48,13 1,64,89"
{
echo "# one block per workgroup (kernel 3), 32768 blocks"; SWEEP_KERNEL=3 timeout 300 python tools/sweep_codes.py 32768 $CODES 2>&1 | grep -v amdgpu.ids
echo "# several blocks per workgroup (kernel 0 = automatic), 32768 blocks"; SWEEP_KERNEL=0 timeout 300 python tools/sweep_codes.py 32768 $CODES 2>&1 | grep -v amdgpu.ids
} | tee $O/small_z_multiblock.txt
