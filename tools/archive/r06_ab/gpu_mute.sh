#!/bin/bash
# round 6: extension-row items whose degree-1 bits have channel LLR 0 are not run in CRC-stop launches (LDPC_MUTE_ITEMS) -- against a
# build without it (tools/build_variant.sh nomute -DLDPC_MUTE_ITEMS=0)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06/mute; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_schsim.py tests/test_gpu_decoder.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python tools/soak_tb.py 40 4 2>&1 | tail -1 | tee $O/soak_tb.txt
timeout 300 python tools/soak.py 40 777 2>&1 | tail -1 | tee $O/soak.txt
B="$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_nomute.so"
for rep in 1 2 3; do
  echo "== base rep $rep"; NRLDPC_HIP_LIB=$B python tools/slot_chain.py 2>&1 | tail -1; NRLDPC_HIP_LIB=$B python tools/slot_chain.py 20 0.18 retx 2>&1 | tail -1
  echo "== mute rep $rep"; python tools/slot_chain.py 2>&1 | tail -1; python tools/slot_chain.py 20 0.18 retx 2>&1 | tail -1
done 2>&1 | grep -v amdgpu.ids | tee $O/ab_mute_slot.txt
echo "== small TBS mix"; NRLDPC_HIP_LIB=$B python tools/small_tbs.py 2>&1 | tail -5 | head -1; python tools/small_tbs.py 2>&1 | tail -5 | head -1
