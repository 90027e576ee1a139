#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/n; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_gpu_offload.py -m gpu -q --maxfail=5 2>&1 | tail -6
for V in 1 0; do
  echo "== NRLDPC_HIP_CLLR_LDS=$V"
  echo "slot: $(NRLDPC_HIP_CLLR_LDS=$V timeout 300 python tools/slot_chain.py 50 | cut -c40-90)  retx: $(NRLDPC_HIP_CLLR_LDS=$V timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-90)"
  NRLDPC_HIP_CLLR_LDS=$V NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | grep "start-order\|kernel span"; rm -f $O/trace.bin
  NRLDPC_HIP_CLLR_LDS=$V timeout 300 python tools/tb_latency.py 2>&1 | grep " 1 TB\| 16 TB"
  NRLDPC_HIP_CLLR_LDS=$V timeout 300 python - <<'PY' 2>/dev/null
import sys, time
sys.path.insert(0, ".")
import torch, numpy as np
import openairinterface5g_amd as pkg
exec(open("tools/bench_extra.py").read().split("# ---- config 3")[0].split("res = {}")[1])
m = pkg.ldpc; pkg.LDPCinit()
for (BG, Z, R) in ((1, 384, 89), (1, 384, 23), (2, 384, 23), (1, 192, 13), (2, 208, 15), (1, 384, 13)):
    _, llr = noisy_llr(BG, Z, R, 1024, -12.0, 11 * R + BG)
    out = torch.zeros((1024, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
    it = torch.zeros(1024, dtype=torch.int32, device="cuda")
    dt = timeit(lambda: pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8), 50)
    print(f"  BG{BG} Z{Z} R{R}: {dt*1e3:.4f} ms / 1024 blocks x 9 passes")
PY
done 2>&1 | tee $O/ab_cllr_lds.txt
