#!/bin/bash
# host-array transport-block calls after the parallel / pipelined bounce: parity first, then the slot's PCIe-inclusive cost
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
( time python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_schsim.py -m gpu -q -x ) > gpurun_out/r05/pytest_host.txt 2>&1
tail -4 gpurun_out/r05/pytest_host.txt
python tools/slot_chain_host.py 10 > gpurun_out/r05/slot_chain_host.json 2> gpurun_out/r05/slot_chain_host.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05/slot_chain_host.json"))
print("link", round(d["link_GBps"],1), "GB/s")
for k,v in d.items():
    if isinstance(v,dict) and "round0_ms" in v: print("%-40s r0 %6.2f ms  r1 %6.2f ms  ack %s" % (k, v["round0_ms"], v["round1_ms"], v.get("all_ack")))
PY
NRLDPC_HIP_BOUNCE_THREADS=1 python tools/slot_chain_host.py 6 > gpurun_out/r05/slot_chain_host_one_thread.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r05/slot_chain_host_one_thread.json"))
for k,v in d.items():
    if isinstance(v,dict) and "round0_ms" in v and "host_harq" in k: print("1 thread %-32s r0 %6.2f ms  r1 %6.2f ms" % (k, v["round0_ms"], v["round1_ms"]))
PY
