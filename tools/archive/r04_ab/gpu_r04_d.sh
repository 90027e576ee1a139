#!/bin/bash
# round 4, call D: host path (chunked copy lane / pull), epilogue with generation slots, bench with loopback + chain_roofline
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/d; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_tb_resident.py tests/test_gpu_tb_chain.py tests/test_gpu_offload.py tests/test_schsim.py tests/test_bench.py -m gpu -q --maxfail=8 2>&1 | tail -40 > $O/pytest.log; tail -8 $O/pytest.log
echo "== slot"; timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain.txt; timeout 300 python tools/slot_chain.py 30 0.18 retx | tee -a $O/slot_chain.txt
NRLDPC_HIP_TB_FUSED=0 timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain_unfused.txt
timeout 300 python tools/tb_latency.py 2>&1 | grep TB | tee $O/tb_latency.txt
NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | tee $O/wg_trace.txt; rm -f $O/trace.bin
echo "== host LLRs (auto)"; timeout 600 python tools/slot_chain_host.py 10 2>/dev/null > $O/slot_chain_host.json; python - <<PY
import json
d=json.load(open("$O/slot_chain_host.json"))
for k,v in list(d.items())+[("one_tb/"+a,b) for a,b in d["one_tb"].items()]:
    if isinstance(v,dict) and "round0_ms" in v: print(k, {a:round(b,3) if isinstance(b,float) else b for a,b in v.items()})
    elif not isinstance(v,dict): print(k,v)
PY
echo "== host LLRs, always pull"; NRLDPC_HIP_TB_PULL=2 timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_pull.json; grep -A3 '"pinned_llr/library_harq"' $O/slot_chain_host_pull.json | head -8
for K in 3 4 8; do echo "== host LLRs, $K chunks"; NRLDPC_HIP_TB_HOST_CHUNKS=$K timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_k$K.json; grep -A3 '"pinned_llr/library_harq"' $O/slot_chain_host_k$K.json | head -4; done
echo "== bench (loopback)"; BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; python - <<PY
import json
l=json.loads([x for x in open("$O/bench_dist1.json") if x.startswith("{")][-1])
print(json.dumps({k:l[k] for k in ("value","ms_per_step","chain_roofline","strong_scaling_slot","build")},indent=1)[:3000]); print(l["roofline"]["binding_resource"])
PY
tail -3 $O/bench_dist1.err
