#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/e; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_tb_resident.py tests/test_gpu_tb_chain.py tests/test_bench.py -m gpu -q --maxfail=8 2>&1 | tail -30 > $O/pytest.log; tail -6 $O/pytest.log
echo "== A/B de-matching kernel on this box (unfused path): r03 library, then this one"
for L in tools/ab/libldpc_hip_r03.so openairinterface5g_amd/lib/libldpc_hip.so; do
  cd /tmp && NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L NRLDPC_HIP_TB_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_ab" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 0.18 retx > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_ab -name '*kernel_stats.csv' | head -1); echo "$L"; grep -E "dematch|dec_fast_kernel|assemble|verdict" "$f" | cut -d, -f1-4,6,7 | cut -c1-150; cp "$f" $O/ab_$(basename $L .so)_kernel_stats.csv; rm -rf $O/prof_ab
done 2>&1 | tee $O/ab_dematch.txt
echo "== host LLRs (auto)"; timeout 600 python tools/slot_chain_host.py 10 2>/dev/null > $O/slot_chain_host.json; python - <<PY
import json
d=json.load(open("$O/slot_chain_host.json"))
for k,v in list(d.items())+[("one_tb/"+a,b) for a,b in d["one_tb"].items()]:
    if isinstance(v,dict) and "round0_ms" in v: print(k, {a:round(b,3) if isinstance(b,float) else b for a,b in v.items()})
    elif not isinstance(v,dict): print(k,v)
PY
for K in 3 4 8; do echo "== host LLRs, $K chunks"; NRLDPC_HIP_TB_HOST_CHUNKS=$K timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_k$K.json; grep -A3 '"pinned_llr/library_harq"' $O/slot_chain_host_k$K.json | head -4; done
echo "== host LLRs, always pull"; NRLDPC_HIP_TB_PULL=2 timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_pull.json; grep -A3 '"pinned_llr/library_harq"' $O/slot_chain_host_pull.json | head -8
