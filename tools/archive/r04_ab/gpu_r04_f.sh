#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/f; mkdir -p $O; export TMPDIR=/tmp
echo "== A/B de-matching kernel variants on this box"
for L in tools/ab/libldpc_hip_r03.so tools/ab/libldpc_hip_u2_b1.so tools/ab/libldpc_hip_u2_b2.so tools/ab/libldpc_hip_u4_b1.so openairinterface5g_amd/lib/libldpc_hip.so tools/ab/libldpc_hip_r03.so; do
  for MODE in "" "0.18 retx"; do
    cd /tmp && NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L NRLDPC_HIP_TB_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_ab" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 $MODE > /dev/null 2>&1
    cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_ab -name '*kernel_stats.csv' | head -1); echo "$(basename $L) [$MODE] dematch avg ns: $(grep dematch "$f" | cut -d, -f4)  decoder: $(grep 'dec_fast_kernel' "$f" | cut -d, -f4)"; rm -rf $O/prof_ab
  done
  case $L in *r03*) ;; *) echo "   fused slot: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tools/slot_chain.py 50 | cut -c40-90)  retx: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-90)";; esac
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
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_tb_resident.py tests/test_gpu_tb_chain.py -m gpu -q --maxfail=8 2>&1 | tail -30 > $O/pytest.log; tail -4 $O/pytest.log
