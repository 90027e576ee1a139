#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/h; mkdir -p $O; export TMPDIR=/tmp
show() { python - <<PY
import json
d=json.load(open("$1"))
for k,v in list(d.items())+[("one_tb/"+a,b) for a,b in d["one_tb"].items()]:
    if isinstance(v,dict) and "round0_ms" in v: print(k, {a:round(b,3) if isinstance(b,float) else b for a,b in v.items() if a in ("round0_ms","round1_ms","frac_of_link_round0","frac_of_link_round1","all_ack")})
    elif not isinstance(v,dict) and k in ("link_h2d_ms","link_GBps"): print(k,v)
PY
}
echo "== host LLRs (auto)"; timeout 600 python tools/slot_chain_host.py 10 2>/dev/null > $O/slot_chain_host.json; show $O/slot_chain_host.json
for K in 3 4 8; do echo "== host LLRs, $K chunks"; NRLDPC_HIP_TB_HOST_CHUNKS=$K timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_k$K.json; show $O/slot_chain_host_k$K.json | grep "^pinned_llr/library"; done
echo "== host LLRs, always pull"; NRLDPC_HIP_TB_PULL=2 timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_pull.json; show $O/slot_chain_host_pull.json | grep "pinned_llr"
echo "== host LLRs, never pull"; NRLDPC_HIP_TB_PULL=0 timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host_nopull.json; show $O/slot_chain_host_nopull.json | grep "pinned_llr"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tl" -- python "$GRAFT_REPO_ROOT/tools/host_timeline.py" 6 2>/dev/null | tail -2
cd "$GRAFT_REPO_ROOT"; python tools/host_timeline_parse.py $O/tl | tee $O/host_timeline_auto.txt | head -30; rm -rf $O/tl
echo "== all tests"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 2>&1 | tail -30 > $O/pytest.log; tail -5 $O/pytest.log
