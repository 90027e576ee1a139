#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/k; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py -m gpu -q --maxfail=5 2>&1 | tail -25 > $O/pytest.log; tail -5 $O/pytest.log
W() { env "$@" timeout 200 python tools/small_tbs.py --worker ${C:-64} 30 0.35 2>/dev/null | tail -1; }
for C in 64 256; do export C
echo "== small TBS x$C"
echo "r03 plan (unfused, class launches): $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_FILL=0)"
echo "unfused, filled rounds:             $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_FILL=1)"
echo "fused, class launches:              $(W NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_FILL=0)"
echo "fused, filled rounds (default):     $(W NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_FILL=1)"
echo "one launch (r02 plan):              $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_CLASSES=0 NRLDPC_HIP_TB_MULTI=0)"
done 2>&1 | tee $O/small_tbs.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tr" -- python "$GRAFT_REPO_ROOT/tools/small_tbs.py" --worker 64 4 0.35 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; echo "== kernel timeline of the last call (default plan)"; python - <<PY | tee $O/small_tbs_timeline.txt
import csv, glob
ev=[]
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Workgroup_Size_X", r.get("Workgroup_Size","?")), r.get("Grid_Size_X", r.get("Grid_Size","?"))))
ev.sort()
idx=[i for i,e in enumerate(ev) if "tb_rx" in e[2] or "ldpc_dec" in e[2]]
last=idx[-1]; start=last
while start>0 and ev[start][0]-ev[start-1][1] < 20_000 and ("tb_rx" in ev[start-1][2] or "ldpc_dec" in ev[start-1][2]): start-=1
t0=ev[start][0]
for s,e,n,wg,grid in ev[start:last+1]:
    print(f"{(s-t0)/1e3:8.1f} -> {(e-t0)/1e3:8.1f} us  wg {wg:>5} grid {grid:>8}  {n}")
PY
rm -rf $O/tr
