#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/i; mkdir -p $O; export TMPDIR=/tmp
W() { env "$@" timeout 200 python tools/small_tbs.py --worker 64 30 0.35 2>/dev/null | tail -1; }
echo "== small TBS x64"
echo "unfused serial (r03 plan): $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_OVERLAP=0)"
echo "unfused overlap:           $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_OVERLAP=1)"
echo "fused serial:              $(W NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_OVERLAP=0)"
echo "fused overlap (default):   $(W NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_OVERLAP=1)"
echo "one launch (r02 plan):     $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_OVERLAP=0 NRLDPC_HIP_TB_CLASSES=0 NRLDPC_HIP_TB_MULTI=0)"
W2() { env "$@" timeout 200 python tools/small_tbs.py --worker 256 20 0.35 2>/dev/null | tail -1; }
echo "== small TBS x256"
echo "unfused serial (r03 plan): $(W2 NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_OVERLAP=0)"
echo "fused serial:              $(W2 NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_OVERLAP=0)"
echo "fused overlap (default):   $(W2 NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_OVERLAP=1)"
echo "one launch (r02 plan):     $(W2 NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_OVERLAP=0 NRLDPC_HIP_TB_CLASSES=0 NRLDPC_HIP_TB_MULTI=0)"
for OV in 1 0; do
  cd /tmp && NRLDPC_HIP_TB_OVERLAP=$OV timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tr$OV" -- python "$GRAFT_REPO_ROOT/tools/small_tbs.py" --worker 64 4 0.35 > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"; echo "== kernel timeline of the last call, overlap=$OV"; python - <<PY
import csv, glob
ev=[]
for f in glob.glob("$O/tr$OV/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Workgroup_Size_X", r.get("Workgroup_Size","?")), r.get("Grid_Size_X", r.get("Grid_Size","?"))))
ev.sort()
idx=[i for i,e in enumerate(ev) if "tb_rx" in e[2] or "ldpc_dec" in e[2]]
last=idx[-1]; start=last
while start>0 and ev[start][0]-ev[start-1][1] < 100_000 and ("tb_rx" in ev[start-1][2] or "ldpc_dec" in ev[start-1][2]): start-=1
t0=ev[start][0]
for s,e,n,wg,grid in ev[start:last+1]:
    print(f"{(s-t0)/1e3:8.1f} -> {(e-t0)/1e3:8.1f} us  wg {wg:>5} grid {grid:>8}  {n}")
PY
  rm -rf $O/tr$OV
done
echo "== slot + tests"; timeout 300 python tools/slot_chain.py 50; timeout 900 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py -m gpu -q 2>&1 | tail -3
