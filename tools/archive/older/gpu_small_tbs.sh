#!/bin/bash
# small transport blocks, several per workgroup vs one per workgroup: timings at three noise levels + kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
C=${1:-64}
for s in 0.2 0.35 0.45; do
  timeout 200 python tools/small_tbs.py $C 20 $s 2>&1 | tee -a $O/small_tbs.txt
done
for mlt in 1 0; do
  (cd /tmp && NRLDPC_HIP_TB_MULTI=$mlt NRLDPC_HIP_TB_CLASSES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/st$mlt" -- python "$GRAFT_REPO_ROOT/tools/small_tbs.py" --worker $C 20 0.35 > /dev/null 2>&1)
  f=$(find $O/st$mlt -name '*kernel_stats.csv' | head -1)
  echo "== NRLDPC_HIP_TB_MULTI=$mlt" >> $O/small_tbs_kernels.txt
  head -12 "$f" | cut -c1-220 >> $O/small_tbs_kernels.txt
  rm -rf $O/st$mlt
done
cat $O/small_tbs_kernels.txt
