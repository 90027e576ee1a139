#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/confirm; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 > $O/run_$i.txt; rc=${PIPESTATUS[0]}
  echo "run $i rc=$rc: $(grep -E "passed|failed" $O/run_$i.txt | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then cat $O/run_$i.txt | cut -c1-300; break; fi
done
