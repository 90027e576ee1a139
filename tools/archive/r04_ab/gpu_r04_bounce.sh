#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/bounce; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > $O/run_$i.txt; rc=${PIPESTATUS[0]}
  echo "run $i rc=$rc: $(grep -E "passed|failed" $O/run_$i.txt | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ]; then cat $O/run_$i.txt | cut -c1-300; fi
done
timeout 600 python tools/slot_chain_host.py 6 2>/dev/null > $O/slot_chain_host.json; python - <<PY
import json
d = json.load(open("$O/slot_chain_host.json"))
for k, v in d.items():
    if isinstance(v, dict) and "round0_ms" in v:
        print(k, round(v["round0_ms"], 3), round(v["round1_ms"], 3), v.get("all_ack"))
PY
