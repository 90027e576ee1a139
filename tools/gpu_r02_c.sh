#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.txt
timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_chain" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 > "$GRAFT_REPO_ROOT/$O/prof_chain.log" 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_chain -name '*kernel_stats.csv' | head -1); echo "stats: $f"; head -12 "$f" | cut -c1-200; cp "$f" $O/chain_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_chain_$c" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 5 > "$GRAFT_REPO_ROOT/$O/pmc_chain_$c.log" 2>&1
  cd "$GRAFT_REPO_ROOT"; python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_chain_$c/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    if "tb_" in k or "ldpc" in k:
        print("$c", k, "n=%d mean=%.1f KiB max=%.1f" % (len(v), sum(v) / len(v), max(v)))
PY
done | tee $O/chain_pmc.txt
timeout 900 python tools/bench_extra.py > $O/bench_extra.json 2> $O/bench_extra.err; echo "bench_extra rc=$?"; python -c "
import json; d=json.load(open('$O/bench_extra.json'))
for k,v in d.items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
