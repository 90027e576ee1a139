#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ak; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== direction (c): the two lanes of a degree-19 row item 32 apart (half = lane >> 5) instead of neighbours: each half-wave reads ONE message row"
timeout 1500 python tools/ab_fast.py base: split32:NRLDPC_HIP_LIB=$A/libldpc_hip_split32.so base2: split322:NRLDPC_HIP_LIB=$A/libldpc_hip_split32.so
echo "== LDS bank conflicts of the headline launch (rocprofv3 --pmc, own runs)"
for L in base split32; do
  LIBENV=""; [ $L = split32 ] && LIBENV="$A/libldpc_hip_split32.so"
  cd /tmp && NRLDPC_HIP_LIB=$LIBENV timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$L" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-operating-point --no-strong --no-chain > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"; python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_$L/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void ldpc_dec_fast_kernel<false, false>"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$L:", {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
  rm -rf $O/pmc_$L
done
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab30_pair_lanes_32_apart.txt
