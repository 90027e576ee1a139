#!/bin/bash
# Round evidence in one GPU call (tools/gpu_profiles.sh [round tag, default r03]): bench line, rocprofv3 kernel statistics, PMC passes (own runs, --kernel-trace only),
# chain kernels, per-segment ABI, host path, link / HBM / BAR micro-benchmarks.  Everything lands in gpurun_out/<tag>/final.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r03}/final; mkdir -p $O; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
rocminfo | grep -E 'Marketing Name|Compute Unit|gfx' | head -6 > $O/rocminfo.txt 2>&1; nproc >> $O/rocminfo.txt; cat /sys/fs/cgroup/cpu.max >> $O/rocminfo.txt 2>&1
echo "== bench"; timeout 900 python bench.py --steps 50 > $O/fast_bench.json 2> $O/fast_bench.err; echo rc=$?
echo "== rocprof stats"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_bench" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --no-cpu-baseline --no-operating-point --no-strong > "$GRAFT_REPO_ROOT/$O/fast_bench_under_rocprof.json" 2> /dev/null
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/fast_kernel_stats.csv; head -4 $O/fast_kernel_stats.csv | cut -c1-160
echo "== pmc"
pmc() { cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$1" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-operating-point --no-strong > /dev/null 2>&1; cd "$GRAFT_REPO_ROOT"; }
pmc sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES"
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
pmc grbm "GRBM_GUI_ACTIVE"
python tools/pmc_summary.py $O > $O/fast_kernel_pmc_summary.txt; tail -30 $O/fast_kernel_pmc_summary.txt | cut -c1-140
echo "== chain"; timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_chain" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_chain -name '*kernel_stats.csv' | head -1); cp "$f" $O/chain_kernel_stats.csv; head -8 $O/chain_kernel_stats.csv | cut -c1-140
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_chain_$c" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 5 > /dev/null 2>&1
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
echo "== chain: retransmission (HARQ round 1: only the received positions of the soft buffers are rewritten)"
timeout 300 python tools/slot_chain.py 20 0.18 retx | tee $O/slot_chain_retx.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_chain_retx" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 5 0.18 retx > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; python - <<PY | tee -a $O/chain_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_chain_retx/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    if "dematch" in k:
        print("WRITE_SIZE (rounds 0 and 1 alternate)", k, "n=%d mean=%.1f KiB min=%.1f max=%.1f" % (len(v), sum(v) / len(v), min(v), max(v)))
PY
echo "== mixed batch of small transport blocks"; timeout 300 python tools/small_tbs.py 64 50 0.35 2>&1 | tee $O/small_tbs.txt; timeout 300 python tools/small_tbs.py 256 30 0.35 2>&1 | tee -a $O/small_tbs.txt
echo "== offload slot"; timeout 120 python tools/offload_latency.py 2>&1 | tee $O/offload_latency.txt
echo "== sims"; timeout 300 python tests/ulschsim_hip.py -R 106 -m 9 -s 13 -n 100 2>&1 | tail -4 | tee $O/ulschsim.txt; timeout 300 python tests/dlschsim_hip.py -R 106 -m 9 -s 13 -n 100 2>&1 | tail -4 | tee $O/dlschsim.txt
echo "== bench --gpus 1 under torch.distributed (RCCL path)"; BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; tail -c 600 $O/bench_dist1.json
echo "== per-call breakdown"; timeout 120 python tools/srv_breakdown.py 3000 2>&1 | tail -1 | tee $O/srv_breakdown.txt; timeout 120 python tools/srv_breakdown.py 3000 1 384 13 -12 2>&1 | tail -1 | tee -a $O/srv_breakdown.txt
echo "== abi"
gcc -O2 -I include tests/abi_threads.c -o tests/abi_threads.bin -ldl -lpthread   # (binaries are not tracked)
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 timeout 60 ./tests/abi_threads.bin $L $2 ${3:-400} $4 2>&1 | tail -1 | cut -c1-330)"; }
{ run X=1 1 3000 1; run X=1 1 2000 0; run NRLDPC_HIP_SRV_BAR=0 1 3000 1; run NRLDPC_HIP_SERVER=0 1 1000 1
  for T in 1 4 16 32 64; do run X=1 $T 1000; done
  for W in yield sleep; do for T in 1 32; do run NRLDPC_HIP_SRV_WAIT=$W $T 1000; done; done
  for T in 1 16 32; do run NRLDPC_HIP_SERVER=0 $T 300; done; } | tee $O/abi_threads.txt
timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 > $O/ldpctest_hip_8448.txt 2>&1; tail -3 $O/ldpctest_hip_8448.txt
timeout 120 python tools/enc_call_latency.py 2000 2>&1 | grep segment | sed "s/^/server: /" > $O/enc_call_latency.txt; NRLDPC_HIP_ENC_SERVER=0 timeout 120 python tools/enc_call_latency.py 1000 2>&1 | grep segment | sed "s/^/launch per call: /" >> $O/enc_call_latency.txt; cat $O/enc_call_latency.txt
timeout 60 python tools/first_use_latency.py 2>&1 | grep call > $O/first_use_latency.txt
echo "== host path + ubench"
timeout 120 python tools/host_path_sweep.py 2>&1 | grep chunk | tee $O/host_path.txt
NRLDPC_HIP_HOST_PULL=0 timeout 120 python tools/host_path_sweep.py 2>&1 | grep chunk | sed "s/^/copy engine: /" | tee -a $O/host_path.txt
timeout 120 tools/ubench/h2d_link.bin > $O/h2d_link.txt 2>&1; timeout 120 tools/ubench/hbm_rw.bin > $O/hbm_rw.txt 2>&1; timeout 120 tools/ubench/bar_write.bin > $O/bar_write.txt 2>&1
echo "== small lifting sizes"
C4="1,64,13 1,32,13 1,16,13 1,8,13 1,48,13 1,24,13 2,64,15 2,32,15 2,16,15 2,8,15"
CX="1,30,13 1,26,13 1,22,13 1,15,13 1,7,13 1,2,13 2,30,15 2,15,15 2,7,15 2,3,15"
{ echo "# Zc % 4 == 0, one block per workgroup (kernel 3), 32768 blocks"; SWEEP_KERNEL=3 timeout 300 python tools/sweep_codes.py 32768 $C4 2>&1 | grep -v amdgpu.ids
  echo "# Zc % 4 == 0, several blocks per workgroup (kernel 0 = automatic)"; SWEEP_KERNEL=0 timeout 300 python tools/sweep_codes.py 32768 $C4 2>&1 | grep -v amdgpu.ids
  echo "# other Zc, generic kernel (kernel 1)"; SWEEP_KERNEL=1 timeout 300 python tools/sweep_codes.py 32768 $CX 2>&1 | grep -v amdgpu.ids
  echo "# other Zc, four blocks interleaved byte-wise, several groups per workgroup (kernel 0 = automatic)"; SWEEP_KERNEL=0 timeout 300 python tools/sweep_codes.py 32768 $CX 2>&1 | grep -v amdgpu.ids
} > $O/small_lifting_sizes.txt; tail -12 $O/small_lifting_sizes.txt
echo "== bench_extra"; timeout 900 python tools/bench_extra.py > $O/bench_extra.json 2> $O/bench_extra.err; echo rc=$?
rm -rf $O/prof_bench/*/*trace.csv $O/prof_chain/*/*trace.csv   # keep the merge small
