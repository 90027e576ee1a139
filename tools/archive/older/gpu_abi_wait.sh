#!/bin/bash
# per-segment ABI: calls/s and host CPU-seconds per 10^5 calls for every NRLDPC_HIP_SRV_WAIT mode; ldpctest -l 8448 -s10
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
gcc -O2 -I include tests/abi_threads.c -o /tmp/abi_threads -ldl -lpthread
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 timeout 120 /tmp/abi_threads $L $2 ${3:-400} $4 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%8.0f calls/s  %6.2f us/call/thread  cpu %.3f s per 1e5 calls  failures %d  served %d  launches %d' % (d['calls_per_s'], d['us_per_call_per_thread'], d['cpu_seconds_per_1e5_calls'], d['failures'], d['served'], d['server_launches']))")"; }
{
  echo "# one caller, BG1 Zc=384 R=1/3, 2 iterations (case 1) -- the per-call latency"
  for m in spin yield sleep; do run NRLDPC_HIP_SRV_WAIT=$m 1 3000 1; done
  run NRLDPC_HIP_SERVER=0 1 1000 1
  echo "# mixed codes, 16 / 32 / 64 callers"
  for T in 16 32 64; do for m in spin yield sleep; do run NRLDPC_HIP_SRV_WAIT=$m $T 1500; done; done
  echo "# nproc $(nproc), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
} | tee $O/abi_wait_modes.txt
timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 > $O/ldpctest_hip_8448.txt 2>&1; tail -3 $O/ldpctest_hip_8448.txt
