#!/bin/bash
cd "$GRAFT_REPO_ROOT"; A=$GRAFT_REPO_ROOT/tools/ab
lscpu | grep -E "Model name|MHz" | head -3; cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null
for i in 1 2 3; do
  echo "previous build: $(NRLDPC_HIP_LIB=$A/libldpc_hip_head.so timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 2>&1 | grep -i 'decoding time mean')"
  echo "this build:     $(timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 2>&1 | grep -i 'decoding time mean')"
done
