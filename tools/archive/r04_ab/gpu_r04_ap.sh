#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ap; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== compiler options for the decoder translation units (head = -O3, committed): -O2; -mllvm -enable-post-misched=0 (nopost); -amdgpu-schedule-metric-bias=30 (bias); -misched-cluster=false (nocluster); -amdgpu-disable-unclustered-high-rp-reschedule (nohrp); -amdgpu-use-amdgpu-trackers (trackers); -amdgpu-prealloc-sgpr-spill-vgprs (prealloc); -amdgpu-disable-clustered-low-occupancy-reschedule (noclo)"
V=""; for n in o2 nopost bias nocluster nohrp trackers prealloc noclo; do V="$V $n:NRLDPC_HIP_LIB=$A/libldpc_hip_$n.so"; done
timeout 2000 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so $V head2:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab33_compiler_options.txt
