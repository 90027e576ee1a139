#!/bin/bash
# round 6, call C: is the persistent form's +20 us the kernel or the launch?  rocprofv3 kernel durations of the same A/B child
mkdir -p gpurun_out/r06
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06
for v in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && NRLDPC_HIP_PERSIST=$v AB_ONLY_R13=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/tr_p$v" -- python "$GRAFT_REPO_ROOT/tools/ab_fast.py" --child > "$GRAFT_REPO_ROOT/$O/ab_child_p$v.json" 2>/dev/null )
  echo "== NRLDPC_HIP_PERSIST=$v"; python tools/prof_kernels.py $O/tr_p$v | grep -i "dec_fast" ; cat $O/ab_child_p$v.json | tail -1
  rm -rf $O/tr_p$v
done 2>&1 | tee $O/persist_kernel_vs_launch.txt
