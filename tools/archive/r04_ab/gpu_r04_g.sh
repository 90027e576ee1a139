#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/g; mkdir -p $O; export TMPDIR=/tmp
for V in "auto NRLDPC_HIP_TB_HOST_CHUNKS=4" "pull NRLDPC_HIP_TB_PULL=2"; do set -- $V
  cd /tmp && env $2 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tl_$1" -- python "$GRAFT_REPO_ROOT/tools/host_timeline.py" 6 2>/dev/null | tail -3
  cd "$GRAFT_REPO_ROOT"; python tools/host_timeline_parse.py $O/tl_$1 | tee $O/host_timeline_$1.txt | head -60; rm -rf $O/tl_$1
done
