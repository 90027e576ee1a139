#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/j; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py -m gpu -q --maxfail=5 2>&1 | tail -25 > $O/pytest.log; tail -5 $O/pytest.log
W() { env "$@" timeout 200 python tools/small_tbs.py --worker ${C:-64} 30 0.35 2>/dev/null | tail -1; }
for C in 64 256; do export C
echo "== small TBS x$C"
echo "r03 plan (unfused, class launches): $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_FILL=0)"
echo "unfused, filled rounds:             $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_FILL=1)"
echo "fused, class launches:              $(W NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_FILL=0)"
echo "fused, filled rounds (default):     $(W NRLDPC_HIP_TB_FUSED=1 NRLDPC_HIP_TB_FILL=1)"
echo "one launch (r02 plan):              $(W NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_TB_CLASSES=0 NRLDPC_HIP_TB_MULTI=0)"
done 2>&1 | tee $O/small_tbs.txt
echo "== bench_extra (config3 part)"; timeout 600 python - <<'PY' 2>/dev/null | tee $O/config3.json
import runpy, sys, json, io, contextlib
src = open("tools/bench_extra.py").read()
cut = src.index("# ---- the decoder's rate modes")
exec(compile(src[:cut] + "\nprint(json.dumps(res, indent=1))\n", "bench_extra_head", "exec"))
PY
