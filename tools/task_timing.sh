#!/bin/bash
# Diagnostic: build the library with -DLDPC_TIMING into tools/ab/libldpc_hip_timing.so (run here, after the normal
# build), then on the GPU box:
#   NRLDPC_HIP_LIB=$PWD/tools/ab/libldpc_hip_timing.so python tools/task_timing.py
exec "$(dirname "$0")/build_variant.sh" timing -DLDPC_TIMING
