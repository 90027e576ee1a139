#!/bin/bash
# Diagnostic: build the fast decoder with -DLDPC_TIMING into lib/libldpc_hip_timing.so (run here, after the normal
# build), then on the GPU box:
#   NRLDPC_HIP_LIB=$PWD/openairinterface5g_amd/lib/libldpc_hip_timing.so python tools/task_timing.py
set -e
cd "$(dirname "$0")/../openairinterface5g_amd/csrc"
mkdir -p build_t
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLDPC_TIMING -c ldpc_decoder_fast.hip -o build_t/fast.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libldpc_hip_timing.so build_t/fast.o build/ldpc_decoder.o build/ldpc_server.o build/ldpc_encoder.o build/ldpc_api.o build/ldpc_graph.o build/tb_chain.o build/nr_coding_host.o -lpthread
rm -rf build_t
