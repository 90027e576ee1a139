#!/bin/bash
# build a variant of the library with extra -D flags for the decoder kernels: tools/build_variant.sh <name> [-DFLAG ...]
# -> openairinterface5g_amd/lib/libldpc_hip_<name>.so (select with NRLDPC_HIP_LIB; needs the normal build first)
set -e
name=$1; shift
cd "$(dirname "$0")/../openairinterface5g_amd/csrc"
B=build_$name; mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c ldpc_decoder_fast.hip -o $B/fast.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libldpc_hip_$name.so $B/fast.o build/ldpc_decoder.o build/ldpc_server.o build/ldpc_encoder.o build/ldpc_api.o build/ldpc_graph.o build/tb_chain.o build/nr_coding_host.o -lpthread
rm -rf $B
