#!/bin/bash
# build ablated variants of the fast kernel into separate libraries (timing only -- results are wrong by design)
set -e
cd "$(dirname "$0")/../openairinterface5g_amd/csrc"
for V in BN CN; do
  rm -rf build_ab && mkdir -p build_ab
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLDPC_ABLATE_$V -c ldpc_decoder_fast.hip -o build_ab/ldpc_decoder_fast.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libldpc_hip_no$V.so build_ab/ldpc_decoder_fast.o build/ldpc_decoder.o build/ldpc_encoder.o build/ldpc_api.o build/ldpc_graph.o build/tb_chain.o build/nr_coding_host.o -lpthread
done
rm -rf build_ab
