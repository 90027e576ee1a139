#!/bin/bash
# diagnostic build of the encoder kernel (-DENC_PHASE_STOP): tools/ab/libldpc_hip_encstop.so (needs `make` first)
set -e
cd "$(dirname "$0")/../openairinterface5g_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DENC_PHASE_STOP "$@" -c ldpc_encoder.hip -o /tmp/enc_stop.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -Wl,--version-script=exports.map -o ../../tools/ab/libldpc_hip_encstop.so \
  build/ldpc_decoder.o build/ldpc_decoder_fast.o build/ldpc_server.o /tmp/enc_stop.o build/ldpc_api.o build/ldpc_graph.o build/tb_chain.o build/tb_rx_fused.o build/nr_coding_host.o -lpthread
