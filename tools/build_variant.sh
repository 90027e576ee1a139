#!/bin/bash
# build a variant of the library with extra -D flags for the translation units that hold the decoder kernels:
#   tools/build_variant.sh <name> [-DFLAG ...]   -> tools/ab/libldpc_hip_<name>.so (select with NRLDPC_HIP_LIB; needs `make` first)
set -e
name=$1; shift
cd "$(dirname "$0")/../openairinterface5g_amd/csrc"
B=/tmp/build_variant_$name; rm -rf $B; mkdir -p $B ../../tools/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
for tu in ldpc_decoder_fast ldpc_server tb_rx_fused; do
  /opt/rocm/bin/hipcc $F "$@" -c $tu.hip -o $B/$tu.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -Wl,--version-script=exports.map \
  -o ../../tools/ab/libldpc_hip_$name.so build/ldpc_decoder.o $B/ldpc_decoder_fast.o $B/ldpc_server.o build/ldpc_encoder.o build/ldpc_api.o \
  build/ldpc_graph.o build/tb_chain.o $B/tb_rx_fused.o build/nr_coding_host.o -lpthread
rm -rf $B
