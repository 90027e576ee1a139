/*
 * ldpc_t2_shim.c -- libldpc_hip_t2.so: the plugin ABI of the reference's OFFLOAD slot on top of libldpc_hip.so.
 *
 * A gNB started with --ldpc-offload-enable loads a second coding library with the version suffix "_t2" into
 * `ldpc_interface_offload` (openair1/PHY/INIT/nr_init.c:138-139; nrLDPC_load.c:45-75 resolves the same four names as for
 * the CPU library) and then calls it per segment with the semantics of nrLDPC_decoder/nrLDPC_decoder_offload.c:1036-1140:
 * rate (de)matching, (de)interleaving and HARQ combining happen inside the library.  Installed as libldpc_t2.so (or
 * selected with --loader.ldpc_t2.shlibversion) this file gives that slot the GPU: the work is done by
 * nrLDPC_hip_offload_decoder / _encoder of libldpc_hip.so (csrc/tb_offload.inc.cpp), found through DT_NEEDED + $ORIGIN.
 * Same names, different meaning than libldpc_hip.so's own LDPCdecoder/LDPCencoder -- hence a separate shared object.
 */
#include "../../include/nrLDPC_hip.h"

int32_t LDPCinit(void) { return nrLDPC_hip_offload_init(); }
int32_t LDPCshutdown(void) { return 0; } /* offload.c:1027-1034 tears the DPDK device down; nothing to tear down here */
int32_t LDPCdecoder(t_nrLDPC_dec_params *p_decParams, uint8_t harq_pid, uint8_t ulsch_id, uint8_t C, int8_t *p_llr, int8_t *p_out,
                    t_nrLDPC_time_stats *p_profiler, decode_abort_t *ab)
{
  return nrLDPC_hip_offload_decoder(p_decParams, harq_pid, ulsch_id, C, p_llr, p_out, p_profiler, ab);
}
int32_t LDPCencoder(uint8_t **input, uint8_t **output, encoder_implemparams_t *impp)
{
  return nrLDPC_hip_offload_encoder(input, output, impp);
}
/* the loader's version hook of this slot: the same answer -- ABI string, NRLDPC_HIP_REQUIRE_BUILD gate -- as for
 * libldpc_hip.so (the gate did not apply to the offload slot) */
int32_t ldpc_checkbuildver(char *mainexec_buildversion, char **shlib_buildversion)
{
  return nrLDPC_hip_checkbuildver(mainexec_buildversion, shlib_buildversion);
}
