/*
 * abi_offsets.c -- prints size and field offsets of the plugin ABI structures, either as the reference declares them
 * (-DUSE_REFERENCE, include paths into the reference tree: openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_types.h compiles on
 * its own) or as include/nrLDPC_hip.h restates them.  tests/test_abi.py compiles it both ways and compares the output
 * line by line (development container only: the reference tree does not travel).  Test infrastructure.
 */
#include <stddef.h>
#include <stdio.h>
#ifdef USE_REFERENCE
#include "PHY/CODING/nrLDPC_decoder/nrLDPC_types.h"
#else
#include "nrLDPC_hip.h"
#endif
#define F(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T *)0)->f))
int main(void)
{
  printf("t_nrLDPC_dec_params %zu\n", sizeof(t_nrLDPC_dec_params));
  F(t_nrLDPC_dec_params, BG); F(t_nrLDPC_dec_params, Z); F(t_nrLDPC_dec_params, R); F(t_nrLDPC_dec_params, F);
  F(t_nrLDPC_dec_params, Qm); F(t_nrLDPC_dec_params, rv); F(t_nrLDPC_dec_params, numMaxIter); F(t_nrLDPC_dec_params, E);
  F(t_nrLDPC_dec_params, outMode); F(t_nrLDPC_dec_params, crc_type); F(t_nrLDPC_dec_params, check_crc); F(t_nrLDPC_dec_params, setCombIn);
  printf("time_stats_t %zu\n", sizeof(time_stats_t));
  F(time_stats_t, in); F(time_stats_t, diff); F(time_stats_t, p_time); F(time_stats_t, diff_square); F(time_stats_t, max);
  F(time_stats_t, trials); F(time_stats_t, meas_flag); F(time_stats_t, meas_name); F(time_stats_t, meas_index);
  F(time_stats_t, meas_enabled); F(time_stats_t, tpoolmsg); F(time_stats_t, tstatptr);
  printf("t_nrLDPC_time_stats %zu\n", sizeof(t_nrLDPC_time_stats));
  F(t_nrLDPC_time_stats, llr2llrProcBuf); F(t_nrLDPC_time_stats, cnProc); F(t_nrLDPC_time_stats, bnProc);
  F(t_nrLDPC_time_stats, llr2bit); F(t_nrLDPC_time_stats, total);
  printf("enum %d %d %d\n", (int)nrLDPC_outMode_BIT, (int)nrLDPC_outMode_BITINT8, (int)nrLDPC_outMode_LLRINT8);
  return 0;
}
