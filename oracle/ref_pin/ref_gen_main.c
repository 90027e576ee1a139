/*
 * ref_gen_main.c -- TEST INFRASTRUCTURE ONLY.  Driver for those of the reference's CODE GENERATORS that compile here from
 * their own sources with the reference's own headers alone (the others include PHY/sse_intrin.h -> the un-vendored SIMDE):
 *   nrLDPC_tools/generator_cnProc/cnProc_gen_BG1_avx2.c     nrLDPC_cnProc_BG1_generator_AVX2    -> cnProc/nrLDPC_cnProc_BG1_R{13,23,89}_AVX2.h
 *   nrLDPC_tools/generator_bnProc/bnProcPc_gen_BG1_avx2.c   nrLDPC_bnProcPc_BG1_generator_AVX2  -> bnProcPc/nrLDPC_bnProcPc_BG1_R*_AVX2.h
 *   nrLDPC_tools/generator_bnProc/bnProc_gen_BG2_avx2.c     nrLDPC_bnProc_BG2_generator_AVX2    -> bnProc/nrLDPC_bnProc_BG2_R{15,13,23}_AVX2.h
 *   nrLDPC_tools/generator_cnProc_avx512/cnProc_gen_BG{1,2}_avx512.c   nrLDPC_cnProc_BG{1,2}_generator_AVX512
 *                                                                       -> cnProc_avx512/nrLDPC_cnProc_BG{1,2}_R*_AVX512.h
 *     (the AVX-512 build is the one SURVEY 8c names canonical: both base graphs' check-node functions)
 * i.e. the text of node functions the SHIPPED decoder calls (nrLDPC_decoder.c:256-830 under UNROLL_*), written by
 * reference-compiled code.  The headers themselves cannot be compiled here (SIMDE intrinsics); tests/test_ref_pin.py reads
 * them: which buffer words feed which, through which operations -- the wiring of [D2], [D3] / [F5] and [D5].
 * The reference's own main.c files call every generator of a directory; this one calls the ones that link.
 */
#include <stdio.h>
void nrLDPC_cnProc_BG1_generator_AVX2(const char *, int);
void nrLDPC_bnProcPc_BG1_generator_AVX2(const char *, int);
void nrLDPC_bnProc_BG2_generator_AVX2(const char *, int);
void nrLDPC_cnProc_BG1_generator_AVX512(const char *, int);
void nrLDPC_cnProc_BG2_generator_AVX512(const char *, int);

int main(int argc, char **argv)
{
  if (argc != 2) {
    fprintf(stderr, "usage: %s <output dir holding cnProc/ bnProc/ bnProcPc/ cnProc_avx512/>\n", argv[0]);
    return 1;
  }
  for (int r = 0; r < 3; r++) {
    nrLDPC_cnProc_BG1_generator_AVX2(argv[1], r);
    nrLDPC_bnProcPc_BG1_generator_AVX2(argv[1], r);
    nrLDPC_bnProc_BG2_generator_AVX2(argv[1], r);
    nrLDPC_cnProc_BG1_generator_AVX512(argv[1], r);
    nrLDPC_cnProc_BG2_generator_AVX512(argv[1], r);
  }
  return 0;
}
