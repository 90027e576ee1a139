/*
 * cn_probe.hip -- one kernel per instantiation of the fast decoder's per-item bodies (ldpc_dec_fast_core.h), so that the
 * instruction mix of "one check-node item of degree D" / "one bit-node item of loop bound M" can be read off the
 * disassembly (tools/valu_issue_model.py).  Never launched; compiled with the same flags as the library.
 */
#include <hip/hip_runtime.h>
#include "ldpc_dec_fast_core.h"

struct probe_args { uint8_t *lds_image; const uint8_t *gllr; uint32_t *out; int e0, j, Z, rstride, astride, ext_global; uint32_t colrec, llr_word; };

__device__ __forceinline__ ldpc_fast_lds probe_lds(uint8_t *fsm, const probe_args &a)
{
  ldpc_fast_lds L;
  L.base = fsm; L.r = fsm; L.app = fsm + 122608; L.ext = fsm + 142576;
  L.etbl = reinterpret_cast<const uint32_t *>(fsm + 158704); L.ctbl = reinterpret_cast<const uint32_t *>(fsm + 159968);
  L.rowtbl = nullptr; L.coltbl = nullptr; L.gllr = a.gllr; L.ext_global = a.ext_global;
  return L;
}
#define CN_PROBE(D, EXT, MODE)                                                                       \
  extern "C" __global__ void __launch_bounds__(1024) probe_cn_##D##_##EXT##_##MODE(const probe_args a) \
  {                                                                                                  \
    extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];                                   \
    const ldpc_fast_lds L = probe_lds(fsm, a);                                                       \
    if (MODE == 0) /* the library's dispatch: prefix/suffix body for everything whose registers fit */                        \
      a.out[threadIdx.x] = ldpc_fast_cn_ps<D, EXT != 0>(L, a.e0 + (int)threadIdx.x, a.j + (int)threadIdx.x, a.Z, a.rstride);         \
    else                                                                                             \
      a.out[threadIdx.x] = ldpc_fast_cn<D, EXT != 0, MODE>(L, a.e0 + (int)threadIdx.x, a.j + (int)threadIdx.x, a.Z, a.rstride); \
  }
CN_PROBE(19, 0, 2) CN_PROBE(10, 0, 0) CN_PROBE(8, 0, 0)
/* a degree-19 row item shared by two lanes (ldpc_fast_cn19_pair): one lane's instructions */
extern "C" __global__ void __launch_bounds__(1024) probe_cn_19pair(const probe_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  const ldpc_fast_lds L = probe_lds(fsm, a);
  a.out[threadIdx.x] = ldpc_fast_cn19_pair<false>(L, a.e0 + (int)threadIdx.x, a.j + (int)threadIdx.x, a.Z, a.rstride, (int)threadIdx.x & 1);
}
CN_PROBE(3, 1, 0) CN_PROBE(4, 1, 0) CN_PROBE(5, 1, 0) CN_PROBE(6, 1, 0) CN_PROBE(7, 1, 0) CN_PROBE(8, 1, 0) CN_PROBE(9, 1, 0) CN_PROBE(10, 1, 0)
/* a double task's item pair (ldpc_fast_cn_ps2: two items of a low-degree extension row walked together): both items' instructions */
#define CN2_PROBE(D)                                                                                  \
  extern "C" __global__ void __launch_bounds__(1024) probe_cn2_##D(const probe_args a)                 \
  {                                                                                                   \
    extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];                                    \
    const ldpc_fast_lds L = probe_lds(fsm, a);                                                        \
    uint32_t mb = 0;                                                                                  \
    const uint32_t ma = ldpc_fast_cn_ps2<D, true, false>(L, a.e0 + (int)threadIdx.x, a.j + (int)threadIdx.x, a.e0 + 7 + (int)threadIdx.x, \
                                                          a.j + 64 + (int)threadIdx.x, a.Z, a.rstride, mb);                                \
    a.out[threadIdx.x] = ma ^ (mb << 4);                                                              \
  }
CN2_PROBE(3) CN2_PROBE(4) CN2_PROBE(5)
#define BN_PROBE(M)                                                                        \
  extern "C" __global__ void __launch_bounds__(1024) probe_bn_##M(const probe_args a)       \
  {                                                                                        \
    extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];                         \
    const ldpc_fast_lds L = probe_lds(fsm, a);                                             \
    ldpc_fast_bn(L, a.colrec + threadIdx.x, M, a.j + (int)threadIdx.x, a.Z, a.astride, a.llr_word); \
  }
BN_PROBE(1) BN_PROBE(2) BN_PROBE(3) BN_PROBE(4) BN_PROBE(5) BN_PROBE(6) BN_PROBE(7) BN_PROBE(8) BN_PROBE(9) BN_PROBE(10) BN_PROBE(11) BN_PROBE(12)
BN_PROBE(13) BN_PROBE(14) BN_PROBE(22) BN_PROBE(23) BN_PROBE(28) BN_PROBE(30)
