/*
 * tb_chain.hip -- transport-block chain kernels for gfx950 (see tb_chain.h).  All of them are byte/int16
 * gather-scatter work bound by HBM/L2 traffic; one workgroup per transport block or per code block.
 *
 * Reference functions replaced (openair1/PHY/...):
 *   TB CRC attach                       NR_TRANSPORT/nr_dlsch_coding.c:300-331 (crc24a / crc16, CODING/crc_byte.c)
 *   nr_segmentation (data part)         CODING/nr_segmentation.c:147-175
 *   nr_rate_matching_ldpc + nr_interleaving_ldpc   CODING/nr_rate_matching.c:424-505, :36-303 (ldpc8blocks, nr_dlsch_coding.c:177-245)
 *   nr_deinterleaving_ldpc + nr_rate_matching_ldpc_rx + int8 pack
 *                                       CODING/nr_rate_matching.c:310-388, :507-603; NR_TRANSPORT/nr_ulsch_decoding.c:153-210
 *   nr_postDecode (reassembly, TB CRC)  SCHED_NR/phy_procedures_nr_gNB.c:271-300
 */
#include <hip/hip_runtime.h>
#include "tb_chain.h"
#include "ldpc_kernels.h"
#include "ldpc_enc_packed_core.h"
#include "ldpc_enc_packed32.h"
/* the separate de-matching launch holds 8 workgroups per CU (64 VGPRs): their occupancy hides the LLRs' latency, and symbols
 * requested ahead of the clearing would only spill (the fused segment kernel, 126 VGPRs and two workgroups per CU, asks
 * for four: tb_rx_fused.hip) */
#define TB_RX_U 0
/* ... and a thread of a first transmission keeps two stores in flight, not all of them: with 32 waves per CU storing
 * nothing but zeros and saturated sums, an unbounded burst runs 3 % slower than a bounded one (HBM write queues;
 * profiles/r04/ab_dematch_store_window.txt: 25.3 us unbounded, 24.5 us with a window of 2 = the r03 launch). */
#ifndef TB_RX_STORE_WINDOW
#define TB_RX_STORE_WINDOW 2
#endif
#include "tb_rx_core.h"

#define TB_THREADS 256

/* XOR-reduce x over the workgroup; every thread gets the result.  red = 2 dwords of LDS. */
__device__ __forceinline__ uint32_t tb_block_xor(uint32_t x, uint32_t *red)
{
  for (int off = 32; off; off >>= 1)
    x ^= __shfl_xor(x, off);
  if (threadIdx.x == 0)
    red[0] = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && x)
    atomicXor(&red[0], x);
  __syncthreads();
  const uint32_t r = red[0];
  __syncthreads();
  return r;
}

/* Left-aligned CRC register of the nbits-bit string at `data` (MSB first), as crc24a()/crc24b()/crc16() return it
 * (crc_byte.c:148-260): linear in the bits, bit i contributes pow[nbits-1-i] = x^(nbits-1-i) * x^deg mod g. */
__device__ __forceinline__ uint32_t tb_block_crc(const uint8_t *__restrict__ data, uint32_t nbits, const uint32_t *__restrict__ pow,
                                                 uint32_t *red)
{
  uint32_t x = 0;
  const uint32_t nbytes = nbits >> 3;
  for (uint32_t q = threadIdx.x; q < nbytes; q += blockDim.x) {
    uint32_t v = data[q];
    const uint32_t top = nbits - 1 - 8 * q; /* exponent of the byte's MSB */
    while (v) {
      const int b = 31 - __clz(v); /* bit b (0 = LSB) sits 7-b positions after the MSB */
      x ^= pow[top - (7 - b)];
      v &= ~(1u << b);
    }
  }
  return tb_block_xor(x, red);
}

/* Byte table of a CRC from its power table: tab[v] = register after the single byte v = XOR of pow[m] over the set
 * bits m of v (bit m is followed by m more bits).  256 entries of LDS; the caller synchronises. */
__device__ __forceinline__ void tb_build_crc_tab(const uint32_t *__restrict__ pow, uint32_t *tab)
{
  uint32_t p[8]; /* eight independent loads in flight: one memory latency for the whole table */
#pragma unroll
  for (int m = 0; m < 8; m++)
    p[m] = pow[m];
  for (uint32_t v = threadIdx.x; v < 256; v += blockDim.x) {
    uint32_t x = 0;
#pragma unroll
    for (int m = 0; m < 8; m++)
      x ^= p[m] & (0u - ((v >> m) & 1u));
    tab[v] = x;
  }
}

/* Partial left-aligned CRC register of bytes [first, first+count) of an nbits-bit string (same linear form as
 * tb_block_crc).  Every thread runs the byte-table recurrence of crc_byte.c:148-182 over its own SPAN-byte piece
 * (reg = (reg << 8) ^ tab[(reg >> 24) ^ byte], LDS look-ups), then moves the piece's register R to the end of the
 * string: R(x) * x^n_after mod g = XOR over the set bits b of R of pow[b - 32 + n_after] (or the bit itself, shifted,
 * while it still fits under the generator's degree).  Valid in lane 0 of every wave. */
/* R(x) * Q(x) mod g for left-aligned registers of a degree-DEG generator (poly = x^DEG mod g, left aligned): one Horner
 * step per coefficient of R -- the power Q = x^n mod g is ONE table look-up, where summing pow[] over the set bits of R
 * was a dozen dependent global loads per thread (the TB CRC kernels spent most of their 13-17 us there). */
template <int DEG> __device__ __forceinline__ uint32_t tb_crc_mulmod(uint32_t R, uint32_t Q, uint32_t poly)
{
  uint32_t x = 0;
#pragma unroll
  for (int k = 31; k >= 32 - DEG; k--) {
    x = (x << 1) ^ ((uint32_t)((int32_t)x >> 31) & poly);
    x ^= (0u - ((R >> k) & 1u)) & Q;
  }
  return x;
}
template <int SPAN, int DEG>
__device__ __forceinline__ uint32_t tb_partial_crc(const uint8_t *__restrict__ data, uint32_t nbits, uint32_t first, uint32_t count,
                                                   const uint32_t *__restrict__ pow, const uint32_t *tab)
{
  uint32_t x = 0;
  const uint32_t end = first + count;
  for (uint32_t q0 = first + threadIdx.x * SPAN; q0 < end; q0 += blockDim.x * SPAN) {
    const uint32_t n = end - q0 < (uint32_t)SPAN ? end - q0 : (uint32_t)SPAN;
    uint32_t reg = 0;
    if (n == (uint32_t)SPAN && ((reinterpret_cast<uintptr_t>(data + q0) & 3) == 0)) {
#pragma unroll
      for (int w = 0; w < SPAN / 4; w++) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(data + q0)[w];
        reg = (reg << 8) ^ tab[(reg >> 24) ^ (v & 0xffu)];
        reg = (reg << 8) ^ tab[(reg >> 24) ^ ((v >> 8) & 0xffu)];
        reg = (reg << 8) ^ tab[(reg >> 24) ^ ((v >> 16) & 0xffu)];
        reg = (reg << 8) ^ tab[(reg >> 24) ^ (v >> 24)];
      }
    } else {
      for (uint32_t i = 0; i < n; i++)
        reg = (reg << 8) ^ tab[(reg >> 24) ^ data[q0 + i]];
    }
    const uint32_t n_after = nbits - 8 * (q0 + n);
    if (n_after == 0)
      x ^= reg;
    else /* x^n_after mod g, left aligned: pow[j] = x^(j + DEG) mod g */
      x ^= tb_crc_mulmod<DEG>(reg, n_after >= (uint32_t)DEG ? pow[n_after - DEG] : 1u << (32 - DEG + n_after), pow[0]);
  }
  for (int off = 32; off; off >>= 1)
    x ^= __shfl_xor(x, off);
  return x;
}

/* ---- TX 1: b = payload || CRC24A / CRC16 -- chunk-parallel: copy + partial CRC, then the CRC bytes ----------- */
__global__ void __launch_bounds__(TB_THREADS) tb_tx_crc_partial_kernel(const tb_tx_tb_job *jobs, const tb_crc_chunk_job *chunks,
                                                                       const uint8_t *payload, uint8_t *scratch, uint32_t *acc,
                                                                       const uint32_t *pow24a, const uint32_t *pow16, int copy)
{
  __shared__ uint32_t tab[256];
  tb_crc_chunk_job ch = chunks[blockIdx.x];
  const bool small = (ch.first_byte >> 31) != 0;
  ch.first_byte &= 0x7fffffffu;
  const uint32_t chunk = small ? TB_CRC_CHUNK_SMALL : TB_CRC_CHUNK;
  const tb_tx_tb_job j = jobs[ch.tb];
  const uint32_t *pow = j.crc_type == 0 ? pow24a : pow16;
  tb_build_crc_tab(pow, tab);
  const uint8_t *a = payload + j.payload_off;
  uint8_t *b = scratch + j.b_off;
  const uint32_t nbytes = j.A >> 3;
  const uint32_t count = ch.first_byte + chunk <= nbytes ? chunk : nbytes - ch.first_byte;
  /* copy = 0 (the fused segment kernel follows: it takes its bytes from the payload itself): no b, nothing but the CRC */
  if (!copy) {
  } else if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) == 0) { /* (chunks start at multiples of 16) */
    const uint4 *a16 = reinterpret_cast<const uint4 *>(a + ch.first_byte);
    uint4 *b16 = reinterpret_cast<uint4 *>(b + ch.first_byte);
    for (uint32_t q = threadIdx.x; q < (count >> 4); q += blockDim.x)
      b16[q] = a16[q];
    for (uint32_t q = ch.first_byte + (count & ~15u) + threadIdx.x; q < ch.first_byte + count; q += blockDim.x)
      b[q] = a[q];
  } else {
    for (uint32_t q = ch.first_byte + threadIdx.x; q < ch.first_byte + count; q += blockDim.x)
      b[q] = a[q];
  }
  __syncthreads();
  uint32_t x;
  if (small)
    x = j.crc_type == 0 ? tb_partial_crc<8, 24>(a, j.A, ch.first_byte, count, pow, tab)
                        : tb_partial_crc<8, 16>(a, j.A, ch.first_byte, count, pow, tab);
  else
    x = j.crc_type == 0 ? tb_partial_crc<32, 24>(a, j.A, ch.first_byte, count, pow, tab)
                        : tb_partial_crc<32, 16>(a, j.A, ch.first_byte, count, pow, tab);
  /* ONE atomic per workgroup: atomics on one address are served one after the other (~0.35 us each on this stack: the launch
   * took 8.7 us with 16 per transport block and 21 us with 56, profiles/r05/README.md), so the waves meet in LDS first */
  __shared__ uint32_t wsum[TB_THREADS / 64];
  if ((threadIdx.x & 63) == 0)
    wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (uint32_t w = 0; w < blockDim.x >> 6; w++)
      t ^= wsum[w];
    if (!copy)
      acc[blockIdx.x] = t; /* the fused segment kernel adds the chunks of a block up itself: no atomic at all */
    else if (t)
      atomicXor(&acc[ch.tb], t);
  }
}
__global__ void __launch_bounds__(TB_THREADS) tb_tx_crc_final_kernel(const tb_tx_tb_job *jobs, uint32_t n_tb, uint8_t *scratch,
                                                                     uint32_t *acc)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tb)
    return;
  const tb_tx_tb_job j = jobs[i];
  uint8_t *b = scratch + j.b_off + (j.A >> 3);
  const uint32_t crc = acc[i];
  acc[i] = 0; /* the accumulators are zero when a plan is uploaded and are left zero by every call that uses them */
  b[0] = (uint8_t)(crc >> 24);
  b[1] = (uint8_t)(crc >> 16);
  if (j.crc_type == 0)
    b[2] = (uint8_t)(crc >> 8);
}

/* ---- TX 2: code-block segmentation: c_r = b[r*(K'-L) ..] || CRC24B (C > 1) || zero fillers ---------------- */
__global__ void __launch_bounds__(TB_THREADS) tb_tx_segment_kernel(const tb_tx_seg_job *jobs, uint8_t *scratch,
                                                                   const uint32_t *pow24b)
{
  __shared__ uint32_t red[2];
  const tb_tx_seg_job j = jobs[blockIdx.x];
  const uint32_t segbytes = (j.Kprime - j.L) >> 3, kbytes = (j.K + 7) >> 3;
  const uint8_t *src = scratch + j.b_off + (size_t)j.r * segbytes;
  uint8_t *c = scratch + j.c_off;
  for (uint32_t q = threadIdx.x; q < segbytes; q += blockDim.x)
    c[q] = src[q];
  for (uint32_t q = (j.Kprime >> 3) + threadIdx.x; q < kbytes; q += blockDim.x)
    c[q] = 0;
  if (j.C > 1) {
    const uint32_t crc = tb_block_crc(src, j.Kprime - j.L, pow24b, red);
    if (threadIdx.x == 0) {
      c[segbytes] = (uint8_t)(crc >> 24);
      c[segbytes + 1] = (uint8_t)(crc >> 16);
      c[segbytes + 2] = (uint8_t)(crc >> 8);
    }
  }
}

/* ---- TX 3: bit selection (rate matching) fused with bit interleaving ---------------------------------------- *
 * f[i + j*Qm] = e[i*E/Qm + j] (nr_rate_matching.c:262-268), e[k] = d[position of rank (rank0 + k) mod V], where the
 * transmittable positions of the circular buffer [0, Ncb) are those outside the filler range. */
__global__ void __launch_bounds__(TB_THREADS) tb_tx_ratematch_kernel(const tb_tx_seg_job *jobs, const uint8_t *scratch,
                                                                     uint8_t *coded)
{
  const tb_tx_seg_job j = jobs[blockIdx.x];
  const uint8_t *__restrict__ d = scratch + j.d_off;
  uint8_t *__restrict__ f = coded + j.out_off;
  const uint32_t EQ = j.E / j.Qm;
  for (uint32_t m = threadIdx.x; m < j.E; m += blockDim.x) {
    const uint32_t jj = m / j.Qm, i = m - jj * j.Qm;
    const uint32_t k = i * EQ + jj;
    const uint32_t rank = (j.rank0 + k) % j.V;
    const uint32_t p = rank < j.Foffset ? rank : rank + j.Fin;
    f[m] = d[p];
  }
}

/* ---- TX 2+3 fused: segmentation + CB CRC + bit-packed LDPC encoding + rate matching + interleaving --------------
 * One workgroup per code block; the segment bytes, the code word (ldpc_enc_packed_core.h) and the selection all stay
 * in LDS: HBM traffic = the segment's payload bytes in, E output bytes out (no c / d round trip through scratch). */
/* Interleaver output of one chunk of modulation symbols from its QM packed sub-streams (tb_tx_fused_kernel):
 * f[sy * QM + i] = bit sy of sub-stream i.  A thread takes 8 symbols: one byte of every sub-stream in, 8 QM bytes out, every
 * shift a compile-time constant (2 VALU per output byte; the first version did a division and a look-up per byte).
 * `dst` = where the chunk's first symbol goes; its alignment decides the store width. */
template <int QM>
__device__ __forceinline__ void tb_tx_store_syms(const uint32_t *sel, uint32_t sel_stride, uint32_t nsym, uint8_t *__restrict__ dst, int tid,
                                                 int nt)
{
  const uint32_t ngrp = nsym >> 3, al = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
  for (uint32_t g = tid; g < ngrp; g += nt) {
    uint32_t win[QM];
#pragma unroll
    for (int i = 0; i < QM; i++)
      win[i] = sel[i * sel_stride + (g >> 2)] >> (8u * (g & 3u));
    /* output dword w of the group (compile-time shifts); formed right where it is stored, so that at most one is live
     * (all 2 QM of them next to the eight windows pushed the kernel to 99 VGPRs = four waves per SIMD, and a 1664-segment
     * slot then runs in two rounds of workgroups) */
    /* QM = 6, 8: the 8 QM output bits of the group as two bit strings first -- symbols 0..3 and 4..7; bit sy of a sub-stream goes
     * to position sy QM by ONE multiplication (x * (1 + 2^(QM-1) + 2^(2QM-2) + 2^(3QM-3)) puts bit k of a nibble at k + (QM-1) j
     * for j = 0..3, of which j = k is the wanted QM k; no two of the sixteen positions coincide when QM - 1 >= 4, so nothing
     * carries) -- then every nibble of a string becomes a dword of bytes by another one.  84 instead of ~150 VALU per group. */
    uint32_t f_lo = 0, f_hi = 0;
    if constexpr (QM == 6 || QM == 8) {
      constexpr uint32_t M = 1u | (1u << (QM - 1)) | (1u << (2 * QM - 2)) | (1u << (3 * QM - 3));
      constexpr uint32_t K = 1u | (1u << QM) | (1u << (2 * QM)) | (1u << (3 * QM));
#pragma unroll
      for (int i = 0; i < QM; i++) {
        f_lo |= (((win[i] & 0xfu) * M) & K) << i;
        f_hi |= ((((win[i] >> 4) & 0xfu) * M) & K) << i;
      }
    }
    auto word = [&](int w) -> uint32_t {
      if constexpr (QM == 6 || QM == 8) {
        const uint32_t nib = ((w < QM ? f_lo : f_hi) >> (4 * (w < QM ? w : w - QM))) & 0xfu;
        return (nib * 0x00204081u) & 0x01010101u;
      }
      uint32_t v = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int m = 4 * w + b, sy = m / QM, i = m - sy * QM;
        v |= ((win[i] >> sy) & 1u) << (8 * b);
      }
      return v;
    };
    uint8_t *o = dst + (size_t)g * (8 * QM);
    if (al == 0) {
#pragma unroll
      for (int w = 0; w < 2 * QM; w++)
        reinterpret_cast<uint32_t *>(o)[w] = word(w);
    } else if (al == 2) {
#pragma unroll
      for (int w = 0; w < 2 * QM; w++) {
        const uint32_t v = word(w);
        reinterpret_cast<uint16_t *>(o)[2 * w] = (uint16_t)v;
        reinterpret_cast<uint16_t *>(o)[2 * w + 1] = (uint16_t)(v >> 16);
      }
    } else {
#pragma unroll
      for (int w = 0; w < 2 * QM; w++) {
        const uint32_t v = word(w);
#pragma unroll
        for (int b = 0; b < 4; b++)
          o[4 * w + b] = (uint8_t)(v >> (8 * b));
      }
    }
  }
  for (uint32_t m = ngrp * 8u * QM + tid; m < nsym * QM; m += nt) { /* the last, partial group */
    const uint32_t sy = m / QM, i = m - sy * QM;
    dst[m] = (uint8_t)((sel[i * sel_stride + (sy >> 5)] >> (sy & 31u)) & 1u);
  }
}

typedef uint32_t tb_u32x4_t __attribute__((ext_vector_type(4)));
template <typename J> __device__ __forceinline__ uint32_t crc_len_of(J j) { return j->crc_len; }
__global__ void __launch_bounds__(512, 8) tb_tx_fused_kernel(const tb_tx_seg_job *jobs, const ldpc_enc_job *ejobs, const uint8_t *scratch,
                                                          uint8_t *coded, const uint32_t *pow24b, uint32_t *acc)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  typedef const tb_tx_seg_job LDPC_CONST_AS *seg_ptr_t;
  typedef const ldpc_enc_job LDPC_CONST_AS *enc_ptr_t;
  const seg_ptr_t j = (seg_ptr_t)jobs + blockIdx.x;
  const enc_ptr_t ej = (enc_ptr_t)ejobs + blockIdx.x;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)ej->code;
  const int Z = code->Z, tid = threadIdx.x, nt = blockDim.x;
  ldpc_encp_lds L;
  ldpc_encp_carve(reinterpret_cast<uint32_t *>(fsm), code, L);
  uint32_t *red = L.RP + code->nrows + 1;
  uint8_t *c = reinterpret_cast<uint8_t *>(red + 2);
#ifdef TB_TIMING /* diagnostic build (tools/tb_tx_timing.py): workgroup 0 logs the clock after every phase into the first
                    bytes BEHIND the last segment's output (the tool's buffer is that much longer) */
  long long *tlog = reinterpret_cast<long long *>(coded + ((jobs[gridDim.x - 1].out_off + jobs[gridDim.x - 1].E + 15) & ~15ull));
  int tlog_n = 0;
#define TB_TLOG() do { if (blockIdx.x == 0 && threadIdx.x == 0) tlog[tlog_n++] = clock64(); } while (0)
  /* + first / last clock (and the wall clock, 100 MHz) of every 128th workgroup behind the phase log: dispatch skew */
  if ((blockIdx.x & 127u) == 0 && threadIdx.x == 0) {
    tlog[32 + 4 * (blockIdx.x >> 7)] = clock64();
    tlog[32 + 4 * (blockIdx.x >> 7) + 2] = wall_clock64();
  }
#else
#define TB_TLOG() do { } while (0)
#endif
  TB_TLOG();

  /* c_r = b[r*(K'-L) ..] || CRC24B (C > 1) || zero fillers (nr_segmentation.c:147-175).  Everything the later phases
   * read from global memory is requested here, in one go: the segment's bytes, the encoder's edge table (the first part
   * of its phase 0), the eight table seeds and this thread's power of x for the CRC. */
  const uint32_t Kprime = j->Kprime, Lcrc = j->L, segbytes = (Kprime - Lcrc) >> 3, kbytes = (j->K + 7) >> 3;
  const uint8_t *src = scratch + j->b_off + (size_t)j->r * segbytes;
  uint32_t *tab = reinterpret_cast<uint32_t *>(c + 1056 + 16); /* CRC byte table: the selection area is free until the end */
  const bool with_crc = j->C > 1;
  /* CRC piece of this thread: bytes [P tid, P tid + P) of the segment, P = the fewest bytes that fit the pieces into ONE wave (the
   * recurrence over a piece is a chain of dependent table look-ups, but moving a piece's register to the end of the string
   * costs ~200 VALU instructions per wave that has a piece: the kernel is issue bound when a slot's segments fill the GPU); n_after = bits behind the piece */
  const uint32_t P = (segbytes + 63u) / 64u < 4u ? 4u : (segbytes + 63u) / 64u;
  const uint32_t q0 = P * (uint32_t)tid, qn = q0 < segbytes ? (segbytes - q0 < P ? segbytes - q0 : P) : 0u;
  const uint32_t n_after = 8u * (segbytes - q0 - qn);
  uint32_t xq = 0;
  if (with_crc) {
    tb_build_crc_tab(pow24b, tab);
    if (qn && n_after >= 24u)
      xq = pow24b[n_after - 24u]; /* x^n_after mod g, left aligned (pow[j] = x^(j + 24) mod g) */
  }
  /* the segment that ends the transport block takes the TB CRC from the accumulator the partial-CRC kernel left (and
   * clears it for the next call): no kernel of its own for three bytes per transport block */
  const uint32_t crc_len = j->crc_len, crc_pos = j->crc_pos;
  uint32_t tb_crc = 0;
  for (uint32_t k = 0; k < j->crc_nchunks; k++) /* (uniform addresses: scalar loads) */
    tb_crc ^= acc[j->crc_chunk0 + k];
  /* every global load of the stage goes out before anything is consumed (written as separate loops the compiler waited
   * for each one in turn: four round trips for the segment's bytes alone): the segment as aligned dwords -- two per LDS
   * dword, the source is only byte aligned --, the edge table, the row pointers */
  const uint32_t a0 = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u), ndw = (segbytes + 3u) >> 2;
  const uint32_t *__restrict__ src32 = reinterpret_cast<const uint32_t *>(src - a0);
  /* `scratch` is the caller's PAYLOAD array: no dword is touched that does not hold a byte of this transport block (the TB
   * CRC bytes, which the last segment's share of b ends with, do not exist there: they come from the accumulator below) */
  const uint32_t src_bytes = crc_len_of(j) ? j->crc_pos : segbytes, last_dw = (a0 + src_bytes + 3u) >> 2;
  uint32_t g_lo[5], g_hi[5], g_et[5], g_rp = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t w = (uint32_t)tid + (uint32_t)k * (uint32_t)nt;
    g_lo[k] = g_hi[k] = g_et[k] = 0;
    if (w < ndw) {
      if (w < last_dw)
        g_lo[k] = src32[w];
      if (w + 1 < last_dw)
        g_hi[k] = src32[w + 1];
    }
    if (w < (uint32_t)code->nedges)
      g_et[k] = code->enc_et[w];
  }
  if (tid <= code->nrows)
    g_rp = (uint32_t)code->row_ptr[tid];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t w = (uint32_t)tid + (uint32_t)k * (uint32_t)nt;
    if (w < ndw) {
      uint32_t v = __builtin_amdgcn_alignbyte(g_hi[k], g_lo[k], a0);
#pragma unroll
      for (int b = 0; b < 4; b++) { /* TB CRC bytes (last segment only); nothing behind the segment's last byte */
        const uint32_t q = 4u * w + (uint32_t)b, kk = q - crc_pos;
        if (kk < crc_len)
          v = (v & ~(0xffu << (8 * b))) | (((tb_crc >> (24 - 8 * kk)) & 0xffu) << (8 * b));
        if (q >= segbytes)
          v &= ~(0xffu << (8 * b));
      }
      reinterpret_cast<uint32_t *>(c)[w] = v;
    }
    if (w < (uint32_t)code->nedges)
      L.ET[w] = g_et[k];
  }
  for (uint32_t w = (uint32_t)tid + 5u * (uint32_t)nt; w < ndw; w += nt) { /* (workgroups of fewer than 64 threads: never) */
    uint32_t v = __builtin_amdgcn_alignbyte(w + 1 < last_dw ? src32[w + 1] : 0u, w < last_dw ? src32[w] : 0u, a0);
    for (int b = 0; b < 4; b++) {
      const uint32_t q = 4u * w + (uint32_t)b, kk = q - crc_pos;
      if (kk < crc_len)
        v = (v & ~(0xffu << (8 * b))) | (((tb_crc >> (24 - 8 * kk)) & 0xffu) << (8 * b));
      if (q >= segbytes)
        v &= ~(0xffu << (8 * b));
    }
    reinterpret_cast<uint32_t *>(c)[w] = v;
  }
  for (int e = tid + 5 * nt; e < code->nedges; e += nt)
    L.ET[e] = code->enc_et[e];
  if (tid <= code->nrows)
    L.RP[tid] = g_rp;
  for (int r = tid + nt; r <= code->nrows; r += nt)
    L.RP[r] = (uint32_t)code->row_ptr[r];
  for (uint32_t q = 4u * ndw + tid; q < kbytes + 8; q += nt) /* fillers (and the CRC bytes' place, until they are known) */
    c[q] = 0;
  if (tid == 0)
    red[0] = 0;
  {
    const int W = ldpc_encp_W(Z), bs = W + 1;
    for (int col = tid; col < code->ncols; col += nt)
      L.B[col * bs + W] = 0u;
    for (int i = tid; i < 4 * bs; i += nt)
      L.LB[i] = 0u;
  }
  __syncthreads();
  TB_TLOG();
  if (with_crc) {
    /* CB CRC24B over the segment's bytes in LDS: a thread runs the byte-table recurrence of crc_byte.c:184-218 over its
     * piece, then moves its 24-bit register R to the end of the string: R(x) * x^n_after mod g, one Horner step per
     * coefficient of R with x^n_after mod g from the power table (one load per thread, requested above).  The first
     * version looked up one power per set BIT in global memory, one dependent load after the other: 40 % of the kernel. */
    uint32_t reg = 0;
    for (uint32_t i = 0; i < qn; i++)
      reg = (reg << 8) ^ tab[(reg >> 24) ^ c[q0 + i]];
    uint32_t x = reg;
    if (qn && n_after) {
      if (n_after >= 24u) {
        x = tb_crc_mulmod<24>(reg, xq, 0x80006300u); /* (crc_byte.c:50: poly24b) */
      } else { /* 8 or 16 bits behind the piece: as many zero bytes through the table */
        for (uint32_t b = 0; b < n_after; b += 8)
          x = (x << 8) ^ tab[x >> 24];
      }
    }
    for (int off = 32; off; off >>= 1)
      x ^= __shfl_xor(x, off);
    if ((tid & 63) == 0 && x)
      atomicXor(&red[0], x);
    __syncthreads();
    if (tid < 3)
      c[segbytes + tid] = (uint8_t)(red[0] >> (24 - 8 * tid));
    __syncthreads();
  }
  TB_TLOG();
  if (ldpc_encp32_applies(code)) {
    /* Zc % 32 == 0 (ldpc_enc_packed32.h): the information columns are the segment's dwords with byte and bit order reversed,
     * every produced word goes straight into its periodic string -- no extension phases, three barriers fewer */
    const ldpc_encp32 g32 = ldpc_encp32_make(code, ej->Kb);
    const uint32_t nd = (uint32_t)(g32.kbf * g32.W);
    for (uint32_t i = (uint32_t)tid; i < nd; i += nt)
      ldpc_encp32_info(L, g32, i, reinterpret_cast<const uint32_t *>(c)[i]);
    __syncthreads();
    TB_TLOG();
    TB_TLOG();
    ldpc_encp32_lambda(L, g32, tid, nt);
    __syncthreads();
    TB_TLOG();
    TB_TLOG();
    if (tid < 64)
      ldpc_encp32_core_parity_wave(code, L, g32, tid);
    __syncthreads();
    TB_TLOG();
    ldpc_encp32_extension(code, L, g32, tid, nt);
    __syncthreads();
    TB_TLOG();
  } else {
  {
    /* the rest of the encoder's phase 0: information columns from the MSB-first bytes */
    const int kbf = code->kb_full, W = ldpc_encp_W(Z), bs = W + 1, nin = (kbf * Z + 7) >> 3;
    for (int i = tid; i < kbf * W; i += nt) {
      const int col = i / W, w = i - col * W;
      const uint32_t b0 = (uint32_t)(col * Z + 32 * w), j0 = b0 >> 3;
      uint64_t v = 0;
      for (int q = 0; q < 5; q++)
        v = (v << 8) | ((int)j0 + q < nin ? c[j0 + q] : 0u);
      const uint32_t m = (uint32_t)(v >> (8 - (b0 & 7u)));
      L.B[col * bs + w] = __builtin_bitreverse32(m) & ldpc_encp_mask(Z, w);
    }
  }
  __syncthreads();
  TB_TLOG();
  for (int ph = 1; ph <= 3; ph++) {
    ldpc_encp_phase(ph, code, ej->Kb, c, L, nullptr, tid, nt);
    __syncthreads();
    TB_TLOG();
  }
  /* phases 4 .. 11 -- the four core parity columns, one after the other, W <= 12 items each: ONE wave walks them without
   * workgroup barriers (ldpc_encp_core_parity_wave), the others wait at the barrier below.  Eight barrier-separated phases of 12 active lanes were 10 k of the kernel's 62 k clocks. */
  if (tid < 64)
    ldpc_encp_core_parity_wave(code, L, tid);
  __syncthreads();
  TB_TLOG();
  ldpc_encp_phase(12, code, ej->Kb, c, L, nullptr, tid, nt);
  __syncthreads();
  TB_TLOG();
  }
  /* Bit selection + interleaving (nr_rate_matching.c:424-501, :240-303): f[i + jj*Qm] = e[i*E/Qm + jj],
   * e[k] = d[position of rank (rank0 + k) mod V], d[p] = code word bit p + 2Z.  In two steps per chunk of TB_TX_SEL_SYMS
   * modulation symbols: (1) the Qm sub-streams e[i*E/Qm + jj0 ..] are packed into LDS, 32 bits per item, gathered from
   * the code word in runs (a run ends at the circular buffer's wrap, at the filler gap, at the end of a lifted column);
   * (2) one thread per 8 symbols: a byte of every sub-stream in, 8 Qm output bytes out (tb_tx_store_syms).  (The first version stored one byte per thread and bit, strided by Qm: 1664 segments x 9450 one-byte
   * stores were most of the kernel's 50 us.) */
  uint8_t *__restrict__ f = coded + j->out_off;
  const uint32_t E = j->E, Qm = j->Qm, EQ = E / Qm, V = j->V, rank0 = j->rank0, Foffset = j->Foffset, Fin = j->Fin;
  const uint32_t z_magic = 0xffffffffu / (uint32_t)Z + 1u, bs = (uint32_t)ldpc_encp_W(Z) + 1u, twoZ = 2u * (uint32_t)Z;
  const uint32_t v_magic = V > 1u ? 0xffffffffu / V + 1u : 0u; /* (V = 1: every rank is 0) */
  uint32_t *sel = reinterpret_cast<uint32_t *>(c + 1056 + 16); /* [Qm][TB_TX_SEL_SYMS / 32 + 1], behind the segment bytes */
  const uint32_t sel_stride = TB_TX_SEL_SYMS / 32 + 1;
  for (uint32_t jj0 = 0; jj0 < EQ; jj0 += TB_TX_SEL_SYMS) {
    const uint32_t nsym = EQ - jj0 < TB_TX_SEL_SYMS ? EQ - jj0 : TB_TX_SEL_SYMS, nw = (nsym + 31) >> 5;
    for (uint32_t it = tid; it < Qm * nw; it += nt) {
      const uint32_t i = it / nw, w = it - i * nw;
      const uint32_t k = i * EQ + jj0 + 32u * w;
      uint32_t nbits = nsym - 32u * w;
      nbits = nbits > 32u ? 32u : nbits;
      /* (rank0 + k) mod V without a division: quotient from the reciprocal, off by one at most either way */
      const uint32_t x = rank0 + k, q = __umulhi(x, v_magic);
      uint32_t r = x - q * V, v = 0, filled = 0;
      r += (int32_t)r < 0 ? V : 0u;
      r -= r >= V ? V : 0u;
      r = V == 1u ? 0u : r;
      while (filled < nbits) {
        const uint32_t p = (r < Foffset ? r : r + Fin) + twoZ;
        const uint32_t col = __umulhi(p, z_magic), t = p - col * (uint32_t)Z;
        uint32_t n = nbits - filled;
        n = n < V - r ? n : V - r;
        if (r < Foffset)
          n = n < Foffset - r ? n : Foffset - r;
        n = n < (uint32_t)Z - t ? n : (uint32_t)Z - t;
        uint32_t chunk = ldpc_bits_at(L.B + col * bs, t);
        if (n < 32u)
          chunk &= (1u << n) - 1u;
        v |= chunk << filled;
        filled += n;
        r += n;
        r = r >= V ? r - V : r;
      }
      sel[i * sel_stride + w] = v;
    }
    __syncthreads();
    TB_TLOG();
    /* output bytes [jj0 Qm, (jj0 + nsym) Qm) of the segment */
    uint8_t *dst = f + (size_t)jj0 * Qm;
    switch (Qm) {
      case 1: tb_tx_store_syms<1>(sel, sel_stride, nsym, dst, tid, nt); break;
      case 2: tb_tx_store_syms<2>(sel, sel_stride, nsym, dst, tid, nt); break;
      case 4: tb_tx_store_syms<4>(sel, sel_stride, nsym, dst, tid, nt); break;
      case 6: tb_tx_store_syms<6>(sel, sel_stride, nsym, dst, tid, nt); break;
      case 8: tb_tx_store_syms<8>(sel, sel_stride, nsym, dst, tid, nt); break;
      default: /* (no such modulation in NR; kept correct) */
        for (uint32_t m = tid; m < nsym * Qm; m += nt) {
          const uint32_t sy = m / Qm, i = m - sy * Qm;
          dst[m] = (uint8_t)((sel[i * sel_stride + (sy >> 5)] >> (sy & 31u)) & 1u);
        }
    }
    __syncthreads();
    TB_TLOG();
  }
#ifdef TB_TIMING
  if ((blockIdx.x & 127u) == 0 && threadIdx.x == 0) {
    tlog[32 + 4 * (blockIdx.x >> 7) + 1] = clock64();
    tlog[32 + 4 * (blockIdx.x >> 7) + 3] = wall_clock64();
  }
#endif
}

/* ---- RX 1: de-interleave + rate de-match (HARQ combining) + decoder input pack --------------------------------
 * nr_deinterleaving_ldpc (nr_rate_matching.c:310-388): e[i*E/Qm + jj] = f[i + jj*Qm];
 * nr_rate_matching_ldpc_rx (:507-603): w[pos(k)] += e[k] (int16, wrapping), pos = circular-buffer position of rank
 * (rank0 + k) mod V among the non-filler positions, after clearing w[0..Ncb) on the first round;
 * caller's pack (nr_ulsch_decoding.c:195-210): punctured columns 0, fillers +127, saturate to int8.
 * One workgroup per code segment; the per-thread phases live in tb_rx_core.h (shared with the fused segment kernel,
 * tb_rx_fused.hip, where the same work is the prologue of the segment's decoder workgroup). */
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) tb_rx_dematch_kernel(const tb_rx_seg_job *jobs, const int16_t *llr,
                                                                   int16_t *harq, int8_t *scratch)
{
  extern __shared__ __attribute__((aligned(16))) int16_t e_lds[];
  typedef const tb_rx_seg_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t j = (job_ptr_t)jobs + blockIdx.x; /* uniform address: the job stays in SGPRs */
  const tb_rx_geom g = tb_rx_geometry(j);
  tb_rx_dematch_block(g, j->Qm, llr + j->llr_off, harq + j->harq_off, scratch + j->l_off, e_lds);
}

/* ---- RX 2: reassemble b from the decoded segments, TB CRC, payload out --------------------------------------------
 * per segment: copy its payload bytes into b and the payload buffer, partial TB CRC of those bytes into acc[tb];
 * per TB: ACK = every segment decoded and (C == 1 or the CRC register of the whole b is zero). */
__global__ void __launch_bounds__(TB_THREADS) tb_rx_assemble_kernel(const tb_rx_tb_job *jobs, const tb_rx_seg_job *segs,
                                                                    const int32_t *n_iter, uint8_t *scratch, uint8_t *payload,
                                                                    uint32_t *acc, const int *tb_abort, const uint32_t *pow24a,
                                                                    const uint32_t *pow16)
{
  __shared__ uint32_t tab[256];
  const tb_rx_seg_job sj = segs[blockIdx.x];
  const tb_rx_tb_job j = jobs[sj.tb];
  const uint32_t *pow = j.crc_type == 0 ? pow24a : pow16;
  tb_build_crc_tab(pow, tab);
  uint8_t *b = scratch + j.b_off;
  const uint32_t bbytes = j.B >> 3, abytes = j.A >> 3, first = sj.r * j.seg_bytes;
  /* A transport block with a failed segment is lost as a whole: its siblings may have given up half way (the abort flag
   * is final here, every decoder workgroup has finished), so what it delivers is defined as all-zero bytes. */
  const bool ok = n_iter[sj.iter_idx] <= (int)j.num_max_iter && !(tb_abort && tb_abort[sj.tb]);
  const uint8_t *c = scratch + sj.c_off;
  uint32_t count = 0;
  if (first < bbytes)
    count = first + j.seg_bytes <= bbytes ? j.seg_bytes : bbytes - first;
  for (uint32_t q = threadIdx.x; q < count; q += blockDim.x) {
    const uint8_t v = ok ? c[q] : (uint8_t)0; /* the reference leaves stale bytes for a failed segment; here: zeros */
    b[first + q] = v;
    if (first + q < abytes)
      payload[j.payload_off + first + q] = v;
  }
  if (j.C > 1 && ok) {
    __syncthreads(); /* the CRC below re-reads b */
    uint32_t x; /* short pieces when the launch is small (latency), long ones when it fills the GPU (work): tb_chain.h */
    if (gridDim.x <= 512u)
      x = j.crc_type == 0 ? tb_partial_crc<8, 24>(b, j.B, first, count, pow, tab) : tb_partial_crc<8, 16>(b, j.B, first, count, pow, tab);
    else
      x = j.crc_type == 0 ? tb_partial_crc<32, 24>(b, j.B, first, count, pow, tab) : tb_partial_crc<32, 16>(b, j.B, first, count, pow, tab);
    if ((threadIdx.x & 63) == 0 && x)
      atomicXor(&acc[sj.tb], x);
  }
}
__global__ void __launch_bounds__(TB_THREADS) tb_rx_verdict_kernel(const tb_rx_tb_job *jobs, uint32_t n_tb, const int32_t *n_iter,
                                                                   uint32_t *acc, int *tb_abort, uint8_t *ack, int32_t *iter_max)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tb)
    return;
  const tb_rx_tb_job j = jobs[i];
  if (j.fused) /* delivered by the block's last segment to finish (tb_rx_fused.hip) */
    return;
  bool all_ok = true;
  int imax = 0;
  for (uint32_t r = 0; r < j.C; r++) {
    const int it = n_iter[j.seg0 + r];
    imax = it > imax ? it : imax;
    all_ok &= it <= (int)j.num_max_iter;
  }
  /* single-segment TBs were CRC-checked inside the decoder (phy_procedures_nr_gNB.c:293-299) */
  ack[i] = (uint8_t)(all_ok && (j.C == 1 || acc[i] == 0));
  acc[i] = 0; /* left zero for the next call (see tb_tx_crc_final_kernel) */
  if (tb_abort)
    tb_abort[i] = 0;
  /* a segment that gave up because a sibling had failed reports numMaxIter + 2 (decoder.c:556-559); which siblings get
   * that far is a matter of timing, so the per-TB figure is capped at "failed" = numMaxIter + 1 */
  iter_max[i] = imax > (int)j.num_max_iter + 1 ? (int)j.num_max_iter + 1 : imax;
}

#define TB_LAUNCH(kernel, n, s, ...)                                              \
  do {                                                                            \
    if ((n) == 0)                                                                 \
      return hipSuccess;                                                          \
    hipLaunchKernelGGL(kernel, dim3(n), dim3(TB_THREADS), 0, s, __VA_ARGS__);     \
    return hipGetLastError();                                                     \
  } while (0)

hipError_t tb_launch_tx_crc(const tb_tx_tb_job *jobs, uint32_t n_tb, const tb_crc_chunk_job *chunks, uint32_t n_chunks,
                            const uint8_t *payload, uint8_t *scratch, uint32_t *acc, const uint32_t *pow24a,
                            const uint32_t *pow16, int with_final, hipStream_t s)
{
  if (n_tb == 0)
    return hipSuccess;
  hipLaunchKernelGGL(tb_tx_crc_partial_kernel, dim3(n_chunks), dim3(TB_THREADS), 0, s, jobs, chunks, payload, scratch, acc, pow24a, pow16,
                     with_final /* the unfused path reads b; the fused segment kernel reads the payload */);
  if (with_final)
    hipLaunchKernelGGL(tb_tx_crc_final_kernel, dim3((n_tb + TB_THREADS - 1) / TB_THREADS), dim3(TB_THREADS), 0, s, jobs, n_tb, scratch, acc);
  return hipGetLastError();
}
hipError_t tb_launch_tx_segment(const tb_tx_seg_job *jobs, uint32_t n, uint8_t *scratch, const uint32_t *pow24b, hipStream_t s)
{
  TB_LAUNCH(tb_tx_segment_kernel, n, s, jobs, scratch, pow24b);
}
hipError_t tb_launch_tx_ratematch(const tb_tx_seg_job *jobs, uint32_t n, const uint8_t *scratch, uint8_t *coded, hipStream_t s)
{
  TB_LAUNCH(tb_tx_ratematch_kernel, n, s, jobs, scratch, coded);
}
hipError_t tb_launch_tx_fused(const tb_tx_seg_job *jobs, const ldpc_enc_job *ejobs, uint32_t n, int n_threads, int lds_bytes,
                              const uint8_t *scratch, uint8_t *coded, const uint32_t *pow24b, uint32_t *acc, hipStream_t s)
{
  if (n == 0)
    return hipSuccess;
  hipLaunchKernelGGL(tb_tx_fused_kernel, dim3(n), dim3(n_threads), lds_bytes, s, jobs, ejobs, scratch, coded, pow24b, acc);
  return hipGetLastError();
}
hipError_t tb_launch_rx_dematch(const tb_rx_seg_job *jobs, uint32_t n, uint32_t lds_elems, const int16_t *llr, int16_t *harq,
                                int8_t *scratch, hipStream_t s, int wide)
{
  if (n == 0)
    return hipSuccess;
  /* wide: a launch that gives every segment a CU of its own (one transport block: 13 -> 5 us) takes 1024 threads per
   * workgroup -- a segment's time is then the latency of its strided loops, not the GPU's throughput */
  hipLaunchKernelGGL(tb_rx_dematch_kernel, dim3(n), dim3(wide ? 1024 : TB_THREADS), (size_t)lds_elems * sizeof(int16_t), s, jobs, llr, harq,
                     scratch);
  return hipGetLastError();
}
hipError_t tb_launch_rx_assemble(const tb_rx_tb_job *jobs, uint32_t n_tb, const tb_rx_seg_job *segs, uint32_t n_seg,
                                 const int32_t *n_iter, uint8_t *scratch, uint8_t *payload, uint8_t *ack, int32_t *iter_max,
                                 uint32_t *acc, int *tb_abort, const uint32_t *pow24a, const uint32_t *pow16, hipStream_t s)
{
  if (n_tb == 0)
    return hipSuccess;
  hipLaunchKernelGGL(tb_rx_assemble_kernel, dim3(n_seg), dim3(TB_THREADS), 0, s, jobs, segs, n_iter, scratch, payload, acc, tb_abort, pow24a, pow16);
  hipLaunchKernelGGL(tb_rx_verdict_kernel, dim3((n_tb + TB_THREADS - 1) / TB_THREADS), dim3(TB_THREADS), 0, s, jobs, n_tb, n_iter, acc, tb_abort, ack, iter_max);
  return hipGetLastError();
}
