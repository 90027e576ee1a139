/*
 * nr_coding_host.h -- host-side arithmetic of the transport-block chain around the LDPC codec
 * (no GPU dependency; unit-tested on CPU against the oracle).
 *
 * Mirrors, with the same argument meaning and results:
 *   nr_segmentation()            openair1/PHY/CODING/nr_segmentation.c:32-140 (parameter computation part)
 *   nr_get_E()                   openair1/PHY/NR_TRANSPORT/nr_tbs_tools.c:50-64
 *   nr_get_R_ldpc_decoder()      openair1/PHY/CODING/nr_rate_matching.c:390-421
 *   lenWithCrc() / crcType()     openair1/PHY/defs_gNB.h:224-235
 *   rate-matching start/limits   openair1/PHY/CODING/nr_rate_matching.c:440-452 (Ncb, k0)
 */
#ifndef NR_CODING_HOST_H
#define NR_CODING_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NR_HIP_CRC24_A 0
#define NR_HIP_CRC24_B 1
#define NR_HIP_CRC16 2
#define NR_HIP_CRC8 3
#define NR_HIP_MAX_PDSCH_TBS 3824 /* openair1/PHY/defs_nr_common.h:84 */

typedef struct {
  uint32_t B;      /* TB bits incl. TB CRC */
  uint32_t C;      /* code blocks */
  uint32_t L;      /* per-CB CRC length: 0 (C == 1) or 24 */
  uint32_t Kprime; /* bits per CB incl. CB CRC, before fillers */
  uint32_t K;      /* 22*Zc / 10*Zc */
  uint32_t Zc;
  uint32_t F;      /* filler bits per CB */
  uint32_t Kb;     /* information columns entering the parity (return value of nr_segmentation) */
} nr_hip_seg_t;

/* 0 on success; -1 when B does not fit (Z > 384) or the byte-alignment assumptions of the reference's copy
 * loops do not hold ((Kprime - L) % 8, F % 8). */
int nr_hip_segmentation(uint32_t B, int BG, nr_hip_seg_t *s);
uint32_t nr_hip_get_E(uint32_t G, uint32_t C, uint32_t Qm, uint32_t Nl, uint32_t r);
int nr_hip_get_R_ldpc_decoder(int rvidx, int E, int BG, int Z, int *llrLen, int round);
static inline int nr_hip_len_with_crc(int nbSeg, int len)
{
  if (nbSeg > 1)
    return (len + 24 + 24 * nbSeg) / nbSeg;
  return len + (len > NR_HIP_MAX_PDSCH_TBS ? 24 : 16);
}
static inline int nr_hip_crc_type(int nbSeg, int len)
{
  if (nbSeg > 1)
    return NR_HIP_CRC24_B;
  return len > NR_HIP_MAX_PDSCH_TBS ? NR_HIP_CRC24_A : NR_HIP_CRC16;
}

/* Rate matching geometry of one code block: circular buffer length Ncb, filler range [Foffset, Foffset+F),
 * start position ind0 (k0, moved behind the fillers when it falls inside them), V = number of transmittable
 * positions per lap, rank0 = rank of ind0 among them.  The k-th transmitted bit is the transmittable position of
 * rank (rank0 + k) mod V. */
typedef struct {
  uint32_t Ncb, Foffset, Fin, V, rank0;
} nr_hip_rm_t;
int nr_hip_rate_match_geometry(uint32_t Tbslbrm, int BG, uint32_t Zc, uint32_t C, uint32_t F, uint32_t K, int rv,
                               uint32_t E, nr_hip_rm_t *g);
/* columns of the decoder input a first transmission of E values can reach (the rest is zero) */
uint32_t nr_hip_first_tx_columns(const nr_hip_rm_t *g, uint32_t E, uint32_t Zc);
/* check_crc() of openair1/PHY/CODING/crc_byte.c:314-380 for any n: 1 when the CRC of the first n - 8 L bits (MSB first,
 * L = 3 / 3 / 2 / 1 bytes for crc_type 0..3) equals the L bytes in front of byte n >> 3; 0 otherwise and for an unknown type */
int nr_hip_check_crc(const uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type);
#ifdef __cplusplus
}
#endif
#endif
