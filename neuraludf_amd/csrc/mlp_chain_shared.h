// Pieces shared by the two fused-chain kernels (mlp_chain.hip: workgroup-shared 32/64-point tiles;
// mlp_chain_rows.hip: wave-private 32-point tiles): tile geometry, the positional encoding, softplus derivatives.
#pragma once
#include "nudf_common.h"
#include "../../include/nudf.h"

#define CH_LD 292          // activation row stride in floats (K <= 288): m*292 mod 64 = 36m -> 16 consecutive rows hit
                           // 16 distinct multiples of 4 -> conflict-free ds_read_b128
#define CH_THREADS 256
#define CH_LD16 296        // row stride of the 16-bit activation tile in HALFWORDS (K <= 288): 592 bytes = 148 dwords, and
                           // 148 m mod 32 = 0, 20, 8, 28, 16, 4, 24, 12: the 16-byte reads of 8 consecutive rows cover the 32
                           // banks exactly -> conflict-free ds_read_b128

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// element (r, c) of a [P, ld] buffer in the BLOCKED layout of nudf.h
__device__ __forceinline__ size_t ch_blk_off(int r, int c, int ld) {
  return (size_t)(r >> 5) * 32 * ld + (size_t)(c >> 2) * 128 + (size_t)(r & 31) * 4 + (c & 3);
}

// 16-bit stored state (NUDF_CH_STATE16, nudf.h): bf16, FOUR CONSECUTIVE POINTS of one feature packed into 8 bytes --
// element (row, col) of a [R, ld] buffer sits at ((row / 4) ld + col) 4 + row % 4.  A lane of a 32x32 accumulator tile
// holds rows 8 g + 4 h + {0..3} of its column: one 8-byte store / load per group g instead of four 2-byte ones, 32
// lanes = 256 contiguous bytes; and a dword is the (k, k + 1) row pair of one column that the 16-bit weight-gradient
// GEMM's MFMA operand image is made of.
__device__ __forceinline__ size_t ch_p4_off(unsigned row, unsigned col, unsigned ld) {
  return ((size_t)(row >> 2) * ld + col) * 4 + (row & 3);
}
// bf16 <-> fp32 of the 16-bit stored state (round to nearest even on the way out)
__device__ __forceinline__ float ch_bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned short ch_f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }

__device__ __forceinline__ f32x16 ch_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// softplus'(a) = s and 1 - s from the STORED activation h = softplus100(a) / xscale (see gemm_f32_mfma.hip)
__device__ __forceinline__ void ch_sp_derivs(float hstored, float xscale, float& s, float& om) {
  const float x = 100.0f * xscale * hstored;
  if (x > 20.0f) {
    s = 1.0f;
    om = 0.0f;
  } else {
    const float e = __expf(-x);
    om = e;
    s = (x < 0.01f) ? x * (1.0f - x * (0.5f - x * 0.16666667f)) : 1.0f - e;
  }
}

// positional-encoding element c of point x (value or JVP with tangent v); same arithmetic as posenc_kernel
__device__ __forceinline__ float ch_pe(const float* x3, const float* v3, int c, int L, float in_scale, int jvp) {
  const int blk = c / 3, j = c - blk * 3;
  const float xv = x3[j] * in_scale;
  if (blk == 0) return jvp ? v3[j] * in_scale : xv;
  const int k = (blk - 1) >> 1;
  const float f = (float)(1 << k);
  const float a = xv * f;
  const bool is_sin = ((blk - 1) & 1) == 0;
  if (!jvp) return is_sin ? sinf(a) : cosf(a);
  return (is_sin ? cosf(a) : -sinf(a)) * f * v3[j] * in_scale;
}

// write PE(x) (or its JVP) * scale into activation columns [col0, col0 + E) of a `rows`-point LDS tile (+ optional
// global mirror), NTHR cooperating threads.  One work item per (point, coordinate, octave): ONE sincosf gives the sin
// and the cos column of that octave (the per-element form called sinf or cosf once per column: 2.2x the libm calls;
// the PE of a 64-point tile took 46 k cycles, 6 % of a forward sweep).  Same arguments 2^k x, same libm kernels.
// TFMT: element type of the LDS tile -- 0 fp32 (row stride CH_LD floats), 1 fp16, 2 bf16 (row stride CH_LD16 halfwords: the
// 16-bit-tile chain kernel, whose tile IS the MFMA operand)
// fp32 -> fp16 for the 16-bit tile.  The fp32-tile kernel converts its operands with v_cvt_pk_f16_f32 (what hipcc emits for a
// vector fptrunc on gfx950); a scalar `(_Float16)x` becomes v_cvt_f16_f32, which treats fp16 DENORMAL results differently
// (measured: softplus outputs below 6.1e-5 made 0.3 % of the points differ between the two kernels).  The packed instruction
// is named here so that both kernels round every value the same way.
__device__ __forceinline__ unsigned ch_f2h2(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned short ch_f2h(float x) { return (unsigned short)(ch_f2h2(x, 0.0f) & 0xffffu); }
template <int NTHR>
__device__ __forceinline__ void ch_write_pe_rows(float* act, const float* xs, const float* vs, int rows, int tid,
                                                 const NudfChain& p, int m0, int col0, float scale, float* gdst,
                                                 int ldg, int gcol0, int zero_to, bool dst16 = false,
                                                 bool dstblk = false, int tfmt = 0, float gscale = 1.0f) {
  if (tfmt != 0) {   // 16-bit tile: same work items, same values, rounded once on their way into the tile
    unsigned short* a16 = reinterpret_cast<unsigned short*>(act);
    const int L = p.pe_L;
    const int E = 3 * (2 * L + 1);
    for (int it = tid; it < rows * 3 * (L + 1); it += NTHR) {
      const int rj = it / (L + 1), k = it - rj * (L + 1) - 1;
      const int r = rj / 3, j = rj - 3 * r;
      const float xv = xs[rj] * p.pe_in_scale;
      const float tv = vs[rj] * p.pe_in_scale;
      unsigned short* arow = a16 + r * CH_LD16 + col0;
      const bool mirror = gdst && (m0 + r) < p.P;
      const size_t goff = (size_t)(m0 + r) * ldg + gcol0;
      auto put = [&](int c, float val) {
        val *= scale;
        arow[c] = (tfmt == 2) ? ch_f2bf(val) : ch_f2h(val);
        if (mirror) {
          if (dst16) reinterpret_cast<unsigned short*>(gdst)[ch_p4_off(m0 + r, gcol0 + c, ldg)] = ch_f2bf(val);
          else gdst[goff + c] = val;
        }
      };
      if (k < 0) {
        put(j, p.pe_jvp ? tv : xv);
      } else {
        const float f = (float)(1 << k);
        float sn, cs;
        sincosf(xv * f, &sn, &cs);
        put(3 + 6 * k + j, p.pe_jvp ? cs * f * tv : sn);
        put(6 + 6 * k + j, p.pe_jvp ? -sn * f * tv : cs);
      }
    }
    const int npad = zero_to - (col0 + E);
    if (npad > 0)
      for (int e = tid; e < rows * npad; e += NTHR) {
        const int r = e / npad, c = e - r * npad;
        a16[r * CH_LD16 + col0 + E + c] = 0;
      }
    return;
  }
  const int L = p.pe_L;
  const int E = 3 * (2 * L + 1);
  for (int it = tid; it < rows * 3 * (L + 1); it += NTHR) {
    const int rj = it / (L + 1), k = it - rj * (L + 1) - 1;   // k = -1: the identity column
    const int r = rj / 3, j = rj - 3 * r;
    const float xv = xs[rj] * p.pe_in_scale;
    const float tv = vs[rj] * p.pe_in_scale;
    float* arow = act + r * CH_LD + col0;
    const bool mirror = gdst && (m0 + r) < p.P;
    const size_t goff = (size_t)(m0 + r) * ldg + gcol0;
    auto put = [&](int c, float val) {
      val *= scale;
      arow[c] = val;
      if (mirror) {
        if (dst16) reinterpret_cast<unsigned short*>(gdst)[ch_p4_off(m0 + r, gcol0 + c, ldg)] = ch_f2bf(val);
        else if (dstblk) gdst[ch_blk_off(m0 + r, gcol0 + c, ldg)] = val;
        else gdst[goff + c] = val * gscale;        // (NudfChain.tile_scale: the tile holds sigma x, memory holds x)
      }
    };
    if (k < 0) {
      put(j, p.pe_jvp ? tv : xv);
    } else {
      const float f = (float)(1 << k);
      float sn, cs;
      sincosf(xv * f, &sn, &cs);
      put(3 + 6 * k + j, p.pe_jvp ? cs * f * tv : sn);
      put(6 + 6 * k + j, p.pe_jvp ? -sn * f * tv : cs);
    }
  }
  // zero padding columns [col0 + E, zero_to) so that the K padding of the next GEMM multiplies finite zeros
  const int npad = zero_to - (col0 + E);
  if (npad > 0)
    for (int e = tid; e < rows * npad; e += NTHR) {
      const int r = e / npad, c = e - r * npad;
      act[r * CH_LD + col0 + E + c] = 0.0f;
    }
}

// accumulator register r of a 32x32 MFMA tile holds row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#define CH_KOFF(r) (((r) & 3) + 8 * ((r) >> 2))

#define CH_USES_X1(e)                                                                                           \
  ((e) == NUDF_CH_MULSP || (e) == NUDF_CH_TANGENT || (e) == NUDF_CH_BWD || (e) == NUDF_CH_MULMASK || (e) == NUDF_CH_ADDMASK)

#define CH_USES_X2(e) ((e) == NUDF_CH_TANGENT || (e) == NUDF_CH_BWD || (e) == NUDF_CH_ADDMASK || (e) == NUDF_CH_RELUADD)

// UDFNetwork.udf_out (fields.py:184-190) and its derivative, the per-point multiplier every sweep behind the head uses:
// type 0 'abs' (|v|, sign v: every shipped conf), 1 'square' (v^2, 2 v), 2 'sdf' (v, 1).  The type travels in the head
// step's iparam (unused by this epilogue otherwise).
__device__ __forceinline__ void ch_udf_head(int type, float v, float scale, float& out, float& mult) {
  if (type == 1) { out = v * v * scale; mult = 2.0f * v; }
  else if (type == 2) { out = v * scale; mult = 1.0f; }
  else { out = fabsf(v) * scale; mult = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }
}
