// Fused MLP layer chains for the UDF network (gfx950): one workgroup owns a tile of TM points and keeps
// the [TM x <=256] activation tile RESIDENT IN LDS across all layers of a sweep.  Per layer the tile is
// the MFMA A operand (ds_read_b128, bank-conflict-free at row stride 260), the weights stream from L2 in
// MFMA-B fragment order straight into registers (one coalesced global_load_dwordx4 per lane per 32-column
// tile per 8 k), and the epilogue writes the next layer's activations back into the same LDS tile -- HBM is
// touched only for what a later backward sweep needs (stored activations) and for epilogue operands that
// were stored by an earlier sweep.  v_mfma_f32_32x32x2_f32: exact fp32.
//
// Four sweeps of models/fields.py:192-231 (UDFNetwork.forward / .gradient and their autograd backward,
// including the second-order part the reference gets from autograd.grad(create_graph=True)):
//   forward   X[l+1] = softplus100(X[l] W_l^T + b_l) (* 1/sqrt2 before the skip layer), abs head
//   gradient  DA[l-1] = (DA[l] W_l) * softplus'(.)          (reverse sweep for d udf / d x)
//   tangent   R[l+1] = (R[l] W_l^T) * softplus'(.), EX[l] = (R[l] W_l^T) * DA[l] * softplus''/softplus'
//   adjoint   ABAR[l-1] = (ABAR[l] W_l) * softplus'(.) + EX[l-1]
// described to the kernel as a table of steps (NudfChain in include/nudf.h).
//
// The K index inside each group of 8 is permuted (lane half h contracts k = 8g + 4h + j in MFMA j) so that
// one 16-byte LDS read / one 16-byte global read feeds 4 consecutive MFMAs; A and B use the same permutation,
// so only the fp32 summation order differs from a k-ordered GEMM.
#include "nudf_common.h"
#include "../../include/nudf.h"

#include "mlp_chain_shared.h"
#include <stdlib.h>
#include <type_traits>

template <int TM>
struct ChainSmem {
  float act[TM * CH_LD];
  float xs[TM * 3];
  float vs[TM * 3];
  float red[8];          // workgroup reductions (NudfChain.tile_scale / tile_amax_out): one value per wave
};

// MODE 3 (16-bit-tile kernel, BASELINE config 5): the activation tile holds the MFMA operand type of the chain's steps (fp16 in
// the forward sweeps, bf16 in the backward ones): 37 KB instead of 75 KB per 64-point workgroup -> three workgroups per CU
template <int TM>
struct ChainSmem16 {
  unsigned short act[TM * CH_LD16];
  float xs[TM * 3];
  float vs[TM * 3];
  float red[8];
};

template <int TM, class SM>
__device__ __forceinline__ void ch_write_pe(SM& sm, const NudfChain& p, int m0, int col0, float scale,
                                            float* gdst, int ldg, int gcol0, int zero_to, bool dst16 = false, int tfmt = 0,
                                            float gscale = 1.0f) {
  ch_write_pe_rows<CH_THREADS>(reinterpret_cast<float*>(sm.act), sm.xs, sm.vs, TM, threadIdx.x, p, m0, col0, scale, gdst, ldg,
                               gcol0, zero_to, dst16, false, tfmt, gscale);
}

// K loop of one step for an NRT x NCT block of 32x32 tiles: two register sets, no copies and no branches in
// the body, so the compiler's s_waitcnt counters let the NEXT group's LDS / L2 reads stay in flight under the
// current group's 4*NRT*NCT MFMAs.  G (groups of 8 k) is even; the last prefetch re-reads the last group.

// Stored-activation operand of the epilogue (X1), fetched for ALL tiles of the wave while the last two k groups
// are still being multiplied: the HBM latency of the epilogue operands then hides under MFMAs instead of
// sitting between the barrier and the first epilogue instruction.
struct ChPrefetch {
  const float* X1;
  int ldx1;
  unsigned vo[2][2];   // per-lane element offset of (row tile i, col tile j), r = 0
};

template <int NRT, int NCT, bool PF>
__device__ __forceinline__ void ch_mma(const float* __restrict__ arow, const f32x4* __restrict__ bptr, size_t bstride,
                                       int G, f32x16 (&acc)[2][2], const ChPrefetch& pf, float (&px1)[2][2][16]) {
  f32x4 a0[NRT], a1[NRT], b0[NCT], b1[NCT];
#pragma unroll
  for (int i = 0; i < NRT; ++i) a0[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD);
#pragma unroll
  for (int j = 0; j < NCT; ++j) b0[j] = bptr[j * 64];
#pragma unroll 1
  for (int g = 0; g < G - 2; g += 2) {
    {
      const f32x4* bq = bptr + (size_t)(g + 1) * bstride;
#pragma unroll
      for (int j = 0; j < NCT; ++j) b1[j] = bq[j * 64];
#pragma unroll
      for (int i = 0; i < NRT; ++i) a1[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + (g + 1) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMA block (the scheduler would sink it)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(a0[i][jj], b0[j][jj], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    {
      const f32x4* bq = bptr + (size_t)(g + 2) * bstride;
#pragma unroll
      for (int j = 0; j < NCT; ++j) b0[j] = bq[j * 64];
#pragma unroll
      for (int i = 0; i < NRT; ++i) a0[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + (g + 2) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(a1[i][jj], b1[j][jj], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // last two groups (G is even): no further weight prefetch; the epilogue operands are issued AFTER the last
  // weight loads so that the in-order vmcnt lets them stay in flight under the remaining MFMAs
  {
    const f32x4* bq = bptr + (size_t)(G - 1) * bstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) b1[j] = bq[j * 64];
#pragma unroll
    for (int i = 0; i < NRT; ++i) a1[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + (G - 1) * 8);
    if (PF) {
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) px1[i][j][r] = (pf.X1 + (size_t)CH_KOFF(r) * pf.ldx1)[pf.vo[i][j]];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(a0[i][jj], b0[j][jj], acc[i][j]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(a1[i][jj], b1[j][jj], acc[i][j]);
  __builtin_amdgcn_sched_barrier(0);
}

// 16-bit-operand K loop (BASELINE config 5: fp16 / bf16 MFMA operands, fp32 accumulate, fp32 everywhere else):
// v_mfma_f32_32x32x16_{f16,bf16}; lane (i, h) contracts k = 16g + 8h .. +7, i.e. two ds_read_b128 of the fp32
// activation tile converted on the fly (v_cvt_pk_*_f32, round-to-nearest-even) and one 16-byte read of the
// pre-converted weight fragments per 32-column tile.  16x the fp32 MFMA rate and a separate matrix pipe: the
// sweep becomes epilogue / LDS bound.
#ifndef NUDF_STAGGER16
#define NUDF_STAGGER16 0u    // s_sleep(127) units the odd wave slots of the 16-bit instantiation start late (A/B switch;
#endif                       // measured 0..4 at the config-5 shape: 7.53 / 7.68 / 7.77 / 7.87 / 7.98 ms -- pure delay there)
#ifndef NUDF_STAGGER32
#define NUDF_STAGGER32 2u    // the same for the fp32 instantiation's 64-point tiles (0 also switches the 32-point tiles' stagger off)
#endif
#ifndef NUDF_MMA16_REPEAT
#define NUDF_MMA16_REPEAT 1  // timing probe: every 16-bit MFMA issued this many times (what a split-operand fp32 emulation costs)
#endif
#ifndef NUDF_MMA16_RING
#define NUDF_MMA16_RING 1    // A/B build switch (scripts/build_variants.sh): 0 = one k step of weight fragments in flight
#endif
template <int NRT, int NCT, bool BF>
__device__ __forceinline__ void ch_mma16(const float* __restrict__ arow, const uint4* __restrict__ bptr, size_t bstride,
                                         int G16, f32x16 (&acc)[2][2]) {
  // (requesting the epilogue's stored-state operands under this loop, as the fp32 loop does, was measured in round 2:
  // 64 more live registers next to the conversions -> 185-511 spilled registers, chains 7.2 -> 17.8 ms at config 5)
  f32x4 a0[NRT][2], a1[NRT][2];
#if !NUDF_MMA16_RING
  uint4 b0[NCT], b1[NCT];
#endif
  auto lda = [&](f32x4 (&a)[NRT][2], int g) {
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      a[i][0] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 16);
      a[i][1] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 16 + 4);
    }
  };
  auto ldb = [&](uint4 (&b)[NCT], int g) {
    const uint4* bq = bptr + (size_t)g * bstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) b[j] = bq[j * 64];
  };
  auto mma = [&](f32x4 (&a)[NRT][2], uint4 (&b)[NCT]) {
#pragma unroll
    for (int rep = 0; rep < NUDF_MMA16_REPEAT; ++rep)     // > 1: timing probes only (results are garbage)
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      const f32x8 av = {a[i][0][0], a[i][0][1], a[i][0][2], a[i][0][3], a[i][1][0], a[i][1][1], a[i][1][2], a[i][1][3]};
#pragma unroll
      for (int j = 0; j < NCT; ++j) {
        if (BF)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_convertvector(av, bf16x8),
                                                              __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_convertvector(av, f16x8),
                                                             __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
      }
    }
  };
#if NUDF_MMA16_RING
  // Weight fragments come from L2 (~600-800 cycles) and a 16-wide k step is only 4 MFMAs = 128 matrix cycles per wave:
  // with the fragments of ONE step in flight (the fp32 loop's depth, where a step is 1024 cycles) two waves per SIMD keep
  // the pipe at ~50 % (scripts/chain_timeline.py, 16-bit mode).  Ring of 4 fragment sets = three k steps in flight; the
  // prefetch index is clamped instead of guarded (branch-free body, exact vmcnt), the last G16 % 4 steps run as a tail.
  uint4 br[4][NCT];
  const int gl = G16 - 1;
  lda(a0, 0);
  ldb(br[0], 0);
  ldb(br[1], min(1, gl));
  ldb(br[2], min(2, gl));
  int g = 0;
#pragma unroll 1
  for (; g + 4 <= G16; g += 4) {
    ldb(br[3], min(g + 3, gl));
    lda(a1, min(g + 1, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, br[0]);
    __builtin_amdgcn_sched_barrier(0);
    ldb(br[0], min(g + 4, gl));
    lda(a0, min(g + 2, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, br[1]);
    __builtin_amdgcn_sched_barrier(0);
    ldb(br[1], min(g + 5, gl));
    lda(a1, min(g + 3, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, br[2]);
    __builtin_amdgcn_sched_barrier(0);
    ldb(br[2], min(g + 6, gl));
    lda(a0, min(g + 4, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, br[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int rem = G16 - g;          // 0..3 steps left: fragments in br[0..rem-1], activations of step g in a0
  if (rem > 0) {
    if (rem > 1) lda(a1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, br[0]);
    if (rem > 1) {
      if (rem > 2) lda(a0, g + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, br[1]);
      if (rem > 2) mma(a0, br[2]);
    }
  }
#else
  lda(a0, 0);
  ldb(b0, 0);
  int g = 0;
#pragma unroll 1
  for (; g + 1 < G16; g += 2) {
    lda(a1, g + 1);
    ldb(b1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int gn = (g + 2 < G16) ? g + 2 : g + 1;
    lda(a0, gn);
    ldb(b0, gn);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (g < G16) mma(a0, b0);   // odd number of 16-wide k steps
#endif
}

// 16-bit-TILE K loop (MODE 3): the LDS tile already holds the operand type, row-major [row][CH_LD16] halfwords -- lane (i, h)
// reads its 8 consecutive k of row i as ONE ds_read_b128 per 32-row tile and k step (the fp32-tile loop above: two reads and
// four v_cvt_pk per tile and step, in every one of the four waves that share the tile), the weight fragments stream exactly
// as in ch_mma16 (ring of four sets, three k steps in flight).  Same operands in the same order as ch_mma16: the results
// are bit-identical to the fp32-tile 16-bit kernel's.
template <int NRT, int NCT, bool BF>
__device__ __forceinline__ void ch_mma16t(const unsigned short* __restrict__ arow, const uint4* __restrict__ bptr,
                                          size_t bstride, int G16, f32x16 (&acc)[2][2]) {
  uint4 a0[NRT], a1[NRT];
  uint4 br[4][NCT];
  auto lda = [&](uint4 (&a)[NRT], int g) {
#pragma unroll
    for (int i = 0; i < NRT; ++i) a[i] = *reinterpret_cast<const uint4*>(arow + i * 32 * CH_LD16 + g * 16);
  };
  auto ldb = [&](uint4 (&b)[NCT], int g) {
    const uint4* bq = bptr + (size_t)g * bstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) b[j] = bq[j * 64];
  };
  auto mma = [&](const uint4 (&a)[NRT], const uint4 (&b)[NCT]) {
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) {
        if (BF)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                              acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]),
                                                             acc[i][j], 0, 0, 0);
      }
  };
  const int gl = G16 - 1;
  lda(a0, 0);
  ldb(br[0], 0);
  ldb(br[1], min(1, gl));
  ldb(br[2], min(2, gl));
  int g = 0;
#pragma unroll 1
  for (; g + 4 <= G16; g += 4) {
    ldb(br[3], min(g + 3, gl));
    lda(a1, min(g + 1, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, br[0]);
    __builtin_amdgcn_sched_barrier(0);
    ldb(br[0], min(g + 4, gl));
    lda(a0, min(g + 2, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, br[1]);
    __builtin_amdgcn_sched_barrier(0);
    ldb(br[1], min(g + 5, gl));
    lda(a1, min(g + 3, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, br[2]);
    __builtin_amdgcn_sched_barrier(0);
    ldb(br[2], min(g + 6, gl));
    lda(a0, min(g + 4, gl));
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, br[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int rem = G16 - g;          // 0..3 steps left: fragments in br[0..rem-1], activations of step g in a0
  if (rem > 0) {
    if (rem > 1) lda(a1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, br[0]);
    if (rem > 1) {
      if (rem > 2) lda(a0, g + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, br[1]);
      if (rem > 2) mma(a0, br[2]);
    }
  }
}

// bf16x3 K loop (NudfChainStep.prec == 3): the fp32 product EMULATED on the bf16 matrix pipe, which is 16x faster than
// the fp32 one.  Every fp32 value is the exact sum of three bf16 parts, x = hi + mid + lo (round to nearest each: 8 + 8 + 8
// significant bits with signed remainders cover fp32's 24); the weights are split once per optimizer step by the pack kernel
// (three fragment planes), the activations on the fly from the fp32 LDS tile (3 v_cvt_pk + 4 subtractions / shifts per
// pair).  Six of the nine partial products are formed -- hi hi, hi mid, mid hi, hi lo, lo hi, mid mid -- with fp32
// accumulation; the three dropped ones (mid lo, lo mid, lo lo) are bounded by 2^-23 |x| |y| per product, the size of one
// fp32 rounding, and random in sign.  12 matrix-pipe cycles per k and tile instead of the fp32 MFMA's 32.
// Lane (i, h) contracts k = 16 g + 8 h .. + 7 as in ch_mma16; per k step a wave issues 6 NRT NCT MFMAs (768 cycles for
// 2 x 2 tiles) against 3 NCT weight loads, 2 NRT LDS reads and ~7 VALU operations per activation element.
typedef float ch_f32x8v __attribute__((ext_vector_type(8)));
typedef float ch_f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 ch_bf16x2v __attribute__((ext_vector_type(2)));
// two values at a time: ONE v_cvt_pk_bf16_f32 gives both leading parts, the shift / mask widen them back, two exact
// subtractions leave the remainders -- 5 + 5 + 1 VALU operations per pair (the vector form below let the compiler convert
// every element a second time on its own to widen it: 7.5 per element)
__device__ __forceinline__ void ch_split3_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(ch_f32x2v{x0, x1}, ch_bf16x2v));
  const float r0 = x0 - __builtin_bit_cast(float, p0 << 16);
  const float r1 = x1 - __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(ch_f32x2v{r0, r1}, ch_bf16x2v));
  const float s0 = r0 - __builtin_bit_cast(float, p1 << 16);
  const float s1 = r1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(ch_f32x2v{s0, s1}, ch_bf16x2v));
}
__device__ __forceinline__ void ch_split3(const f32x4& lo4, const f32x4& hi4, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  uint4 a, b, c;
  ch_split3_pair(lo4[0], lo4[1], a.x, b.x, c.x);
  ch_split3_pair(lo4[2], lo4[3], a.y, b.y, c.y);
  ch_split3_pair(hi4[0], hi4[1], a.z, b.z, c.z);
  ch_split3_pair(hi4[2], hi4[3], a.w, b.w, c.w);
  p0 = __builtin_bit_cast(bf16x8, a);
  p1 = __builtin_bit_cast(bf16x8, b);
  p2 = __builtin_bit_cast(bf16x8, c);
}
#ifndef NUDF_X3_BDIST
#define NUDF_X3_BDIST 1     // A/B build switch: k steps the weight fragments are requested ahead.  2 (three register sets, one
                            // set of split activations; 5 spilled registers in the 64-point tile) measured the same as 1:
                            // the fetch is not latency-bound (profiles/r04_bf16x3_experiments.txt item 9)
#endif
#ifndef NUDF_X3_PIPE
#define NUDF_X3_PIPE 2      // the next step's split interleaved with this step's MFMAs (sched_group_barrier pattern): 0 = never,
                            // 1 = always (measured slower on the 64-point tiles: 4.63 vs 4.44 ms per step -- two in-phase waves
                            // per SIMD already cover each other's split), 2 = only where a wave has its SIMD to itself: the
                            // 32-point tiles of the 8 192-point launches (81-85 -> 70-75 us per launch, experiments item 10)
#endif
template <int NRT, int NCT, bool PIPE>
__device__ __forceinline__ void ch_mma16x3(const float* __restrict__ arow, const uint4* __restrict__ bptr, size_t bstride3,
                                           int G16, f32x16 (&acc)[2][2]) {
  // bptr: this lane's uint4 of plane 0 of column tile ct0 in k step 0; planes 64 uint4 apart, column tiles 192, k steps
  // bstride3 = 3 * NT * 64.
  // Software pipeline: the split of step g + 1's activations (VALU) is interleaved with step g's MFMAs, four VALU
  // operations behind every MFMA (sched_group_barrier pins the pattern) -- a wave that runs the split in front of its MFMAs
  // leaves the matrix pipe to its partner alone for ~350 cycles per step, and with both workgroups of a CU in phase that
  // is idle time (measured: forward sweep 436 us back to back).
  f32x4 raw[NRT][2];
  bf16x8 pa[2][NRT][3];
  uint4 b[2][NCT][3];
  auto lda = [&](int g) {
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      raw[i][0] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 16);
      raw[i][1] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 16 + 4);
    }
  };
  auto ldb = [&](uint4 (&bb)[NCT][3], int g) {
#ifdef NUDF_X3_PROBE_B0      // timing probe only (wrong results): every k step re-reads step 0's fragments -- L1 hits
    g = 0;
#endif
    const uint4* bq = bptr + (size_t)g * bstride3;
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bb[j][pl] = bq[j * 192 + pl * 64];
  };
  auto split = [&](bf16x8 (&q)[NRT][3]) {
#pragma unroll
    for (int i = 0; i < NRT; ++i) ch_split3(raw[i][0], raw[i][1], q[i][0], q[i][1], q[i][2]);
  };
  auto mfmas = [&](const bf16x8 (&q)[NRT][3], const uint4 (&bb)[NCT][3]) {
    // smallest terms first; (A part, B plane) pairs: hi lo, lo hi, mid mid, hi mid, mid hi, hi hi
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          const bf16x8 av = (t == 0 || t == 3 || t == 5) ? q[i][0] : ((t == 1) ? q[i][2] : q[i][1]);
          const uint4 bv = (t == 0) ? bb[j][2] : ((t == 2 || t == 3) ? bb[j][1] : bb[j][0]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bv), acc[i][j], 0, 0, 0);
        }
  };
  auto pattern = [&]() {
    if constexpr (PIPE) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRT, 0);     // the next step's LDS reads ...
      __builtin_amdgcn_sched_group_barrier(0x020, 3 * NCT, 0);     // ... and weight loads first
#pragma unroll
      for (int m = 0; m < 6 * NRT * NCT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (44 * NRT + 6 * NRT * NCT - 1) / (6 * NRT * NCT), 0);
      }
    }
  };
  const int gl = G16 - 1;
#if NUDF_X3_BDIST == 2
  // Weight fragments requested TWO k steps ahead (three register sets in rotation), one set of split activations: the
  // split of step g + 1 runs after step g's MFMAs have been issued and have read their operands, so it may overwrite them.
  // (Built because a probe with every k step re-reading step 0's fragments -- L1 hits, NUDF_X3_PROBE_B0 -- runs every chain
  // launch 12-17 % faster; the distance turned out not to be the reason.)
  {
    uint4 b3[3][NCT][3];
    lda(0);
    ldb(b3[0], 0);
    ldb(b3[1], min(1, gl));
    split(pa[0]);
    int g3 = 0;
#define CH_X3_STEP(cur, nxt2, gg)                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    lda(min((gg) + 1, gl));                                         \
    ldb(b3[nxt2], min((gg) + 2, gl));                               \
    __builtin_amdgcn_sched_barrier(0);                              \
    mfmas(pa[0], b3[cur]);                                          \
    split(pa[0]);
#pragma unroll 1
    for (; g3 + 3 <= G16; g3 += 3) {
      CH_X3_STEP(0, 2, g3)
      CH_X3_STEP(1, 0, g3 + 1)
      CH_X3_STEP(2, 1, g3 + 2)
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g3 + 2 <= G16) {          // two steps left: sets 0 and 1 hold them
      CH_X3_STEP(0, 2, g3)
      __builtin_amdgcn_sched_barrier(0);
      mfmas(pa[0], b3[1]);
    } else if (g3 < G16) {
      mfmas(pa[0], b3[0]);
    }
#undef CH_X3_STEP
    return;
  }
#endif
  lda(0);
  ldb(b[0], 0);
  split(pa[0]);
  int g = 0;
#pragma unroll 1
  for (; g + 2 <= G16; g += 2) {
    __builtin_amdgcn_sched_barrier(0);
    lda(min(g + 1, gl));
    ldb(b[1], min(g + 1, gl));
    if constexpr (!PIPE) __builtin_amdgcn_sched_barrier(0);
    mfmas(pa[0], b[0]);
    split(pa[1]);
    pattern();
    __builtin_amdgcn_sched_barrier(0);
    lda(min(g + 2, gl));
    ldb(b[0], min(g + 2, gl));
    if constexpr (!PIPE) __builtin_amdgcn_sched_barrier(0);
    mfmas(pa[1], b[1]);
    split(pa[0]);
    pattern();
  }
  __builtin_amdgcn_sched_barrier(0);
  if (g < G16) mfmas(pa[0], b[0]);   // odd number of 16-wide k steps
}

// f16x2 K loop (NudfChainStep.prec == 4): the fp32 product emulated on the fp16 matrix pipe with THREE products instead of
// bf16x3's six (round 6: the matrix-heavy launches run against the 1 400 W package limit -- the lever is MFMA products per
// fp32 product, not cycles).  x = hi + 2^-11 lo with hi = fp16(x), lo = fp16((x - hi) 2^11): fp16 carries 11 significant bits,
// the remainder x - hi is exact in fp32 and at most half an ulp of hi, and scaled by 2^11 it sits in fp16's normal range
// whenever hi does -- 22 bits + the sign of the remainder.  acc0 += hi hi' (exact products), acc1 += hi lo' + lo hi' (the
// correction terms, in their OWN fp32 accumulator so that their scale never meets acc0's rounding), result acc0 + 2^-11 acc1
// (Ootomo & Yokota's form); dropped: lo lo' <= 2^-22 |x| |y|.  Range: fp16's (|x| < 65504; below 6e-5 the parts go
// denormal and the split keeps an absolute resolution of ~3e-11) -- the forward-order sweeps' operands (encodings in [-1, 1],
// softplus / ReLU activations, weight-normed weights), not the 1e-6 ... 1e-9 adjoints of the backward sweeps, which stay
// on bf16x3.  Weights: two fragment planes per (k step, column tile) from the pack kernel (NudfPackFrag.dtype 4), 4 B per
// weight instead of 6.  Same geometry as ch_mma16x3: lane (i, h) contracts k = 16 g + 8 h .. + 7; per k step a wave issues
// 3 NRT NCT MFMAs (384 matrix cycles for 2 x 2 tiles), 2 NCT weight loads, 2 NRT LDS reads and 2.5-4 VALU operations per
// activation element (one packed conversion gives both high parts, the remainders are fp32 subtractions of the widened parts,
// scaled, and converted once more).
typedef _Float16 ch_f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ch_split2_pair(float x0, float x1, unsigned& p0, unsigned& p1) {
  const ch_f16x2v hv = __builtin_convertvector(ch_f32x2v{x0, x1}, ch_f16x2v);
  p0 = __builtin_bit_cast(unsigned, hv);
  const float r0 = (x0 - (float)hv[0]) * 2048.0f;
  const float r1 = (x1 - (float)hv[1]) * 2048.0f;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(ch_f32x2v{r0, r1}, ch_f16x2v));
}
__device__ __forceinline__ void ch_split2(const f32x4& lo4, const f32x4& hi4, f16x8& p0, f16x8& p1) {
  uint4 a, b;
  ch_split2_pair(lo4[0], lo4[1], a.x, b.x);
  ch_split2_pair(lo4[2], lo4[3], a.y, b.y);
  ch_split2_pair(hi4[0], hi4[1], a.z, b.z);
  ch_split2_pair(hi4[2], hi4[3], a.w, b.w);
  p0 = __builtin_bit_cast(f16x8, a);
  p1 = __builtin_bit_cast(f16x8, b);
}
template <int NRT, int NCT, bool PIPE>
__device__ __forceinline__ void ch_mma16x2(const float* __restrict__ arow, const uint4* __restrict__ bptr, size_t bstride2,
                                           int G16, f32x16 (&acc)[2][2]) {
  // bptr: this lane's uint4 of plane 0 (hi) of column tile ct0 in k step 0; the lo plane 64 uint4 on, column tiles 128, k
  // steps bstride2 = 2 * NT * 64.
  f32x4 raw[NRT][2];
  f16x8 pa[2][NRT][2];
  uint4 b[2][NCT][2];
  f32x16 acc1[NRT][NCT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.0f;
  auto lda = [&](int g) {
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
      raw[i][0] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 16);
      raw[i][1] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 16 + 4);
    }
  };
  auto ldb = [&](uint4 (&bb)[NCT][2], int g) {
#ifdef NUDF_X2_PROBE_B0      // timing probe only (wrong results): every k step re-reads step 0's fragments -- L1 hits
    g = 0;
#endif
    const uint4* bq = bptr + (size_t)g * bstride2;
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bb[j][pl] = bq[j * 128 + pl * 64];
  };
  auto split = [&](f16x8 (&q)[NRT][2]) {
#ifdef NUDF_X2_PROBE_NOSPLIT     // timing probe only (wrong results): the operands are the raw LDS words -- what a K loop costs
#pragma unroll                   // whose tile already holds the two fp16 parts (the split done once, by the producing epilogue)
    for (int i = 0; i < NRT; ++i) {
      uint4 u0 = __builtin_bit_cast(uint4, raw[i][0]), u1 = __builtin_bit_cast(uint4, raw[i][1]);
      u0.x &= 0x33ff33ffu; u0.y &= 0x33ff33ffu; u0.z &= 0x33ff33ffu; u0.w &= 0x33ff33ffu;      // finite, |.| < 0.25
      u1.x &= 0x33ff33ffu; u1.y &= 0x33ff33ffu; u1.z &= 0x33ff33ffu; u1.w &= 0x33ff33ffu;
      q[i][0] = __builtin_bit_cast(f16x8, u0);
      q[i][1] = __builtin_bit_cast(f16x8, u1);
    }
#else
#pragma unroll
    for (int i = 0; i < NRT; ++i) ch_split2(raw[i][0], raw[i][1], q[i][0], q[i][1]);
#endif
  };
  auto mfmas = [&](const f16x8 (&q)[NRT][2], const uint4 (&bb)[NCT][2]) {
    // the two correction products first (their accumulator), then hi hi'
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          const f16x8 av = (t == 1) ? q[i][1] : q[i][0];
          const f16x8 bv = __builtin_bit_cast(f16x8, (t == 0) ? bb[j][1] : bb[j][0]);
          if (t < 2) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc1[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i][j], 0, 0, 0);
        }
  };
  auto pattern = [&]() {
    if constexpr (PIPE) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRT, 0);     // the next step's LDS reads ...
      __builtin_amdgcn_sched_group_barrier(0x020, 2 * NCT, 0);     // ... and weight loads first
#pragma unroll
      for (int m = 0; m < 3 * NRT * NCT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (32 * NRT + 3 * NRT * NCT - 1) / (3 * NRT * NCT), 0);
      }
    }
  };
  const int gl = G16 - 1;
  // (Weight fragments TWO k steps ahead -- a k step is 384 matrix cycles per wave against 600-800 of L2 latency, and a probe
  // that re-reads step 0's fragments, NUDF_X2_PROBE_B0, runs the f16x2 sweeps 7-12 % faster -- was built as a ring of three
  // fragment sets: 208 live registers next to the two accumulator sets, and hipcc spilled 397 of them.  Not kept.)
  {
  lda(0);
  ldb(b[0], 0);
  split(pa[0]);
  int g = 0;
#pragma unroll 1
  for (; g + 2 <= G16; g += 2) {
    __builtin_amdgcn_sched_barrier(0);
    lda(min(g + 1, gl));
    ldb(b[1], min(g + 1, gl));
    if constexpr (!PIPE) __builtin_amdgcn_sched_barrier(0);
    mfmas(pa[0], b[0]);
    split(pa[1]);
    pattern();
    __builtin_amdgcn_sched_barrier(0);
    lda(min(g + 2, gl));
    ldb(b[0], min(g + 2, gl));
    if constexpr (!PIPE) __builtin_amdgcn_sched_barrier(0);
    mfmas(pa[1], b[1]);
    split(pa[0]);
    pattern();
  }
  __builtin_amdgcn_sched_barrier(0);
  if (g < G16) mfmas(pa[0], b[0]);   // odd number of 16-wide k steps
  }
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(acc1[i][j][r], 1.0f / 2048.0f, acc[i][j][r]);
}

// 16-bit stored state, 4-point packed (ch_p4_off): the 16 accumulator rows of a lane are 4 groups of 4 consecutive
// points = 4 accesses of 8 bytes.  q0 = the lane's first point quad (tile row 0 + 4 h) / 4, groups are 2 quads apart.
typedef __bf16 ch_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ch_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ch_p4_widen(const uint2& w, float* x) {
  x[0] = __builtin_bit_cast(float, w.x << 16);
  x[1] = __builtin_bit_cast(float, w.x & 0xffff0000u);
  x[2] = __builtin_bit_cast(float, w.y << 16);
  x[3] = __builtin_bit_cast(float, w.y & 0xffff0000u);
}
__device__ __forceinline__ void ch_p4_load(const float* X, int ld, unsigned q0, unsigned col, uint2 (&w)[4]) {
  const uint2* Xq = reinterpret_cast<const uint2*>(X) + (size_t)q0 * ld + col;
#pragma unroll
  for (int g = 0; g < 4; ++g) w[g] = Xq[(size_t)(2 * g) * ld];
}
__device__ __forceinline__ void ch_p4_store(float* C, int ld, unsigned q0, unsigned col, const float (&v)[16]) {
  uint2* Cq = reinterpret_cast<uint2*>(C) + (size_t)q0 * ld + col;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint2 w;
    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(ch_f32x2{v[4 * g], v[4 * g + 1]}, ch_bf16x2));
    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(ch_f32x2{v[4 * g + 2], v[4 * g + 3]}, ch_bf16x2));
    Cq[(size_t)(2 * g) * ld] = w;
  }
}

// epilogue of one step for this wave's tiles: new activations -> LDS (+ HBM where a later sweep needs them)
// One 32x32 accumulator tile of a step's epilogue.  Branch-free per element: uniform options are tested once
// around whole 16-element loops, the column predicate (N may end inside the tile) is one exec region, rows
// need no predicate because every output / operand buffer is row-padded to the tile size (see nudf.h).
// Global addresses are <uniform row pointer> + <per-lane 32-bit offset>.
// S16: the step's stored-state arrays (X1, X2, C1, and the TANGENT mirror C2) hold bf16 (config-5 mode, NUDF_CH_STATE16)
// RT16: ReLU-family step in the 16-bit instantiation -- X1 / C1 are bf16 (packed) where the step's layout bits say so
// T16: the activation tile is the 16-bit one of the MODE 3 kernel (`act` points at halfwords, row stride CH_LD16; tile_bf: it
// holds bf16, else fp16) -- the new activations are rounded ONCE here, on their way into the tile, instead of in the K loop of
// each of the four waves that read them (same values: the next step's MFMAs see the same operands)
struct ChNoX3 {};
// Returns the largest |value| the tile contributes to NudfChain.absmax_out (its C1 output and its rank-1 operand); callers that
// do not track it drop the result and the compiler drops its computation.
template <int EPI, bool X2IN = false, bool S16 = false, bool RT16 = false, bool T16 = false, bool HX3 = false, class X3V = ChNoX3>
__device__ __forceinline__ float ch_epilogue_tile(const NudfChainStep& st, float* act, int m0, int rtile, int ctile,
                                                 int h, int ln, f32x16 a, float (&x1)[16], bool load_x1,
                                                 const float* x2in = nullptr, const float* bias_pre = nullptr,
                                                 bool tile_bf = false, const X3V& x3in = X3V(), float tsig = 1.0f,
                                                 float tinv = 1.0f) {
  // tsig / tinv (NudfChain.tile_scale): the activation tile of a LINEAR sweep holds sigma times the values -- operands that
  // enter from memory (the rank-1 term, X2 of BWD / ADDMASK) are multiplied by sigma, what goes to memory by 1 / sigma.  Powers
  // of two: exact.
  float tmax = 0.0f;
  const int col = ctile * 32 + ln;
  const bool col_ok = col < st.N;
  const unsigned colc = col_ok ? col : 0;
  const int r0 = rtile * 32 + 4 * h;
  const unsigned grow0 = (unsigned)(m0 + r0);
  float v[16], out[16];
  {
    // (bias_pre: requested by the caller before the K loop -- the load here is an exposed L2 round trip per tile)
    const float bias = bias_pre ? *bias_pre : (st.bias ? st.bias[colc] : 0.0f);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = a[r] + bias;
  }
  if (st.r1_row) {  // rank-1 term (column 0 of the abs head in the adjoint sweep)
    const float r1c = st.r1_col[colc];
    const unsigned vo = grow0 * (unsigned)st.ldr1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float rv = (st.r1_row + (size_t)CH_KOFF(r) * st.ldr1)[vo];
      tmax = fmaxf(tmax, fabsf(rv));        // (NudfChain.absmax_out: the rank-1 operand is a GEMM operand too)
      v[r] += (rv * tsig) * r1c;
    }
  }
  float x2[16];   // x1 (the stored activation) was prefetched under the fp32 K loop; the short 16-bit loops load it here
  if (CH_USES_X1(EPI) && load_x1) {
    const unsigned vo = grow0 * (unsigned)st.ldx1 + colc;
    if (S16 || (RT16 && (st.layout & NUDF_CH_P4_X1))) {
      uint2 w[4];
      ch_p4_load(st.X1, st.ldx1, (grow0 >> 2), colc, w);
#pragma unroll
      for (int g = 0; g < 4; ++g) ch_p4_widen(w[g], x1 + 4 * g);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) x1[r] = (st.X1 + (size_t)CH_KOFF(r) * st.ldx1)[vo];
    }
  }
  if (CH_USES_X2(EPI)) {
    if (X2IN) {   // fetched by the caller one tile ahead (ch_epilogue_seq)
#pragma unroll
      for (int r = 0; r < 16; ++r) x2[r] = x2in[r];
    } else if (st.X2) {
      const unsigned vo = grow0 * (unsigned)st.ldx2 + colc;
      if (S16) {
        uint2 w[4];
        ch_p4_load(st.X2, st.ldx2, (grow0 >> 2), colc, w);
#pragma unroll
        for (int g = 0; g < 4; ++g) ch_p4_widen(w[g], x2 + 4 * g);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) x2[r] = (st.X2 + (size_t)CH_KOFF(r) * st.ldx2)[vo];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) x2[r] = 0.0f;
    }
  }
  if (EPI == NUDF_CH_BWD || EPI == NUDF_CH_ADDMASK) {      // X2 joins the (scaled) adjoint
#pragma unroll
    for (int r = 0; r < 16; ++r) x2[r] *= tsig;
  }
  // NUDF_CH_BWD with a third stored operand (NudfChainStep.X3): x2 = R[l], x3 = DA[l-1] -- the second-order term is formed
  // below instead of read.  Fetched by the caller one tile ahead (x3in) or here.
  // (HX3 is a template flag: the two-operand BWD path keeps its register allocation)
  constexpr bool has_x3 = HX3 && EPI == NUDF_CH_BWD;
  float x3[16];
  if constexpr (has_x3) {
    if constexpr (!std::is_same<X3V, ChNoX3>::value) {      // fetched by the caller one tile ahead (a register array)
#pragma unroll
      for (int r = 0; r < 16; ++r) x3[r] = x3in[r];
    } else {
      const unsigned vo = grow0 * (unsigned)st.ldx3 + colc;
      if (S16) {
        uint2 w[4];
        ch_p4_load(st.X3, st.ldx3, (grow0 >> 2), colc, w);
#pragma unroll
        for (int g = 0; g < 4; ++g) ch_p4_widen(w[g], x3 + 4 * g);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) x3[r] = (st.X3 + (size_t)CH_KOFF(r) * st.ldx3)[vo];
      }
    }
  }
  float out2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (EPI == NUDF_CH_SOFTPLUS) {
      // softplus100(v) = [max(t, 0) + log1p(exp(-|t|))] / 100, t = 100 v, from the two hardware transcendentals
      // (v_exp_f32 / v_log_f32 are base 2).  Select-free on purpose: the compiler turns selects around
      // transcendentals into per-element branches.  Above torch's threshold (t > 20) the log term is < 2.1e-9,
      // i.e. the result equals v to fp32 resolution; for t << 0 the absolute error of log2(1 + z) is < 1e-9.
      const float t = 100.0f * v[r];
      const float z = __builtin_amdgcn_exp2f(fabsf(t) * -1.44269504f);
      out[r] = (fmaxf(t, 0.0f) + __builtin_amdgcn_logf(1.0f + z) * 0.69314718f) * (0.01f * st.scale);
    } else if (EPI == NUDF_CH_NONE) {
      out[r] = v[r] * st.scale;
    } else if (EPI == NUDF_CH_UDFHEAD) {
      ch_udf_head(st.iparam, v[r], st.scale, out[r], out2[r]);
    } else if (EPI == NUDF_CH_RELU) {
      out[r] = fmaxf(v[r], 0.0f);
      out2[r] = out[r];                                       // optional mirror (hidden tap of the colour net)
    } else if (EPI == NUDF_CH_RELUADD) {
      out[r] = fmaxf(v[r] + x2[r], 0.0f);                     // skip layer: X2 = (other input part) x (its weight rows)
    } else if (EPI == NUDF_CH_SIGMOIDN) {
      // columns < iparam through a sigmoid (torch.sigmoid accuracy: libm exp), the rest raw
      out2[r] = v[r];
      out[r] = (col < st.iparam) ? 1.0f / (1.0f + expf(-v[r])) : 0.0f;
    } else if (EPI == NUDF_CH_MULMASK) {
      out[r] = (x1[r] > 0.0f) ? v[r] * st.scale : 0.0f;      // ReLU backward
    } else if (EPI == NUDF_CH_ADDMASK) {
      out[r] = (x1[r] > 0.0f) ? (v[r] + x2[r]) * st.scale : 0.0f;   // ReLU backward at an adjoint join
    } else {
      // softplus'(a) = s and 1 - s recovered from the stored activation (see ch_sp_derivs)
      // stored h = softplus100(a) / xscale  ->  1 - s = exp(-100 h xscale), s = softplus'(a) = sigmoid(100 a)
      const float x = 100.0f * st.xscale * x1[r];
      const float om = __builtin_amdgcn_exp2f(x * -1.44269504f);
      const float sg = 1.0f - om;
      if (EPI == NUDF_CH_MULSP) {
        const bool hid = st.iparam <= 0 || col < st.iparam;
        out[r] = hid ? v[r] * sg * st.scale : 0.0f;
        out2[r] = v[r] * st.scale;                            // embedding branch of the skip split
      } else if (EPI == NUDF_CH_TANGENT) {
        out[r] = v[r] * sg * st.scale;
        out2[r] = v[r] * x2[r] * 100.0f * om;
      } else {  // NUDF_CH_BWD
        if constexpr (has_x3) {
          // EX = (R / (s scale)) DA 100 (1 - s): R = (R W^T) s scale is divided back into the pre-activation tangent.  Where
          // s underflows to 0 both R and DA are exact zeros (each carries the factor s): the clamped reciprocal keeps the
          // product at 0 without a select (selects around transcendentals become per-element branches).
          const float f = 100.0f * om * __builtin_amdgcn_rcpf(fmaxf(sg * st.scale, 1e-30f));
          out[r] = v[r] * st.scale * sg + x2[r] * x3[r] * f;
        } else {
          out[r] = v[r] * st.scale * sg + x2[r];
        }
      }
    }
  }
  if (EPI == NUDF_CH_SIGMOIDN && st.row_w) {
    // compositing sum of this 32-point block inside the epilogue: lane (ln, h) holds 16 rows of column `col`
    const float* wr = st.row_w + grow0;
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += wr[CH_KOFF(r)] * out[r];
    sacc += __shfl_xor(sacc, 32);
    if (h == 0 && col < st.iparam && col < 4) st.row_sums[(size_t)(grow0 >> 5) * 4 + col] = sacc;
  }
#ifdef NUDF_X3_PROBE_NOSTORE   // timing probe (wrong results downstream): the stored-state arrays are not written
  if (EPI == NUDF_CH_SOFTPLUS || EPI == NUDF_CH_MULSP || EPI == NUDF_CH_TANGENT || EPI == NUDF_CH_BWD) {
    if (st.act_write) {
      float* ap = act + r0 * CH_LD + st.act_col0 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) ap[CH_KOFF(r) * CH_LD] = col_ok ? out[r] : 0.0f;
    }
    return tmax;
  }
#endif
  // ---- stores ----
  if (EPI == NUDF_CH_UDFHEAD) {
    if (col == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (st.C2) st.C2[grow0 + CH_KOFF(r)] = out[r];
        if (st.C1) st.C1[grow0 + CH_KOFF(r)] = out2[r];
      }
    }
    return tmax;
  }
  if (EPI == NUDF_CH_SIGMOIDN) {
    if (col_ok) {
      if (col < st.iparam) {
        if (st.C1) {
          const unsigned vo = grow0 * (unsigned)st.ldc1 + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) (st.C1 + (size_t)CH_KOFF(r) * st.ldc1)[vo] = out[r];
        }
        if (st.C2 && st.N <= st.iparam) {   // no raw columns: C2 mirrors the sigmoid outputs (view-branch input)
          const unsigned vo = grow0 * (unsigned)st.ldc2 + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) (st.C2 + (size_t)CH_KOFF(r) * st.ldc2)[vo] = out[r];
        }
      } else if (st.C2) {
        const unsigned vo = grow0 * (unsigned)st.ldc2 + (col - st.iparam);
#pragma unroll
        for (int r = 0; r < 16; ++r) (st.C2 + (size_t)CH_KOFF(r) * st.ldc2)[vo] = out2[r];
      }
    }
  } else if (col_ok) {
    if (st.C1) {      // NudfChain.absmax_out: the largest value this launch stores for a weight-gradient GEMM to read
      float omax = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) omax = __builtin_fmaxf(omax, __builtin_fmaxf(fabsf(out[r]), fabsf(out[r + 1])));
      tmax = __builtin_fmaxf(tmax, omax * tinv);
    }
    if (EPI == NUDF_CH_MULSP && st.iparam > 0 && col >= st.iparam) {
      if (st.C2) {
        const unsigned vo = grow0 * (unsigned)st.ldc2 + (col - st.iparam);
#pragma unroll
        for (int r = 0; r < 16; ++r) (st.C2 + (size_t)CH_KOFF(r) * st.ldc2)[vo] = out2[r] * tinv;
      }
    } else if (st.C1) {
      const unsigned vo = grow0 * (unsigned)st.ldc1 + col;
      if (S16 || (RT16 && (st.layout & NUDF_CH_P4_C1))) {
        ch_p4_store(st.C1, st.ldc1, (grow0 >> 2), (unsigned)col, out);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) (st.C1 + (size_t)CH_KOFF(r) * st.ldc1)[vo] = out[r] * tinv;
      }
    }
    if (EPI == NUDF_CH_TANGENT || (EPI == NUDF_CH_RELU && st.C2)) {
      const unsigned vo = grow0 * (unsigned)st.ldc2 + col;
      if (S16 && EPI == NUDF_CH_TANGENT) {
        ch_p4_store(st.C2, st.ldc2, (grow0 >> 2), (unsigned)col, out2);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) (st.C2 + (size_t)CH_KOFF(r) * st.ldc2)[vo] = out2[r] * tinv;
      }
    }
  }
  if (st.act_write) {
    if constexpr (T16) {
      unsigned short* ap = reinterpret_cast<unsigned short*>(act) + r0 * CH_LD16 + st.act_col0 + col;
      if (tile_bf) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ap[CH_KOFF(r) * CH_LD16] = ch_f2bf(col_ok ? out[r] : 0.0f);
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {      // rows CH_KOFF(r), CH_KOFF(r + 1) are consecutive: one packed conversion
          const unsigned w = ch_f2h2(col_ok ? out[r] : 0.0f, col_ok ? out[r + 1] : 0.0f);
          ap[CH_KOFF(r) * CH_LD16] = (unsigned short)(w & 0xffffu);
          ap[CH_KOFF(r + 1) * CH_LD16] = (unsigned short)(w >> 16);
        }
      }
    } else {
      float* ap = act + r0 * CH_LD + st.act_col0 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) ap[CH_KOFF(r) * CH_LD] = col_ok ? out[r] : 0.0f;
    }
  }
  return tmax;
}

// Epilogues with a SECOND stored operand (X2: tangent, adjoint, ReLU joins), full-height wave blocks: the tiles as
// straight-line code, the X2 values of tile t+1 requested BEFORE tile t is computed and stored.  The per-tile loop
// below asked for X2 at the top of each tile, i.e. after the previous tile's 16-32 stores: vmcnt is in order, so the
// wait for those loads also waited for the stores' acknowledgements -- one full HBM round trip per tile with nothing
// else in flight (X2 loads + stores were 10 % + 13 % of the tangent sweep, scripts/chain_timeline.py).  Accumulators
// and the prefetched X1 are read from their home registers (no per-tile copies), which pays for the second buffer.
template <int EPI, int NRT, int NCT>
__device__ __forceinline__ void ch_epilogue_seq(const NudfChainStep& st, float* act, int m0, int rt0, int ct0, int h,
                                                int ln, f32x16 (&acc)[2][2], float (&px1)[2][2][16], const float (&bpre)[2]) {
  float xb[2][16];
  auto issue = [&](float (&x)[16], int i, int j) {
    const int col = (ct0 + j) * 32 + ln;
    const unsigned vo = (unsigned)(m0 + (rt0 + i) * 32 + 4 * h) * (unsigned)st.ldx2 + (unsigned)((col < st.N) ? col : 0);
    if (st.X2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = (st.X2 + (size_t)CH_KOFF(r) * st.ldx2)[vo];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = 0.0f;
    }
  };
  constexpr int NTL = NRT * NCT;
  issue(xb[0], 0, 0);
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t + 1 < NTL) issue(xb[(t + 1) & 1], (t + 1) / NCT, (t + 1) % NCT);
    ch_epilogue_tile<EPI, true>(st, act, m0, rt0 + t / NCT, ct0 + t % NCT, h, ln, acc[t / NCT][t % NCT],
                                px1[t / NCT][t % NCT], false, xb[t & 1], &bpre[t % NCT]);
  }
}

// The same for the 16-bit stored state (config-5 mode: X1, X2 hold bf16): BOTH stored operands of tile t + 1 are requested
// before tile t is computed and stored.  The per-tile loop below loads them at the top of each tile -- after the previous
// tile's 16-32 two-byte stores, with nothing else in flight -- i.e. one exposed HBM round trip per tile and operand set:
// at config 5's shape the tangent sweep (two operands in, two arrays out) took 1.95 ms against 1.13 ms for the adjoint.
// Raw 16-bit values wait in registers and are widened at use.  The registers come from the X1 prefetch of the fp32 K
// loop, which the 16-bit instantiation does not use at all (see mlp_chain_kernel).
#ifndef NUDF_SEQ16
#define NUDF_SEQ16 1     // A/B build switch (scripts/build_variants.sh): 0 = per-tile loop for the 16-bit stored state
#endif
template <int EPI, int NRT, int NCT, bool T16 = false, bool X3 = false>
__device__ __forceinline__ void ch_epilogue_seq16(const NudfChainStep& st, float* act, int m0, int rt0, int ct0, int h,
                                                  int ln, f32x16 (&acc)[2][2], const float (&bpre)[2], bool tile_bf = false) {
  constexpr bool U1 = CH_USES_X1(EPI), U2 = CH_USES_X2(EPI), U3 = (EPI == NUDF_CH_BWD) && X3;
  uint2 r1[2][4], r2[2][4], r3[2][4];      // raw 4-point packs of tile t and t + 1
  auto issue = [&](uint2 (&a1)[4], uint2 (&a2)[4], uint2 (&a3)[4], int i, int j) {
    const int col = (ct0 + j) * 32 + ln;
    const unsigned colc = (unsigned)((col < st.N) ? col : 0);
    const unsigned q0 = (unsigned)(m0 + (rt0 + i) * 32 + 4 * h) >> 2;
    if (U1) ch_p4_load(st.X1, st.ldx1, q0, colc, a1);
    if (U2) {
      if (st.X2) {
        ch_p4_load(st.X2, st.ldx2, q0, colc, a2);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) a2[g] = uint2{0u, 0u};
      }
    }
    if (U3) ch_p4_load(st.X3, st.ldx3, q0, colc, a3);
  };
  constexpr int NTL = NRT * NCT;
  issue(r1[0], r2[0], r3[0], 0, 0);
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t + 1 < NTL) issue(r1[(t + 1) & 1], r2[(t + 1) & 1], r3[(t + 1) & 1], (t + 1) / NCT, (t + 1) % NCT);
    float x1[16], x2[16], x3[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (U1) ch_p4_widen(r1[t & 1][g], x1 + 4 * g);
      if (U2) ch_p4_widen(r2[t & 1][g], x2 + 4 * g);
      if (U3) ch_p4_widen(r3[t & 1][g], x3 + 4 * g);
    }
    if (!U1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x1[r] = 0.0f;
    }
    if (!U2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x2[r] = 0.0f;
    }
    if constexpr (U3)
      ch_epilogue_tile<EPI, true, true, false, T16, true, float[16]>(st, act, m0, rt0 + t / NCT, ct0 + t % NCT, h, ln,
                                                                     acc[t / NCT][t % NCT], x1, false, x2, &bpre[t % NCT], tile_bf, x3);
    else
      ch_epilogue_tile<EPI, true, true, false, T16>(st, act, m0, rt0 + t / NCT, ct0 + t % NCT, h, ln, acc[t / NCT][t % NCT], x1,
                                                    false, x2, &bpre[t % NCT], tile_bf);
  }
}

#ifndef NUDF_X3_EPI_AHEAD
#define NUDF_X3_EPI_AHEAD 2  // A/B build switch: tiles the epilogue's stored operands are requested ahead (1 = as the fp32 kernels)
#endif
// bf16x3 mode (fp32 stored state, no operand prefetch under the K loop: the split needs those registers): BOTH stored
// operands of tile t + 1 are requested before tile t is computed and stored, as in ch_epilogue_seq16.
template <int EPI, int NRT, int NCT, bool X3 = false>
__device__ __forceinline__ float ch_epilogue_seq32(const NudfChainStep& st, float* act, int m0, int rt0, int ct0, int h,
                                                  int ln, f32x16 (&acc)[2][2], const float (&bpre)[2], float tsig = 1.0f,
                                                  float tinv = 1.0f) {
  float smax = 0.0f;
  constexpr bool U1 = CH_USES_X1(EPI), U2 = CH_USES_X2(EPI), U3 = (EPI == NUDF_CH_BWD) && X3;
  // (three stored operands: one tile ahead -- the same 96 registers as two operands two tiles ahead)
  constexpr int AHEAD = U3 ? 1 : NUDF_X3_EPI_AHEAD;
  float xa[AHEAD + 1][16], xb[AHEAD + 1][16], xc[U3 ? AHEAD + 1 : 1][16];
  auto issue = [&](float (&a1)[16], float (&a2)[16], float (&a3)[16], int i, int j) {
    const int col = (ct0 + j) * 32 + ln;
    const unsigned colc = (unsigned)((col < st.N) ? col : 0);
    const unsigned row0 = (unsigned)(m0 + (rt0 + i) * 32 + 4 * h);
#ifdef NUDF_X3_PROBE_NOX      // timing probe (wrong results): the epilogues' stored operands cost nothing -- the upper bound
    if (U1) {                 // of what an LDS-DMA prefetch of them under the K loop could buy
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.01f * (float)(row0 & 7);
    }
    if (U2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[r] = 0.5f;
    }
    return;
#endif
    if (U1) {
      const unsigned vo = row0 * (unsigned)st.ldx1 + colc;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = (st.X1 + (size_t)CH_KOFF(r) * st.ldx1)[vo];
    }
    if (U2) {
      if (st.X2) {
        const unsigned vo = row0 * (unsigned)st.ldx2 + colc;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = (st.X2 + (size_t)CH_KOFF(r) * st.ldx2)[vo];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.0f;
      }
    }
    if (U3) {
      const unsigned vo = row0 * (unsigned)st.ldx3 + colc;
#pragma unroll
      for (int r = 0; r < 16; ++r) a3[r] = (st.X3 + (size_t)CH_KOFF(r) * st.ldx3)[vo];
    }
  };
  constexpr int NTL = NRT * NCT;
  // Stored operands requested AHEAD tiles in advance (ring of AHEAD + 1 register sets).  The K loop's operand registers
  // are dead here (this mode prefetches nothing under the K loop), so there is room for more than the one tile of the
  // fp32 kernels: per-wave timelines showed the tangent epilogue at 32 k cycles per layer against 25 k for the K loop --
  // one exposed HBM round trip per tile.  (All four tiles up front = 128 registers: 33 spilled.)
  constexpr int AH = (AHEAD < NTL) ? AHEAD : NTL - 1;
#pragma unroll
  for (int t = 0; t < AH; ++t) issue(xa[t % (AH + 1)], xb[t % (AH + 1)], xc[U3 ? t % (AH + 1) : 0], t / NCT, t % NCT);
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t + AH < NTL)
      issue(xa[(t + AH) % (AH + 1)], xb[(t + AH) % (AH + 1)], xc[U3 ? (t + AH) % (AH + 1) : 0], (t + AH) / NCT, (t + AH) % NCT);
    if constexpr (U3)
      smax = fmaxf(smax, ch_epilogue_tile<EPI, true, false, false, false, true, float[16]>(
                             st, act, m0, rt0 + t / NCT, ct0 + t % NCT, h, ln, acc[t / NCT][t % NCT], xa[t % (AH + 1)], false,
                             xb[t % (AH + 1)], &bpre[t % NCT], false, xc[t % (AH + 1)], tsig, tinv));
    else
      smax = fmaxf(smax, ch_epilogue_tile<EPI, true>(st, act, m0, rt0 + t / NCT, ct0 + t % NCT, h, ln, acc[t / NCT][t % NCT],
                                                     xa[t % (AH + 1)], false, xb[t % (AH + 1)], &bpre[t % NCT], false, ChNoX3(),
                                                     tsig, tinv));
  }
  return smax;
}

// NUDF_CH_BWD with THREE stored operands (NudfChainStep.X3: X1 = X[l], X2 = R[l], X3 = DA[l-1]; fp32 state, split modes), the
// wave's tiles as EIGHT-row halves: the three operands of half-tile u + 1 are requested before half-tile u is computed and
// stored -- 2 x 3 x 8 = 48 registers in flight instead of the 96 of whole tiles (which spilled 22 registers into a K loop
// whose every scratch access waits vmcnt(0)).  out = (acc + bias [+ rank-1 term]) scale s + X2 X3 100 (1 - s) / (s scale).
template <int NRT, int NCT>
__device__ __forceinline__ float ch_epilogue_bwd3(const NudfChainStep& st, float* act, int m0, int rt0, int ct0, int h, int ln,
                                                  f32x16 (&acc)[2][2], const float (&bpre)[2], float tsig = 1.0f,
                                                  float tinv = 1.0f) {
  float bmax = 0.0f;
  constexpr int NU = NRT * NCT * 2;
  float xa[2][8], xb[2][8], xc[2][8];
  auto coords = [&](int u, unsigned& row0, int& col, unsigned& colc) {
    const int t = u >> 1, i = t / NCT, j = t % NCT;
    col = (ct0 + j) * 32 + ln;
    colc = (unsigned)((col < st.N) ? col : 0);
    row0 = (unsigned)(m0 + (rt0 + i) * 32 + 4 * h + 16 * (u & 1));     // accumulator registers 8 (u & 1) .. + 7
  };
  auto issue = [&](float (&a1)[8], float (&a2)[8], float (&a3)[8], int u) {
    unsigned row0, colc;
    int col;
    coords(u, row0, col, colc);
    const unsigned v1 = row0 * (unsigned)st.ldx1 + colc, v2 = row0 * (unsigned)st.ldx2 + colc, v3 = row0 * (unsigned)st.ldx3 + colc;
#pragma unroll
    for (int r = 0; r < 8; ++r) a1[r] = (st.X1 + (size_t)CH_KOFF(r) * st.ldx1)[v1];
#pragma unroll
    for (int r = 0; r < 8; ++r) a2[r] = (st.X2 + (size_t)CH_KOFF(r) * st.ldx2)[v2];
#pragma unroll
    for (int r = 0; r < 8; ++r) a3[r] = (st.X3 + (size_t)CH_KOFF(r) * st.ldx3)[v3];
  };
  issue(xa[0], xb[0], xc[0], 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (u + 1 < NU) issue(xa[(u + 1) & 1], xb[(u + 1) & 1], xc[(u + 1) & 1], u + 1);
    const int t = u >> 1, i = t / NCT, j = t % NCT, hf = u & 1;
    unsigned row0, colc;
    int col;
    coords(u, row0, col, colc);
    const bool col_ok = col < st.N;
    const float bias = bpre[j];
    float r1c = 0.0f;
    if (st.r1_row) r1c = st.r1_col[colc];
    float out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = acc[i][j][8 * hf + r] + bias;
      if (st.r1_row) {
        const float rv = (st.r1_row + (size_t)CH_KOFF(r) * st.ldr1)[row0 * (unsigned)st.ldr1];
        bmax = fmaxf(bmax, fabsf(rv));
        v += (rv * tsig) * r1c;
      }
      const float x = 100.0f * st.xscale * xa[u & 1][r];
      const float om = __builtin_amdgcn_exp2f(x * -1.44269504f);
      const float sg = 1.0f - om;
      const float f = 100.0f * om * __builtin_amdgcn_rcpf(fmaxf(sg * st.scale, 1e-30f));
      out[r] = v * st.scale * sg + (xb[u & 1][r] * tsig) * xc[u & 1][r] * f;
    }
    if (col_ok && st.C1) {
      const unsigned vo = row0 * (unsigned)st.ldc1 + (unsigned)col;
#pragma unroll
      for (int r = 0; r < 8; ++r) (st.C1 + (size_t)CH_KOFF(r) * st.ldc1)[vo] = out[r] * tinv;
      float omax = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; r += 2) omax = __builtin_fmaxf(omax, __builtin_fmaxf(fabsf(out[r]), fabsf(out[r + 1])));
      bmax = __builtin_fmaxf(bmax, omax * tinv);
    }
    if (st.act_write) {
      float* ap = act + ((rt0 + i) * 32 + 4 * h + 16 * hf) * CH_LD + st.act_col0 + col;
#pragma unroll
      for (int r = 0; r < 8; ++r) ap[CH_KOFF(r) * CH_LD] = col_ok ? out[r] : 0.0f;
    }
  }
  return bmax;
}

// epilogue of one step for this wave's tiles: new activations -> LDS (+ HBM where a later sweep needs them)
template <int EPI, int MODE>
__device__ __forceinline__ float ch_epilogue(const NudfChain& p, const NudfChainStep& st, float* act, int m0, int rt0,
                                            int ct0, int nrt, int nct, int h, int ln, f32x16 (&acc)[2][2],
                                            const float (&px1)[2][2][16], const float (&bpre)[2], bool tile_bf = false,
                                            float tsig = 1.0f, float tinv = 1.0f) {
  float emax = 0.0f;
  constexpr bool ANY16 = MODE == 1 || MODE == 3 || MODE == 4, X3 = MODE == 2, T16 = MODE == 3 || MODE == 4;
  constexpr bool PF = CH_USES_X1(EPI);
  if constexpr (CH_USES_X2(EPI) && !T16) {
    if (!X3 && st.prec == 0 && nrt == 2 && !(ANY16 && (st.layout & NUDF_CH_STATE16))) {
      float(&x1)[2][2][16] = const_cast<float(&)[2][2][16]>(px1);
      if (nct == 2) ch_epilogue_seq<EPI, 2, 2>(st, act, m0, rt0, ct0, h, ln, acc, x1, bpre);
      else ch_epilogue_seq<EPI, 2, 1>(st, act, m0, rt0, ct0, h, ln, acc, x1, bpre);
      return emax;
    }
  }
  if constexpr (X3 && EPI == NUDF_CH_BWD) {
    if (st.X3) {      // second-order term formed from R and DA (NudfChainStep.X3): three stored operands, fp32 state
      if (nrt == 2 && nct == 2) return ch_epilogue_bwd3<2, 2>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
      else if (nrt == 2) return ch_epilogue_bwd3<2, 1>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
      else if (nct == 2) return ch_epilogue_bwd3<1, 2>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
      else return ch_epilogue_bwd3<1, 1>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
    }
  }
  if constexpr (X3 && CH_USES_X1(EPI)) {
    if (st.prec >= 3 && nrt * nct >= 2) {
      if (nrt == 2 && nct == 2) return ch_epilogue_seq32<EPI, 2, 2>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
      else if (nrt == 2) return ch_epilogue_seq32<EPI, 2, 1>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
      else return ch_epilogue_seq32<EPI, 1, 2>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tsig, tinv);
    }
  }
  if constexpr (ANY16 && (EPI == NUDF_CH_MULSP || EPI == NUDF_CH_TANGENT || EPI == NUDF_CH_BWD)) {
    if (NUDF_SEQ16 && (st.layout & NUDF_CH_STATE16) && nrt == 2) {
      if constexpr (EPI == NUDF_CH_BWD && MODE != 3) {     // (the dispatcher sends three-operand chains to MODE 4)
        if (st.X3) {
          if (nct == 2) ch_epilogue_seq16<EPI, 2, 2, T16, true>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tile_bf);
          else ch_epilogue_seq16<EPI, 2, 1, T16, true>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tile_bf);
          return emax;
        }
      }
      if (nct == 2) ch_epilogue_seq16<EPI, 2, 2, T16>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tile_bf);
      else ch_epilogue_seq16<EPI, 2, 1, T16>(st, act, m0, rt0, ct0, h, ln, acc, bpre, tile_bf);
      return emax;
    }
  }
  const int ntiles = nrt * nct;
#pragma unroll 1
  for (int t = 0; t < ntiles; ++t) {
    const int i = (nct == 2) ? (t >> 1) : t, j = (nct == 2) ? (t & 1) : 0;
    f32x16 a;
    float x1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x1[r] = 0.0f;
    switch (2 * i + j) {   // one copy of the tile body in the code; the accumulators are selected by moves
      case 0:
        a = acc[0][0];
        if (PF) {
#pragma unroll
          for (int r = 0; r < 16; ++r) x1[r] = px1[0][0][r];
        }
        break;
      case 1:
        a = acc[0][1];
        if (PF) {
#pragma unroll
          for (int r = 0; r < 16; ++r) x1[r] = px1[0][1][r];
        }
        break;
      case 2:
        a = acc[1][0];
        if (PF) {
#pragma unroll
          for (int r = 0; r < 16; ++r) x1[r] = px1[1][0][r];
        }
        break;
      default:
        a = acc[1][1];
        if (PF) {
#pragma unroll
          for (int r = 0; r < 16; ++r) x1[r] = px1[1][1][r];
        }
        break;
    }
    if constexpr (ANY16 && (EPI == NUDF_CH_SOFTPLUS || EPI == NUDF_CH_MULSP || EPI == NUDF_CH_TANGENT || EPI == NUDF_CH_BWD)) {
      if (st.layout & NUDF_CH_STATE16) {
        if constexpr (EPI == NUDF_CH_BWD && MODE != 3) {
          if (st.X3) {       // (ragged blocks of a wave: the three-operand form loads its operands inside the tile)
            ch_epilogue_tile<EPI, false, true, false, T16, true>(st, act, m0, rt0 + i, ct0 + j, h, ln, a, x1, true, nullptr, &bpre[j],
                                                                 tile_bf);
            continue;
          }
        }
        ch_epilogue_tile<EPI, false, true, false, T16>(st, act, m0, rt0 + i, ct0 + j, h, ln, a, x1, true, nullptr, &bpre[j], tile_bf);
        continue;
      }
    }
    if constexpr (EPI == NUDF_CH_BWD && MODE == 1) {       // (fp32 state inside a 16-bit chain)
      if (st.X3) {
        ch_epilogue_tile<EPI, false, false, false, T16, true>(st, act, m0, rt0 + i, ct0 + j, h, ln, a, x1, true, nullptr, &bpre[j],
                                                              tile_bf);
        continue;
      }
    }
    constexpr bool RT16 = ANY16 && (EPI == NUDF_CH_RELU || EPI == NUDF_CH_MULMASK || EPI == NUDF_CH_ADDMASK);
    emax = fmaxf(emax, ch_epilogue_tile<EPI, false, false, RT16, T16>(st, act, m0, rt0 + i, ct0 + j, h, ln, a, x1, ANY16 || X3, nullptr,
                                                                      &bpre[j], tile_bf, ChNoX3(), tsig, tinv));
  }
  return emax;
}

// ANY16: some step of the chain uses 16-bit MFMA operands (config-5 mode).  The fp32 chains get a kernel without any of
// the 16-bit code: its register pressure (conversions, raw-bf16 operand prefetch) would otherwise spill into them.
// MODE 2: some step runs the bf16x3 K loop (fp32 emulated on the bf16 pipe, fp32 stored state): its own instantiation for
// the same reason -- the split's registers next to the 16-bit state code spilled.
// MODE 3: every step of the chain uses the SAME 16-bit operand type (fp16: forward sweeps, bf16: backward sweeps) and the LDS
// tile holds that type (ChainSmem16, ch_mma16t): no conversion and half the LDS reads in the K loop, 37 KB per 64-point
// workgroup -> THREE workgroups per CU (168 VGPRs).  Results are bit-identical to MODE 1 (the same values are rounded to the
// same type, once in the epilogue instead of in every reading wave).
// MODE 4: the same 16-bit-tile kernel register-allocated for TWO workgroups per CU (201 VGPRs, no spills): the TANGENT
// sweeps -- two stored operands in, two arrays out per tile, 176 live registers in the epilogue -- spill 54 registers at the
// 168 of three workgroups and run 1115 instead of 872 us at config 5's shape; every other sweep is faster with three.
#ifndef NUDF_T16_WGS
#define NUDF_T16_WGS 3       // A/B build switch: workgroups per CU MODE 3 is register-allocated for
#endif
#ifndef NUDF_TS_LOG2
#define NUDF_TS_LOG2 (-6)    // NudfChain.tile_scale: the tile's largest seed is scaled into [2^NUDF_TS_LOG2, 2 x that)
#endif
template <int TM, int MODE>
__global__ __launch_bounds__(CH_THREADS, (MODE == 3) ? NUDF_T16_WGS : 2) void mlp_chain_kernel(NudfChain p_arg) {
  constexpr bool ANY16 = MODE == 1 || MODE == 3 || MODE == 4, X3 = MODE == 2, T16 = MODE == 3 || MODE == 4;
  // The descriptor is read where it lies, in the kernarg segment: the by-value parameter is otherwise a private copy that
  // the optimiser has to prove away, and once it fails (it did when the 16-bit epilogues grew) all 2 KB go to scratch and
  // every K loop pays vmcnt(0) for it.
  (void)p_arg;
  const NudfChain& p = *(const NudfChain*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int WMT = TM / 32;  // row tiles per wave in the wide layout
  using Smem = typename std::conditional<T16, ChainSmem16<TM>, ChainSmem<TM>>::type;
  __shared__ __attribute__((aligned(16))) Smem sm;
  // MODE 3: operand type of the tile = the (common) operand type of the chain's steps
  const bool tile_bf = T16 && p.step[0].prec == 2;
  const int tfmt = T16 ? (tile_bf ? 2 : 1) : 0;
  float* const act_f = reinterpret_cast<float*>(sm.act);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5, ln = lane & 31;
  const int m0 = blockIdx.x * TM;
  // NudfChain.absmax_out (split modes): the largest |value| this thread puts into arrays a weight-gradient GEMM will read
  float amax = 0.0f;
  // NudfChain.tile_scale (split modes): sigma and 1 / sigma of this tile (wave-uniform; 1 without the option)
  float tsig = 1.0f, tinv = 1.0f;
  const bool tscale = X3 && p.tile_scale != 0;
  // max over the workgroup of a per-thread value (all threads call it; two barriers)
  auto wg_max = [&](float v) -> float {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();                       // (a previous reduction's readers are done)
    if (lane == 0) sm.red[wave] = v;
    __syncthreads();
    float m = sm.red[0];
#pragma unroll
    for (int w = 1; w < CH_THREADS / 64; ++w) m = fmaxf(m, sm.red[w]);
    return m;
  };
  // sigma = 2^(-6 - floor(log2 m)): the tile's largest seed lands in [2^-6, 2^-5) -- 2^21 of growth below fp16's largest
  // number, and every element down to 2^-8 of that seed keeps the full 22 bits of the f16x2 split (below: 2^-36 absolute)
  auto set_sigma = [&](float m) {
    const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;
    const int se = 254 + NUDF_TS_LOG2 - (int)e;
    if (e != 0u && e != 255u && se >= 1 && se <= 254) {
      tsig = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readfirstlane(se) << 23);
      tinv = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readfirstlane(254 - se) << 23);
    }
  };
  // what the tile's seeds other than the initial tile can be: the rank-1 operands of the steps, the caller's bound
  auto other_seeds = [&]() -> float {
    float m = 0.0f;
    for (int i = 0; i < p.n_steps; ++i) {
      const NudfChainStep& s0 = p.step[i];
      if (s0.r1_row)
        for (int e = tid; e < TM; e += CH_THREADS) m = fmaxf(m, fabsf(s0.r1_row[(size_t)min(m0 + e, p.P - 1) * s0.ldr1]));
    }
    if (p.tile_amax_in && tid < TM / 32) m = fmaxf(m, p.tile_amax_in[(m0 >> 5) + tid]);
    return m;
  };

  // ---- tile initialisation ---------------------------------------------------------------------
  if (p.x) {
    for (int e = tid; e < TM * 3; e += CH_THREADS) {
      int r = m0 + e / 3;
      if (r > p.P - 1) r = p.P - 1;
      sm.xs[e] = p.x[(size_t)(r / p.x_div) * 3 + (e % 3)];   // x_div = samples per ray for per-ray directions
      sm.vs[e] = p.v ? p.v[(size_t)r * 3 + (e % 3)] : 0.0f;
      // (JVP encoding: every element of the initial tile is (2^k or 1) v in_scale times a sine / cosine, k < pe_L)
      if (X3 && p.v && p.init == NUDF_CH_INIT_POSENC)
        amax = fmaxf(amax, fabsf(sm.vs[e] * p.pe_in_scale) * (float)(1 << (p.pe_L > 0 ? p.pe_L - 1 : 0)));
    }
  }
  if constexpr (X3) {
    if (tscale && p.init == NUDF_CH_INIT_POSENC && p.v) {   // JVP encoding: the tangents are the seeds; `amax` is their bound so far
      set_sigma(wg_max(fmaxf(amax, other_seeds())));
      for (int e = tid; e < TM * 3; e += CH_THREADS) sm.vs[e] *= tsig;
    }
  }
  __syncthreads();
  if (p.init == NUDF_CH_INIT_LOAD) {
    const int k4 = p.k0 >> 2;
    for (int e = tid; e < TM * k4; e += CH_THREADS) {
      const int r = e / k4, c4 = e - r * k4;
      int gr = m0 + r;
      if (gr > p.P - 1) gr = p.P - 1;
      const f32x4 val = *reinterpret_cast<const f32x4*>(p.A0 + (size_t)gr * p.lda0 + c4 * 4);
      if constexpr (X3) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3]))));
      if constexpr (T16) {
        uint2 w;
        if (tile_bf) {
          w.x = (unsigned)ch_f2bf(val[0]) | ((unsigned)ch_f2bf(val[1]) << 16);
          w.y = (unsigned)ch_f2bf(val[2]) | ((unsigned)ch_f2bf(val[3]) << 16);
        } else {
          w.x = ch_f2h2(val[0], val[1]);
          w.y = ch_f2h2(val[2], val[3]);
        }
        *reinterpret_cast<uint2*>(sm.act + r * CH_LD16 + c4 * 4) = w;
      } else {
        *reinterpret_cast<f32x4*>(act_f + r * CH_LD + c4 * 4) = val;
      }
    }
    if constexpr (X3) {
      if (tscale) {          // the loaded tile is the seed: scale it in place (each thread the elements it wrote)
        set_sigma(wg_max(fmaxf(amax, other_seeds())));
        for (int e = tid; e < TM * k4; e += CH_THREADS) {
          const int r = e / k4, c4 = e - r * k4;
          f32x4* q = reinterpret_cast<f32x4*>(act_f + r * CH_LD + c4 * 4);
          f32x4 val = *q;
          val[0] *= tsig; val[1] *= tsig; val[2] *= tsig; val[3] *= tsig;
          *q = val;
        }
      }
    }
  } else if (p.init == NUDF_CH_INIT_POSENC) {
    ch_write_pe<TM>(sm, p, m0, 0, 1.0f, p.G0, p.ldg0, 0, p.k0, false, tfmt, tinv);
  } else if (p.init == NUDF_CH_INIT_SEED) {
    // da[r, c] = sign[r] * w_row0[c] * inv_scale * softplus'(.)   (reverse-sweep seed, fields.py:219-231)
    const int C = p.k0;
    if ((p.init_state16 & 3) == 3 && p.G0) {
      // 16-bit stored state on both sides (4-point packed, ch_p4_off): a work item is FOUR consecutive points of one column --
      // one 8-byte load of the stored activations, one 8-byte store of the seed (lanes = consecutive columns: 512 contiguous
      // bytes per wave instruction).  The element-wise form below moved the same 134 MB at config 5's size as 2-byte accesses
      // 8 bytes apart -- a quarter of every 32-byte sector used -- and cost the input-gradient sweep ~250 us against the
      // tangent sweep's identical steps (round 6, profiles/r06_bench_cfg5_1024x256_mixed16.json).  Buffers are row-padded to
      // the tile (nudf.h), so the quads of a ragged last tile exist; points >= P compute on clamped rows as everywhere.
      const uint2* A4 = reinterpret_cast<const uint2*>(p.A0);
      uint2* G4 = reinterpret_cast<uint2*>(p.G0);
      for (int e = tid; e < (TM / 4) * C; e += CH_THREADS) {
        const int q = e / C, c = e - q * C;
        const unsigned gq = (unsigned)(m0 >> 2) + (unsigned)q;
        float hst[4], val[4];
        ch_p4_widen(A4[(size_t)gq * p.lda0 + c], hst);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int gr = m0 + 4 * q + k;
          if (gr > p.P - 1) gr = p.P - 1;
          float sd, om;
          ch_sp_derivs(hst[k], p.seed_xscale, sd, om);
          val[k] = p.seed_sign[gr] * p.seed_wrow[c] * p.seed_scale * sd;
          const int r = 4 * q + k;
          if constexpr (T16) sm.act[r * CH_LD16 + c] = tile_bf ? ch_f2bf(val[k]) : ch_f2h(val[k]);
          else act_f[r * CH_LD + c] = val[k];
        }
        uint2 w;
        w.x = (unsigned)ch_f2bf(val[0]) | ((unsigned)ch_f2bf(val[1]) << 16);
        w.y = (unsigned)ch_f2bf(val[2]) | ((unsigned)ch_f2bf(val[3]) << 16);
        G4[(size_t)gq * p.ldg0 + c] = w;
      }
    } else
    for (int e = tid; e < TM * C; e += CH_THREADS) {
      const int r = e / C, c = e - r * C;
      int gr = m0 + r;
      const bool live = gr < p.P;
      if (!live) gr = p.P - 1;
      float s, om;
      const float hst = (p.init_state16 & 1) ? ch_bf2f(reinterpret_cast<const unsigned short*>(p.A0)[ch_p4_off(gr, c, p.lda0)])
                                             : p.A0[(size_t)gr * p.lda0 + c];
      ch_sp_derivs(hst, p.seed_xscale, s, om);
      const float val = p.seed_sign[gr] * p.seed_wrow[c] * p.seed_scale * s;
      if constexpr (T16) sm.act[r * CH_LD16 + c] = tile_bf ? ch_f2bf(val) : ch_f2h(val);
      else act_f[r * CH_LD + c] = val;
      if (p.G0 && live) {
        if (p.init_state16 & 2) reinterpret_cast<unsigned short*>(p.G0)[ch_p4_off(gr, c, p.ldg0)] = ch_f2bf(val);
        else p.G0[(size_t)gr * p.ldg0 + c] = val;
      }
    }
  }
  __syncthreads();

  // Workgroups that share a CU start in lock-step and would then always be in the same phase (K loop: both
  // want the MFMA pipe; epilogue: both want the VALU).  Delay the waves in odd hardware wave slots once, so
  // that one workgroup's epilogues run under the other's MFMA phases.  Speed only.
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID.wave_id
  if (gridDim.x > 256) {
    const unsigned k = (ANY16 || X3) ? (slot & 1u) * NUDF_STAGGER16 : ((TM == 64) ? (slot & 1u) * NUDF_STAGGER32 : (slot % 3u) * (NUDF_STAGGER32 ? 1u : 0u));
    for (unsigned d = 0; d < k; ++d) __builtin_amdgcn_s_sleep(127);
  }

#ifdef NUDF_X3_CU_STAGGER   // experiment (profiles/r05_chain_cu_stagger.txt): de-phase the CUs against each other so that the
  // epilogue phases (HBM) of one half of the chip fall into the K-loop phases (no HBM traffic) of the other half
  if (X3 && gridDim.x > 256) {
    const unsigned hw = __builtin_amdgcn_s_getreg((8 << 11) | (8 << 6) | 4);      // HW_REG_HW_ID bits [15:8]: cu_id, sh_id, se_id
#if NUDF_X3_CU_STAGGER > 0
    const unsigned late = hw & 1u;                                                // odd CUs
    for (int d = 0; d < (int)late * NUDF_X3_CU_STAGGER; ++d) __builtin_amdgcn_s_sleep(127);
#else
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    for (int d = 0; d < (int)(xcc & 1u) * (-NUDF_X3_CU_STAGGER); ++d) __builtin_amdgcn_s_sleep(127);
#endif
  }
#endif
  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * 4 + wave) * 64 : nullptr;
  if (dbg && lane == 0) {
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    dbg[1] = __builtin_amdgcn_s_memtime();
    dbg[62] = wall_clock64();      // 100 MHz reference: (s_memtime span) / (wall span) = the shader clock the sweep ran at
  }

  // ---- the layer chain ---------------------------------------------------------------------------
  for (int si = 0; si < p.n_steps; ++si) {
    const NudfChainStep& st = p.step[si];
    // The SIMD's issue arbiter prefers the OLDER of the two co-resident waves, so without help one workgroup of a
    // CU runs every layer at full MFMA rate and its partner only in the gaps (the kernel then ends with half the
    // wave slots idle for ~13 % of its time).  Alternate the priority layer by layer: speed only.
    if ((si + slot) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    const int G = st.K >> 3;                 // k groups of 8
    const int NT = (st.N + 31) >> 5;         // 32-column tiles
    const f32x4* __restrict__ Bp = reinterpret_cast<const f32x4*>(st.Bp);

    // tile ownership of this wave: nrt x nct tiles of 32x32 starting at (rt0, ct0)
    int rt0, ct0, nrt, nct;
    if (WMT == 2) {
      if (NT <= 2) { rt0 = wave >> 1; ct0 = wave & 1; nrt = 1; nct = (ct0 < NT) ? 1 : 0; }
      else if (NT <= 4) { rt0 = wave >> 1; ct0 = 2 * (wave & 1); nrt = 1; nct = min(2, max(0, NT - ct0)); }
      else { rt0 = 0; ct0 = 2 * wave; nrt = 2; nct = min(2, max(0, NT - ct0)); }
    } else {
      rt0 = 0; nrt = 1;
      if (NT <= 4) { ct0 = wave; nct = (ct0 < NT) ? 1 : 0; }
      else { ct0 = 2 * wave; nct = min(2, max(0, NT - ct0)); }
    }

    f32x16 acc[2][2];
    float px1[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][j][r] = 0.0f;
          px1[i][j][r] = 0.0f;
        }
    // this lane's bias values (column tiles ct0, ct0 + 1), requested now: they arrive under the K loop
    float bpre[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (ct0 + j) * 32 + ln;
      bpre[j] = st.bias ? st.bias[(col < st.N) ? col : 0] : 0.0f;
    }

    if constexpr (T16) {
      if (nct > 0) {     // (the dispatcher sends only chains whose steps all have the tile's operand type here)
        const unsigned short* arow16 = sm.act + (rt0 * 32 + ln) * CH_LD16 + 8 * h;
        const uint4* bp16 = reinterpret_cast<const uint4*>(st.Bp) + (size_t)ct0 * 64 + lane;
        const size_t bstride = (size_t)NT * 64;
        const int G16 = st.K >> 4;
        if (tile_bf) {
          if (nrt == 2 && nct == 2) ch_mma16t<2, 2, true>(arow16, bp16, bstride, G16, acc);
          else if (nrt == 2) ch_mma16t<2, 1, true>(arow16, bp16, bstride, G16, acc);
          else if (nct == 2) ch_mma16t<1, 2, true>(arow16, bp16, bstride, G16, acc);
          else ch_mma16t<1, 1, true>(arow16, bp16, bstride, G16, acc);
        } else {
          if (nrt == 2 && nct == 2) ch_mma16t<2, 2, false>(arow16, bp16, bstride, G16, acc);
          else if (nrt == 2) ch_mma16t<2, 1, false>(arow16, bp16, bstride, G16, acc);
          else if (nct == 2) ch_mma16t<1, 2, false>(arow16, bp16, bstride, G16, acc);
          else ch_mma16t<1, 1, false>(arow16, bp16, bstride, G16, acc);
        }
      }
    } else
    if (nct > 0) {
      const float* arow = act_f + (rt0 * 32 + ln) * CH_LD + 4 * h;
      const f32x4* bptr = Bp + (size_t)ct0 * 64 + lane;
      const size_t bstride = (size_t)NT * 64;  // float4 per k group
      // the 16-bit instantiation never prefetches X1 under the K loop (its conversions need the registers; the fp32 steps
      // of a 16-bit chain -- the abs-head column -- have no stored operand): px1 stays dead there, the epilogues load X1
      const bool pfx = !ANY16 && !X3 && CH_USES_X1(st.epi);
      ChPrefetch pf;
      pf.X1 = st.X1;
      pf.ldx1 = st.ldx1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = (ct0 + j) * 32 + ln;
          pf.vo[i][j] = (unsigned)(m0 + (rt0 + i) * 32 + 4 * h) * (unsigned)st.ldx1 + (unsigned)((col < st.N) ? col : 0);
        }
      if (X3 && st.prec == 4) {
        const float* arow16 = act_f + (rt0 * 32 + ln) * CH_LD + 8 * h;
        const uint4* bp2 = reinterpret_cast<const uint4*>(st.Bp) + (size_t)ct0 * 128 + lane;
        const size_t bstride2 = (size_t)NT * 128;
        const int G16 = st.K >> 4;
#ifdef NUDF_X2_PIPE64          // A/B: the split interleaved with the MFMAs on the 64-point tiles as well
        constexpr bool PIPE2 = true;
#else
        constexpr bool PIPE2 = (NUDF_X3_PIPE == 1) || (NUDF_X3_PIPE == 2 && TM == 32);
#endif
        if (nrt == 2 && nct == 2) ch_mma16x2<2, 2, PIPE2>(arow16, bp2, bstride2, G16, acc);
        else if (nrt == 2) ch_mma16x2<2, 1, PIPE2>(arow16, bp2, bstride2, G16, acc);
        else if (nct == 2) ch_mma16x2<1, 2, PIPE2>(arow16, bp2, bstride2, G16, acc);
        else ch_mma16x2<1, 1, PIPE2>(arow16, bp2, bstride2, G16, acc);
      } else if (X3 && st.prec == 3) {
        const float* arow16 = act_f + (rt0 * 32 + ln) * CH_LD + 8 * h;
        const uint4* bp3 = reinterpret_cast<const uint4*>(st.Bp) + (size_t)ct0 * 192 + lane;
        const size_t bstride3 = (size_t)NT * 192;
        const int G16 = st.K >> 4;
        constexpr bool PIPE3 = (NUDF_X3_PIPE == 1) || (NUDF_X3_PIPE == 2 && TM == 32);
#ifdef NUDF_X3_PROBE_COLSPLIT
        // probe of VERDICT r4's column split (profiles/r05_chain_colsplit_probe.txt): the wave's two column tiles as two
        // passes over K -- the K-loop shape a half-step of 128 columns would have (each activation split feeds 12 MFMAs
        // instead of 24), bit-identical results (same k order per output)
        if (nrt == 2 && nct == 2) {
          f32x16 t[2][2];
          t[0][0] = acc[0][0]; t[1][0] = acc[1][0];
          ch_mma16x3<2, 1, PIPE3>(arow16, bp3, bstride3, G16, t);
          acc[0][0] = t[0][0]; acc[1][0] = t[1][0];
          t[0][0] = acc[0][1]; t[1][0] = acc[1][1];
          ch_mma16x3<2, 1, PIPE3>(arow16, bp3 + 192, bstride3, G16, t);
          acc[0][1] = t[0][0]; acc[1][1] = t[1][0];
        } else
#endif
        if (nrt == 2 && nct == 2) ch_mma16x3<2, 2, PIPE3>(arow16, bp3, bstride3, G16, acc);
        else if (nrt == 2) ch_mma16x3<2, 1, PIPE3>(arow16, bp3, bstride3, G16, acc);
        else if (nct == 2) ch_mma16x3<1, 2, PIPE3>(arow16, bp3, bstride3, G16, acc);
        else ch_mma16x3<1, 1, PIPE3>(arow16, bp3, bstride3, G16, acc);
      } else if (ANY16 && st.prec != 0) {
        const float* arow16 = act_f + (rt0 * 32 + ln) * CH_LD + 8 * h;
        const uint4* bp16 = reinterpret_cast<const uint4*>(st.Bp) + (size_t)ct0 * 64 + lane;
        const int G16 = st.K >> 4;
        const bool bf = st.prec == 2;
        if (nrt == 2 && nct == 2) { if (bf) ch_mma16<2, 2, true>(arow16, bp16, bstride, G16, acc); else ch_mma16<2, 2, false>(arow16, bp16, bstride, G16, acc); }
        else if (nrt == 2) { if (bf) ch_mma16<2, 1, true>(arow16, bp16, bstride, G16, acc); else ch_mma16<2, 1, false>(arow16, bp16, bstride, G16, acc); }
        else if (nct == 2) { if (bf) ch_mma16<1, 2, true>(arow16, bp16, bstride, G16, acc); else ch_mma16<1, 2, false>(arow16, bp16, bstride, G16, acc); }
        else { if (bf) ch_mma16<1, 1, true>(arow16, bp16, bstride, G16, acc); else ch_mma16<1, 1, false>(arow16, bp16, bstride, G16, acc); }
      } else if (pfx) {
        if (nrt == 2 && nct == 2) ch_mma<2, 2, true>(arow, bptr, bstride, G, acc, pf, px1);
        else if (nrt == 2) ch_mma<2, 1, true>(arow, bptr, bstride, G, acc, pf, px1);
        else if (nct == 2) ch_mma<1, 2, true>(arow, bptr, bstride, G, acc, pf, px1);
        else ch_mma<1, 1, true>(arow, bptr, bstride, G, acc, pf, px1);
      } else {
        if (nrt == 2 && nct == 2) ch_mma<2, 2, false>(arow, bptr, bstride, G, acc, pf, px1);
        else if (nrt == 2) ch_mma<2, 1, false>(arow, bptr, bstride, G, acc, pf, px1);
        else if (nct == 2) ch_mma<1, 2, false>(arow, bptr, bstride, G, acc, pf, px1);
        else ch_mma<1, 1, false>(arow, bptr, bstride, G, acc, pf, px1);
      }
    }
    if (dbg && lane == 0) dbg[2 + 4 * si] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // every wave is done reading the activation tile
    if (dbg && lane == 0) dbg[3 + 4 * si] = __builtin_amdgcn_s_memtime();

    if (nct > 0) {
      switch (st.epi) {
        case NUDF_CH_SOFTPLUS: ch_epilogue<NUDF_CH_SOFTPLUS, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); break;
        case NUDF_CH_NONE: { const float em = ch_epilogue<NUDF_CH_NONE, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); if constexpr (X3) amax = fmaxf(amax, em); } break;
        case NUDF_CH_MULSP: { const float em = ch_epilogue<NUDF_CH_MULSP, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); if constexpr (X3) amax = fmaxf(amax, em); } break;
        case NUDF_CH_TANGENT:     // (MODE 3 never runs a TANGENT chain: the dispatcher sends those to MODE 4)
          if constexpr (MODE != 3) ch_epilogue<NUDF_CH_TANGENT, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv);
          break;
        case NUDF_CH_BWD: { const float em = ch_epilogue<NUDF_CH_BWD, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); if constexpr (X3) amax = fmaxf(amax, em); } break;
        case NUDF_CH_RELU: ch_epilogue<NUDF_CH_RELU, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); break;
        case NUDF_CH_SIGMOIDN: ch_epilogue<NUDF_CH_SIGMOIDN, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); break;
        case NUDF_CH_MULMASK: { const float em = ch_epilogue<NUDF_CH_MULMASK, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); if constexpr (X3) amax = fmaxf(amax, em); } break;
        case NUDF_CH_ADDMASK: { const float em = ch_epilogue<NUDF_CH_ADDMASK, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); if constexpr (X3) amax = fmaxf(amax, em); } break;
        case NUDF_CH_RELUADD: ch_epilogue<NUDF_CH_RELUADD, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); break;
        default: ch_epilogue<NUDF_CH_UDFHEAD, MODE>(p, st, act_f, m0, rt0, ct0, nrt, nct, h, ln, acc, px1, bpre, tile_bf, tsig, tinv); break;
      }
    }
    if (dbg && lane == 0) dbg[4 + 4 * si] = __builtin_amdgcn_s_memtime();
    if (st.pe_tail_col >= 0) {
      __syncthreads();
      // zero up to the next multiple of 16 columns: the K padding of the step that consumes [.. | PE] must multiply
      // finite zeros (columns never written before hold arbitrary LDS contents)
      const int pe_end = st.pe_tail_col + 3 * (2 * p.pe_L + 1);
      ch_write_pe<TM>(sm, p, m0, st.pe_tail_col, st.pe_tail_scale, st.pe_dst, st.ld_pe, st.pe_tail_col,
                      min((pe_end + 15) & ~15, 288), (st.layout & NUDF_CH_STATE16) != 0, tfmt, tinv);
    }
    __syncthreads();
    if (dbg && lane == 0) dbg[5 + 4 * si] = __builtin_amdgcn_s_memtime();
  }
  if (dbg && lane == 0) dbg[63] = wall_clock64();
  if constexpr (X3) {
    if (p.tile_amax_out) {   // per 32 points: the largest |value| the launch stored for these rows (a later sweep's tile_amax_in)
      const float m = wg_max(amax);
      if (tid < TM / 32) p.tile_amax_out[(m0 >> 5) + tid] = m;
    }
    if (p.absmax_out) {      // non-negative floats order like their bit patterns: one unsigned atomic max per wave
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
      if (lane == 0 && amax > 0.0f && amax < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(p.absmax_out), __builtin_bit_cast(unsigned, amax));
    }
  }
}

int nudf_chain_rows_class(const NudfChain& p, bool allow_blocked = false);  // mlp_chain_rows.hip
int nudf_mlp_chain_rows_launch(const NudfChain& p, int cls, hipStream_t st);
int nudf_mlp_chain_tq_launch(const NudfChain& p, int cls, hipStream_t st, int force_pair = 0);
int nudf_chain_pair_mode();

// NUDF_CHAIN_ROWS=1 lets large launches choose the wave-private kernel on their own (measured in round 2: equal to
// the workgroup-shared tiles on the forward sweeps, 10-20 % slower on the sweeps that stream stored state, see
// DESIGN.md section 4.1b); tile_rows = 128 requests it explicitly.  Read once.
static bool nudf_chain_rows_auto() {
  static const int on = [] {
    const char* e = getenv("NUDF_CHAIN_ROWS");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return on != 0;
}

// The transposed-product form of the 64-point tile (mlp_chain_rows.hip, tile_rows = 66: 16-byte epilogue accesses).
// Measured at 65 536 points against mlp_chain_kernel<64> (profiles/r02_chain_timeline.txt): per-wave time -8 % / -6 % on
// the UDF forward / input-gradient sweeps, +9 % / +5 % on the tangent / adjoint sweeps (two stored operands and two
// outputs per layer: with row-major buffers each 16-byte-per-lane access touches 32 different 128-byte lines); with the
// BLOCKED layout of nudf.h addressed instead (timing only) -8 / -10 / -7 / -4 %.  Over a whole train step the
// forward / gradient gain is inside the noise (3.95 vs 3.98 ms of chain time), so the default stays mlp_chain_kernel:
// NUDF_CHAIN_QUAD=1 uses the transposed form for launches with at most one stored operand, =2 for every launch.
static int nudf_chain_quad_mode() {
  static const int mode = [] {
    const char* e = getenv("NUDF_CHAIN_QUAD");
    return e ? atoi(e) : 0;
  }();
  return mode;
}

static int g_chain_t16 = -1;     // NUDF_CHAIN_T16 / nudf_set_chain_t16
static bool nudf_chain_t16_enabled() {
  if (g_chain_t16 < 0) {
    const char* e = getenv("NUDF_CHAIN_T16");
    g_chain_t16 = (e && e[0] == '0') ? 0 : 1;
  }
  return g_chain_t16 != 0;
}
extern "C" int nudf_set_chain_t16(int on) {
  const int old = nudf_chain_t16_enabled() ? 1 : 0;
  g_chain_t16 = on ? 1 : 0;
  return old;
}

extern "C" int nudf_mlp_chain(const NudfChain* args, void* stream) {
  const NudfChain& p = *args;
  if (p.P <= 0 || p.n_steps <= 0) return 0;
  bool bad = p.n_steps > NUDF_CH_MAX_STEPS || (p.k0 & 3) || p.k0 > 288 || p.x_div < 1;
  for (int i = 0; i < p.n_steps && !bad; ++i) {
    const NudfChainStep& s = p.step[i];
    bad = (s.K & 15) || s.K <= 0 || s.K > 288 || s.N <= 0 || s.N > 256 || (((uintptr_t)s.Bp) & 15) ||
          (s.act_write && s.act_col0 + ((s.N + 31) / 32) * 32 > 288) || s.prec < 0 || s.prec > 4 ||
          ((s.layout & NUDF_CH_STATE16) && s.epi != NUDF_CH_SOFTPLUS && s.epi != NUDF_CH_MULSP &&
           s.epi != NUDF_CH_TANGENT && s.epi != NUDF_CH_BWD) ||
          ((s.layout & NUDF_CH_P4_X1) && s.epi != NUDF_CH_MULMASK && s.epi != NUDF_CH_ADDMASK) ||
          ((s.layout & NUDF_CH_P4_C1) && s.epi != NUDF_CH_RELU && s.epi != NUDF_CH_MULMASK && s.epi != NUDF_CH_ADDMASK) ||
          ((s.layout & (NUDF_CH_P4_X1 | NUDF_CH_P4_C1)) && s.prec == 0) ||
          (s.X3 && (s.epi != NUDF_CH_BWD || s.prec == 0 || !s.X2 || !s.X1 || p.tile_rows == 66 || p.tile_rows == 128 ||
                    p.tile_rows == 130));
  }
  if (bad) {
    nudf_set_error("nudf_mlp_chain: K%16, K<=288, N<=256, x_div>=1, 16-byte aligned packed weights required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipStream_t st = (hipStream_t)stream;
  bool any16 = false, any3 = false;
  for (int i = 0; i < p.n_steps; ++i) {
    any3 = any3 || p.step[i].prec >= 3;        // the split modes (bf16x3, f16x2) share the MODE 2 instantiation
    any16 = any16 || (p.step[i].prec != 0 && p.step[i].prec < 3) ||
            (p.step[i].layout & (NUDF_CH_STATE16 | NUDF_CH_P4_X1 | NUDF_CH_P4_C1));
  }
  if (any3 && any16) {
    nudf_set_error("nudf_mlp_chain: split steps (prec 3 / 4) do not mix with 16-bit steps / 16-bit stored state", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  if (p.tile_scale || p.tile_amax_in || p.tile_amax_out) {      // include/nudf.h: NudfChain.tile_scale
    bool ok = true;      // (every step in a split mode: the launch runs on mlp_chain_kernel<TM, 2> whatever tile_rows asks for)
    if (p.tile_scale)
      ok = ok && (p.init == NUDF_CH_INIT_LOAD || (p.init == NUDF_CH_INIT_POSENC && p.v && p.pe_jvp));
    for (int i = 0; i < p.n_steps && ok; ++i) {
      const NudfChainStep& s = p.step[i];
      ok = s.prec >= 3;
      if (p.tile_scale)
        ok = ok && !s.bias && (s.pe_tail_col < 0 || p.pe_jvp) &&
             (s.epi == NUDF_CH_NONE || s.epi == NUDF_CH_MULSP || s.epi == NUDF_CH_TANGENT || s.epi == NUDF_CH_BWD ||
              s.epi == NUDF_CH_MULMASK || s.epi == NUDF_CH_ADDMASK);
    }
    if (!ok) {
      nudf_set_error("nudf_mlp_chain: tile_scale / tile_amax_* need a linear sweep in a split mode (INIT_LOAD or the JVP encoding; "
                     "NONE / MULSP / TANGENT / BWD / MULMASK / ADDMASK steps of prec 3 / 4 without bias)", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
    }
  }
  // Large launches: wave-private 32-point tiles (mlp_chain_rows.hip), one free-running wave per SIMD.  A "round" of
  // that kernel is 1024 waves = 32 768 points, so it is chosen when the last round is at least ~80 % full; the
  // up-sampling rounds (5-8 k points) and awkward sizes keep the workgroup-shared tiles below.
  bool roww = false;
  for (int i = 0; i < p.n_steps; ++i) {
    const NudfChainStep& s = p.step[i];
    if (!s.row_w) continue;
    roww = true;
    if (s.epi != NUDF_CH_SIGMOIDN || !s.row_sums || p.tile_rows == 66 || p.tile_rows == 128 || p.tile_rows == 130) {
      nudf_set_error("nudf_mlp_chain: row_w / row_sums belong to SIGMOIDN steps of the workgroup-shared kernel", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
    }
  }
  bool rows_ok = !roww && (p.tile_rows == 128 || p.tile_rows == 0);
  if (rows_ok && p.tile_rows == 0) {
    const long long round = 1024LL * 32, rounds = (p.P + round - 1) / round;
    rows_ok = nudf_chain_rows_auto() && p.P >= 24576 && (double)p.P >= 0.8 * (double)(rounds * round);
  }
  if (rows_ok) {
    const int cls = nudf_chain_rows_class(p);
    if (cls >= 0) return nudf_mlp_chain_rows_launch(p, cls, st);
    // contract of the wave-private kernel not met (16-bit operands, unaligned row buffers): workgroup-shared tiles
  }
  bool blocked = (p.init_state16 & 12) != 0;
  for (int i = 0; i < p.n_steps; ++i) blocked = blocked || (p.step[i].layout & 31) != 0;
  if (blocked) {   // only the transposed-product shared tile addresses the blocked layout
    const int cls = nudf_chain_rows_class(p, true);
    if (cls < 0 || (p.tile_rows != 66 && p.tile_rows != 130 && p.tile_rows != 0)) {
      nudf_set_error("nudf_mlp_chain: blocked-layout buffers need the transposed-product kernel (tile_rows 0 / 66, fp32 "
                     "steps, 16-byte aligned rows)", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
    }
    return nudf_mlp_chain_tq_launch(p, cls, st, p.tile_rows == 130);
  }
  if (p.tile_rows == 130) {     // paired tiles requested explicitly (tests, A/B): any size
    const int cls = nudf_chain_rows_class(p);
    if (cls >= 0) return nudf_mlp_chain_tq_launch(p, cls, st, 1);
  }
  if (!roww && (p.tile_rows == 66 || (p.tile_rows == 0 && p.P > 256 * 64 && nudf_chain_quad_mode() > 0))) {
    const int cls = nudf_chain_rows_class(p);
    if (cls >= 0 && (p.tile_rows == 66 || cls <= 1 || nudf_chain_quad_mode() >= 2)) return nudf_mlp_chain_tq_launch(p, cls, st);
  }
  // NUDF_CHAIN_PAIR=2: paired tiles (mlp_chain_pair_kernel, reached through the transposed-product launcher) for every
  // fp32 launch of at least 32 768 points that meets that kernel's contract -- a measured counter-example, off by default
  if (!roww && p.tile_rows == 0 && p.P >= 32768 && !any16 && !any3 && nudf_chain_pair_mode() >= 2) {
    const int cls = nudf_chain_rows_class(p);
    if (cls >= 0) return nudf_mlp_chain_tq_launch(p, cls, st);
  }
  // 16-bit mode, 64-point tiles: the 16-bit-TILE kernel (MODE 3) when every step contracts in the same 16-bit type (the
  // tile holds ONE type): the UDF network's four sweeps with the 16-bit head, the colour / NeRF chains.  NUDF_CHAIN_T16=0
  // keeps the fp32-tile kernel (A/B; the two are bit-identical).
  bool t16 = any16 && !roww && nudf_chain_t16_enabled();
  bool t16_tangent = false;
  for (int i = 0; i < p.n_steps && t16; ++i) {
    t16 = (p.step[i].prec == 1 || p.step[i].prec == 2) && p.step[i].prec == p.step[0].prec;
    t16_tangent = t16_tangent || p.step[i].epi == NUDF_CH_TANGENT || p.step[i].X3 != nullptr;   // (MODE 4: room for 3 operands)
  }
  // small launches: 32-point tiles fill the 256 CUs sooner (up-sampling rounds are 5-8 k points)
  if (p.tile_rows == 32 || (p.tile_rows != 64 && p.P <= 256 * 64)) {
    if (any3) hipLaunchKernelGGL((mlp_chain_kernel<32, 2>), dim3((p.P + 31) / 32), dim3(CH_THREADS), 0, st, p);
    else if (any16) hipLaunchKernelGGL((mlp_chain_kernel<32, 1>), dim3((p.P + 31) / 32), dim3(CH_THREADS), 0, st, p);
    else hipLaunchKernelGGL((mlp_chain_kernel<32, 0>), dim3((p.P + 31) / 32), dim3(CH_THREADS), 0, st, p);
  } else {
    if (any3) hipLaunchKernelGGL((mlp_chain_kernel<64, 2>), dim3((p.P + 63) / 64), dim3(CH_THREADS), 0, st, p);
    else if (t16 && t16_tangent) hipLaunchKernelGGL((mlp_chain_kernel<64, 4>), dim3((p.P + 63) / 64), dim3(CH_THREADS), 0, st, p);
    else if (t16) hipLaunchKernelGGL((mlp_chain_kernel<64, 3>), dim3((p.P + 63) / 64), dim3(CH_THREADS), 0, st, p);
    else if (any16) hipLaunchKernelGGL((mlp_chain_kernel<64, 1>), dim3((p.P + 63) / 64), dim3(CH_THREADS), 0, st, p);
    else hipLaunchKernelGGL((mlp_chain_kernel<64, 0>), dim3((p.P + 63) / 64), dim3(CH_THREADS), 0, st, p);
  }
  NUDF_CHECK_LAUNCH("nudf_mlp_chain");
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// weights -> MFMA-B fragment order.  B is the [K, N] row-major GEMM operand (W^T for forward sweeps, W for
// reverse sweeps; row stride ldb).  out[((g*NT + T)*64 + lane)*4 + j] = B[8g + 4(lane>>5) + j][32T + (lane&31)],
// zero outside K x N.  Kpad = roundup(K, 16).
// ---------------------------------------------------------------------------------------------------
__global__ void pack_frag_kernel(const float* __restrict__ B, int ldb, int K, int N, int NT, int total,
                                 float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3, lane = (idx >> 2) & 63, gt = idx >> 8;
  const int T = gt % NT, g = gt / NT;
  const int k = 8 * g + 4 * (lane >> 5) + j, n = 32 * T + (lane & 31);
  out[idx] = (k < K && n < N) ? B[(size_t)k * ldb + n] : 0.0f;
}

extern "C" int nudf_pack_frag(const float* B, int ldb, int K, int N, float* out, void* stream) {
  if (K <= 0 || N <= 0) return 0;
  const int G = 2 * ((K + 15) / 16), NT = (N + 31) / 32;
  const int total = G * NT * 256;
  hipLaunchKernelGGL(pack_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, ldb, K, N, NT,
                     total, out);
  NUDF_CHECK_LAUNCH("nudf_pack_frag");
  return 0;
}
