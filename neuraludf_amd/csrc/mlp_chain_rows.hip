// Fused MLP layer chains, wave-private layout (gfx950): every WAVE owns a tile of 32 points for the whole
// sweep -- all <= 256 output features of every layer -- so no wave ever waits for another one: the kernel has no
// barrier at all, each of the CU's 4 SIMDs runs one free-running wave (a 128-point workgroup = 4 such waves,
// 150 KB of LDS).  Same step tables (NudfChain), same arithmetic and same fp32 summation order as the
// workgroup-shared kernel of mlp_chain.hip, which stays the path for launches too small to fill 1024 waves.
//
// Differences that matter on CDNA4:
//  * the product is computed TRANSPOSED: v_mfma_f32_32x32x2_f32(weights, activations) -> the accumulator lane owns
//    ONE POINT and its 16 registers are 4 groups of 4 CONSECUTIVE FEATURES, so the epilogue moves 16 bytes per
//    lane per instruction everywhere: ds_write_b128 into the LDS tile (conflict-free at row stride 292), and
//    global_load/store_dwordx4 for the stored-state operands / outputs (4 instead of 16 VMEM instructions per
//    32x32 tile and operand -- what lets a wave keep four tiles of two operands in flight inside the 6-bit vmcnt);
//  * one ds_read_b128 of the activation tile feeds 4 * NT MFMAs (NT = feature tiles, up to 8) instead of 8;
//  * nothing covers a wave's latencies but the wave itself (one wave per SIMD), so every epilogue operand is
//    requested ahead of its use: the stored-state operands (X1, X2) of feature tiles 0..3 under the last two k
//    groups of the K loop, those of tile t + 4 as soon as tile t has been consumed; the bias of step s + 1 is staged
//    through LDS during step s and enters as the accumulators' initial value (bias-first summation: results differ
//    from the workgroup-shared kernel by the rounding of that one addition);
//  * the 4 waves of a workgroup read the same weight fragments; re-aligning them with a barrier per step
//    (CHR_STEP_SYNC) so that three of the four reads hit the CU's vector L1 measured 3 % SLOWER and is off.
#include "mlp_chain_shared.h"
#include <stdio.h>
#include <stdlib.h>

// A/B build switches of the transposed-product shared-tile kernel (scripts/build_variants.sh): alternating wave priority
// per layer (1, shipping) / none (0) / fixed by wave slot (2); the one-off start-up delay of odd wave slots (1 / 0)
#ifndef NUDF_TQ_SPREAD
#define NUDF_TQ_SPREAD 1     // K loop: next group's operand requests spread over the current group's MFMA quarters (A/B: 0)
#endif
#ifndef NUDF_TQ_PRIO
#define NUDF_TQ_PRIO 1
#endif
#ifndef NUDF_TQ_SLEEP
#define NUDF_TQ_SLEEP 1
#endif
#define CHR_WAVES 4
#ifndef CHR_STEP_SYNC
#define CHR_STEP_SYNC 0      // 1: barrier at the top of every step (L1 sharing of the weight stream): measured -3 %
#endif
#define CHR_ROWS 32

struct ChainRowsSmem {
  float act[CHR_WAVES][CHR_ROWS * CH_LD];   // 149 504 B
  float bias[CHR_WAVES][2][256];            //   8 192 B (double-buffered: step s reads [s & 1], stages [~s & 1])
  float xs[CHR_WAVES][CHR_ROWS * 3];
  float vs[CHR_WAVES][CHR_ROWS * 3];
};

// LDS traffic of one wave is processed in program order; only the compiler has to be told not to move a tile read
// above the tile writes of other lanes
__device__ __forceinline__ void chr_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void chr_write_pe(float* act, const float* xs, const float* vs, const NudfChain& p, int lane,
                                             int m0, int col0, float scale, float* gdst, int ldg, int gcol0, int zero_to) {
  const int E = 3 * (2 * p.pe_L + 1);
  for (int e = lane; e < CHR_ROWS * E; e += 64) {
    const int r = e / E, c = e - r * E;
    const float val = ch_pe(xs + r * 3, vs + r * 3, c, p.pe_L, p.pe_in_scale, p.pe_jvp) * scale;
    act[r * CH_LD + col0 + c] = val;
    if (gdst && (m0 + r) < p.P) gdst[(size_t)(m0 + r) * ldg + gcol0 + c] = val;
  }
  const int npad = zero_to - (col0 + E);
  if (npad > 0)
    for (int e = lane; e < CHR_ROWS * npad; e += 64) {
      const int r = e / npad, c = e - r * npad;
      act[r * CH_LD + col0 + E + c] = 0.0f;
    }
}

// Launch-time contract of this kernel (checked by nudf_chain_rows_supported, else the workgroup-shared kernel
// runs): every [P, ld] buffer that is accessed quad-wise -- X1, X2, C1 (except the UDFHEAD / SIGMOIDN heads) and the
// TANGENT / RELU mirrors in C2 -- is 16-byte aligned with ld % 4 == 0, and act_col0 % 4 == 0 (except SIGMOIDN).

// Values the epilogue needs per step, pinned in SGPRs: left to itself the compiler re-reads NudfChainStep fields
// from the kernel-argument segment inside the tile loop (an s_load + lgkmcnt(0) wait per quad).
template <class T>
__device__ __forceinline__ T chr_pin(T x) {
  asm volatile("" : "+s"(x));
  return x;
}

// global-address-space views of the step's buffers: a pointer that went through chr_pin is a generic pointer to
// the compiler (flat_load: counted by lgkmcnt as well, so it would tie the LDS waits to HBM latency)
typedef const float __attribute__((address_space(1)))* chr_gcp;
typedef float __attribute__((address_space(1)))* chr_gp;

struct ChrStep {
  unsigned row, x1, x2, c1, c2;          // per lane: row and the offset of its first feature in the step's buffers
  unsigned m1, m2, mc1, mc2;             // offset of feature f = base + f * m: 1 (row-major) or 32 (blocked layout,
                                         // f % 4 == 0: quads of a 32-row block are 512 contiguous bytes, see nudf.h)
  chr_gcp X1, X2;
  chr_gp C1, C2;
  int N, nq, iparam, act_col0, act_write, lim1, lim2;
  float scale, xscale;
};

// Features f0..f0+3 of one point of a stored [P, ld] operand (accumulator layout: lane = point, 4 consecutive
// registers = 4 consecutive features).  Columns are clamped into [0, N): the values of features >= N are never
// used, so the loads carry no predicate.
typedef const f32x4 __attribute__((address_space(1)))* chr_gcp4;
typedef f32x4 __attribute__((address_space(1)))* chr_gp4;
__device__ __forceinline__ f32x4 chr_load_quad(chr_gcp X, unsigned base, int f0, int nq, unsigned mult = 1u) {
  return *(chr_gcp4)(X + (base + (unsigned)min(f0, nq) * mult));
}

// features [lo, hi) -> C[row, f - shift], element-wise (heads, split outputs)
__device__ __forceinline__ void chr_store_range(chr_gp C, unsigned base, int f0, int lo, int hi, int shift,
                                                const f32x4& v, unsigned mult = 1u) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (f0 + i >= lo && f0 + i < hi) {
      const unsigned c = (unsigned)(f0 + i - shift);
      C[base + (mult == 1u ? c : (c & ~3u) * 32u + (c & 3u))] = v[i];
    }
}

// Epilogue of one [32 points x 32 features] accumulator tile (bias already inside the accumulator), streamed as four
// quads of consecutive features: compute -> re-request this quad of the operand window for tile t + CHR_WIN -> store.
// Features >= N of the last tile need no masking: their accumulators are exactly 0 (zero-padded weight fragments,
// zero staged bias), every epilogue maps that to a FINITE value (see the |h| below), and such values only ever reach
// pad columns --
// of the LDS tile, where the next step's zero weight rows meet them, and of the [P, ld] buffers, which no consumer
// reads (operand loads clamp their columns into [0, N)).
template <int EPI>
__device__ __forceinline__ void chr_epi_tile(const NudfChainStep& st, float* act, const ChrStep& cs, int t, int h, int ln,
                                             const f32x16& a, float (&w1)[16], float (&w2)[16], bool reload, int t_next,
                                             const float* bias_lds = nullptr) {
  constexpr bool U1 = CH_USES_X1(EPI), U2 = CH_USES_X2(EPI);
  const int fb = 32 * t + 4 * h;
  const int N = cs.N;
  float r1 = 0.0f;
  // rank-1 term (column 0 of the abs head in the UDF adjoint sweep, the density head's adjoint in the NeRF's)
  const bool has_r1 = (EPI == NUDF_CH_BWD || EPI == NUDF_CH_MULMASK) && st.r1_row != nullptr;
  if (has_r1) r1 = st.r1_row[(size_t)cs.row * st.ldr1];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f0 = fb + 8 * q;
    f32x4 o, o2;
    f32x4 bq = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias_lds) bq = *reinterpret_cast<const f32x4*>(bias_lds + f0);   // bias added AFTER the products (shared-tile form)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      float v = a[r] + bq[i];
      if (has_r1) v += r1 * st.r1_col[min(f0 + i, N - 1)];
      const float x1 = U1 ? w1[r] : 0.0f, x2 = U2 ? w2[r] : 0.0f;
      if (EPI == NUDF_CH_SOFTPLUS) {
        // softplus100 from the two hardware transcendentals, see mlp_chain.hip
        const float tt = 100.0f * v;
        const float z = __builtin_amdgcn_exp2f(fabsf(tt) * -1.44269504f);
        o[i] = (fmaxf(tt, 0.0f) + __builtin_amdgcn_logf(1.0f + z) * 0.69314718f) * (0.01f * cs.scale);
      } else if (EPI == NUDF_CH_NONE) {
        o[i] = v * cs.scale;
      } else if (EPI == NUDF_CH_UDFHEAD) {
        float hv, hm;
        ch_udf_head(cs.iparam, v, cs.scale, hv, hm);
        o[i] = hv;
        o2[i] = hm;
      } else if (EPI == NUDF_CH_RELU) {
        o[i] = fmaxf(v, 0.0f);
        o2[i] = o[i];
      } else if (EPI == NUDF_CH_RELUADD) {
        o[i] = fmaxf(v + x2, 0.0f);
      } else if (EPI == NUDF_CH_SIGMOIDN) {
        o2[i] = v;
        o[i] = (f0 + i < cs.iparam) ? 1.0f / (1.0f + expf(-v)) : 0.0f;
      } else if (EPI == NUDF_CH_MULMASK) {
        o[i] = (x1 > 0.0f) ? v * cs.scale : 0.0f;
      } else if (EPI == NUDF_CH_ADDMASK) {
        o[i] = (x1 > 0.0f) ? (v + x2) * cs.scale : 0.0f;
      } else {
        // 1 - softplus'(a) = exp(-100 h) from the stored activation h >= 0.  |h|: features >= N of the last tile read
        // a clamped column of X1 that need not be a softplus output (the skip layer's input carries PE values < 0
        // behind its 217 hidden columns); exp(+70) would overflow to inf and 0 * inf = NaN would reach the tile's
        // pad columns, where the next step multiplies them by its zero weight rows.  Free (source modifier).
        const float x = 100.0f * cs.xscale * fabsf(x1);
        const float om = __builtin_amdgcn_exp2f(x * -1.44269504f);
        const float sg = 1.0f - om;
        if (EPI == NUDF_CH_MULSP) {
          o[i] = v * sg * cs.scale;               // columns >= iparam (skip split): finite, meet zero weight rows
          o2[i] = v * cs.scale;
        } else if (EPI == NUDF_CH_TANGENT) {
          o[i] = v * sg * cs.scale;
          o2[i] = v * x2 * 100.0f * om;
        } else {  // NUDF_CH_BWD
          o[i] = v * cs.scale * sg + x2;
        }
      }
    }
    // this quad of the window is consumed: request it for tile t_next (in place)
    if (reload) {
      if (U1) {
        const f32x4 nv = chr_load_quad(cs.X1, cs.x1, 32 * t_next + 4 * h + 8 * q, cs.nq, cs.m1);
#pragma unroll
        for (int i = 0; i < 4; ++i) w1[4 * q + i] = nv[i];
      }
      if (U2 && cs.X2) {
        const f32x4 nv = chr_load_quad(cs.X2, cs.x2, 32 * t_next + 4 * h + 8 * q, cs.nq, cs.m2);
#pragma unroll
        for (int i = 0; i < 4; ++i) w2[4 * q + i] = nv[i];
      }
    }
    // ---- stores ----
    if (EPI == NUDF_CH_UDFHEAD) {
      if (q == 0 && t == 0 && h == 0) {     // feature 0 = register 0 of the lower half-wave
        if (cs.C2) cs.C2[cs.row] = o[0];
        if (cs.C1) cs.C1[cs.row] = o2[0];
      }
      continue;
    }
    if (EPI == NUDF_CH_SIGMOIDN) {
      if (cs.C1) chr_store_range(cs.C1, cs.c1, f0, 0, min(N, cs.iparam), 0, o, cs.mc1);
      if (cs.C2) {
        if (N <= cs.iparam) chr_store_range(cs.C2, cs.c2, f0, 0, N, 0, o, cs.mc2);             // C2 mirrors the sigmoid outputs
        else chr_store_range(cs.C2, cs.c2, f0, cs.iparam, N, cs.iparam, o2, cs.mc2);   // raw columns
      }
    } else {
      // whole quads; lim1 / lim2 = columns the row holds (the tail of the last quad lands in pad columns)
      if (cs.C1 && f0 < cs.lim1) *(chr_gp4)(cs.C1 + (cs.c1 + (unsigned)f0 * cs.mc1)) = o;
      if (EPI == NUDF_CH_MULSP) {
        if (cs.iparam > 0 && cs.C2) chr_store_range(cs.C2, cs.c2, f0, cs.iparam, N, cs.iparam, o2, cs.mc2);   // embedding branch
      } else if (EPI == NUDF_CH_TANGENT || EPI == NUDF_CH_RELU) {
        if (cs.C2 && f0 < cs.lim2) *(chr_gp4)(cs.C2 + (cs.c2 + (unsigned)f0 * cs.mc2)) = o2;
      }
    }
    if (cs.act_write) {
      float* ap = act + ln * CH_LD + cs.act_col0 + f0;
      if (EPI != NUDF_CH_SIGMOIDN) {
        *reinterpret_cast<f32x4*>(ap) = o;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = o[i];
      }
    }
  }
}

// All feature tiles of a step, straight-line (accumulator tile and window slot are compile-time registers).  Tile t's
// stored-state operands sit in window slot t % W (the first W tiles were requested under the K loop) and are
// re-requested for tile t + W, quad by quad, as they are consumed.  The scheduling barriers keep the tile bodies
// apart: hoisting the accumulator reads / window loads of later tiles is what made the register allocator spill.
template <int EPI, int W>
__device__ __forceinline__ void chr_epilogue(const NudfChainStep& st, float* act, const ChrStep& cs, int NT, int h, int ln,
                                             f32x16 (&acc)[8], float (&px1)[W][16], float (&px2)[W][16]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t >= NT) return;
    chr_epi_tile<EPI>(st, act, cs, t, h, ln, acc[t], px1[t % W], px2[t % W], t + W < NT, t + W);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// K loop of one step: NCT feature tiles x 32 points.  Two register sets, the next k group's LDS / L2 reads are
// issued above the current group's 4 * NCT MFMAs (sched_barriers pin them there); `tail` issues the epilogue's
// operand requests after the last weight loads, so that the in-order vmcnt lets them stay in flight under the last
// 4 * NCT MFMAs.  MFMA operand order (weights, activations): the product comes out transposed.
template <int NCT, class Tail>
__device__ __forceinline__ void chr_mma(const float* __restrict__ arow, const f32x4* __restrict__ bptr, size_t bstride,
                                        int G, const float* bb, f32x16 (&acc)[8], Tail&& tail) {
  f32x4 a0, a1, b0[NCT], b1[NCT];
  // accumulators start as the bias (staged in LDS by the previous step): lane half h holds features 4h + 8q + i
#pragma unroll
  for (int j = 0; j < NCT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bb + 32 * j + 8 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][4 * q + i] = b[i];
    }
  a0 = *reinterpret_cast<const f32x4*>(arow);
#pragma unroll
  for (int j = 0; j < NCT; ++j) b0[j] = bptr[j * 64];
#pragma unroll 1
  for (int g = 0; g < G - 2; g += 2) {
    {
      const f32x4* bq = bptr + (size_t)(g + 1) * bstride;
#pragma unroll
      for (int j = 0; j < NCT; ++j) b1[j] = bq[j * 64];
      a1 = *reinterpret_cast<const f32x4*>(arow + (g + 1) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[j] = ch_mfma(b0[j][jj], a0[jj], acc[j]);
    __builtin_amdgcn_sched_barrier(0);
    {
      const f32x4* bq = bptr + (size_t)(g + 2) * bstride;
#pragma unroll
      for (int j = 0; j < NCT; ++j) b0[j] = bq[j * 64];
      a0 = *reinterpret_cast<const f32x4*>(arow + (g + 2) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[j] = ch_mfma(b1[j][jj], a1[jj], acc[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const f32x4* bq = bptr + (size_t)(G - 1) * bstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) b1[j] = bq[j * 64];
    a1 = *reinterpret_cast<const f32x4*>(arow + (G - 1) * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int j = 0; j < NCT; ++j) acc[j] = ch_mfma(b0[j][jj], a0[jj], acc[j]);
  __builtin_amdgcn_sched_barrier(0);
  tail();      // b0 / a0 are dead here: the window's registers do not add to the K loop's peak
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int j = 0; j < NCT; ++j) acc[j] = ch_mfma(b1[j][jj], a1[jj], acc[j]);
  __builtin_amdgcn_sched_barrier(0);
}

// bias of a step -> this wave's LDS staging buffer (zeros beyond N / without a bias): lane l stages features 4l..4l+3
__device__ __forceinline__ f32x4 chr_bias_fetch(const NudfChainStep& st, int lane) {
  f32x4 b = {0.0f, 0.0f, 0.0f, 0.0f};
  if (st.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 4 * lane + i;
      b[i] = (f < st.N) ? st.bias[f] : 0.0f;
    }
  }
  return b;
}

// XCLS = which stored-state operands the sweep's epilogues read: 0 none (forward sweeps), 1 X1 only (input-gradient
// sweeps), 2 X1 and X2 (tangent / adjoint sweeps); W = tiles per operand window.  Separate instantiations keep the
// windows (16 W registers per operand) and the unused epilogues out of the sweeps that do not need them.
template <int XCLS, int W>
__global__ __launch_bounds__(CHR_WAVES * 64, 1) void mlp_chain_rows_kernel(NudfChain p_arg) {
  (void)p_arg;   // read where it lies (see mlp_chain_kernel): no private copy for the optimiser to prove away
  const NudfChain& p = *(const NudfChain*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) ChainRowsSmem sm;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, ln = lane & 31;
  const int m0 = (blockIdx.x * CHR_WAVES + wave) * CHR_ROWS;
  if (m0 >= p.P) return;     // whole-wave exit: nothing in this kernel waits for another wave
  float* act = sm.act[wave];
  float* xs = sm.xs[wave];
  float* vs = sm.vs[wave];

  // ---- tile initialisation ---------------------------------------------------------------------
  if (p.x) {
    for (int e = lane; e < CHR_ROWS * 3; e += 64) {
      int r = m0 + e / 3;
      if (r > p.P - 1) r = p.P - 1;
      xs[e] = p.x[(size_t)(r / p.x_div) * 3 + (e % 3)];
      vs[e] = p.v ? p.v[(size_t)r * 3 + (e % 3)] : 0.0f;
    }
  }
  *reinterpret_cast<f32x4*>(sm.bias[wave][0] + 4 * lane) = chr_bias_fetch(p.step[0], lane);
  chr_wave_sync();
  if (p.init == NUDF_CH_INIT_LOAD) {
    const int k4 = p.k0 >> 2;
    for (int e = lane; e < CHR_ROWS * k4; e += 64) {
      const int r = e / k4, c4 = e - r * k4;
      int gr = m0 + r;
      if (gr > p.P - 1) gr = p.P - 1;
      const f32x4 val = *reinterpret_cast<const f32x4*>(p.A0 + (size_t)gr * p.lda0 + c4 * 4);
      *reinterpret_cast<f32x4*>(act + r * CH_LD + c4 * 4) = val;
    }
  } else if (p.init == NUDF_CH_INIT_POSENC) {
    chr_write_pe(act, xs, vs, p, lane, m0, 0, 1.0f, p.G0, p.ldg0, 0, p.k0);
  } else if (p.init == NUDF_CH_INIT_SEED) {
    const int C = p.k0;
    for (int e = lane; e < CHR_ROWS * C; e += 64) {
      const int r = e / C, c = e - r * C;
      int gr = m0 + r;
      const bool live = gr < p.P;
      if (!live) gr = p.P - 1;
      float s, om;
      ch_sp_derivs(p.A0[(size_t)gr * p.lda0 + c], p.seed_xscale, s, om);
      const float val = p.seed_sign[gr] * p.seed_wrow[c] * p.seed_scale * s;
      act[r * CH_LD + c] = val;
      if (p.G0 && live) p.G0[(size_t)gr * p.ldg0 + c] = val;
    }
  }
  chr_wave_sync();

  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * CHR_WAVES + wave) * 64 : nullptr;
  if (dbg && lane == 0) {
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    dbg[1] = __builtin_amdgcn_s_memtime();
  }

  const float* arow = act + ln * CH_LD + 4 * h;
  ChrStep cs;
  cs.row = (unsigned)(m0 + ln);
  cs.m1 = cs.m2 = cs.mc1 = cs.mc2 = 1u;

  // ---- the layer chain ---------------------------------------------------------------------------
  for (int si = 0; si < p.n_steps; ++si) {
    const NudfChainStep& st = p.step[si];
    const int G = st.K >> 3;
    const int NT = (st.N + 31) >> 5;
    const f32x4* __restrict__ bptr = reinterpret_cast<const f32x4*>(st.Bp) + lane;
    const size_t bstride = (size_t)NT * 64;
    cs.x1 = cs.row * (unsigned)st.ldx1;
    cs.x2 = cs.row * (unsigned)st.ldx2;
    cs.c1 = cs.row * (unsigned)st.ldc1;
    cs.c2 = cs.row * (unsigned)st.ldc2;
    cs.X1 = (chr_gcp)chr_pin(st.X1);
    cs.X2 = (chr_gcp)chr_pin(st.X2);
    cs.C1 = (chr_gp)chr_pin(st.C1);
    cs.C2 = (chr_gp)chr_pin(st.C2);
    cs.N = chr_pin(st.N);
    cs.nq = chr_pin(((st.N + 3) & ~3) - 4);
    cs.iparam = chr_pin(st.iparam);
    cs.act_col0 = chr_pin(st.act_col0);
    cs.act_write = chr_pin(st.act_write);
    {
      const int flim = (st.epi == NUDF_CH_MULSP && st.iparam > 0) ? min(st.N, st.iparam) : st.N;
      cs.lim1 = chr_pin(min((flim + 3) & ~3, st.ldc1));
      cs.lim2 = chr_pin(min((st.N + 3) & ~3, st.ldc2));
    }
    cs.scale = chr_pin(st.scale);
    cs.xscale = chr_pin(st.xscale);

#if CHR_STEP_SYNC
    __builtin_amdgcn_s_barrier();     // speed only (L1 sharing of the weight stream); terminated waves are not counted
#endif
    f32x16 acc[8];       // initialised inside chr_mma (only the NT tiles that exist) from the staged bias
    const float* bb = sm.bias[wave][si & 1] + 4 * h;
    // next step's bias: requested now, staged into LDS after the K loop
    f32x4 nbias = {0.0f, 0.0f, 0.0f, 0.0f};
    if (si + 1 < p.n_steps) nbias = chr_bias_fetch(p.step[si + 1], lane);

    float px1[W][16], px2[W][16];
    const bool u1 = XCLS >= 1 && CH_USES_X1(st.epi), u2 = XCLS >= 2 && CH_USES_X2(st.epi) && st.X2 != nullptr;
    auto tail = [&]() {
      if (XCLS == 0) return;
#pragma unroll
      for (int u = 0; u < W; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v1 = {0.0f, 0.0f, 0.0f, 0.0f}, v2 = {0.0f, 0.0f, 0.0f, 0.0f};
          if (u1 && u < NT) v1 = chr_load_quad(cs.X1, cs.x1, 32 * u + 4 * h + 8 * q, cs.nq);
          if (XCLS >= 2 && u2 && u < NT) v2 = chr_load_quad(cs.X2, cs.x2, 32 * u + 4 * h + 8 * q, cs.nq);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            px1[u][4 * q + i] = v1[i];
            if (XCLS >= 2) px2[u][4 * q + i] = v2[i];
          }
        }
    };
    switch (NT) {
      case 1: chr_mma<1>(arow, bptr, bstride, G, bb, acc, tail); break;
      case 2: chr_mma<2>(arow, bptr, bstride, G, bb, acc, tail); break;
      case 3: chr_mma<3>(arow, bptr, bstride, G, bb, acc, tail); break;
      case 4: chr_mma<4>(arow, bptr, bstride, G, bb, acc, tail); break;
      case 5: chr_mma<5>(arow, bptr, bstride, G, bb, acc, tail); break;
      case 6: chr_mma<6>(arow, bptr, bstride, G, bb, acc, tail); break;
      case 7: chr_mma<7>(arow, bptr, bstride, G, bb, acc, tail); break;
      default: chr_mma<8>(arow, bptr, bstride, G, bb, acc, tail); break;
    }
    *reinterpret_cast<f32x4*>(sm.bias[wave][(si + 1) & 1] + 4 * lane) = nbias;
    if (dbg && lane == 0) dbg[2 + 4 * si] = dbg[3 + 4 * si] = __builtin_amdgcn_s_memtime();

    switch (st.epi) {
      case NUDF_CH_SOFTPLUS: chr_epilogue<NUDF_CH_SOFTPLUS, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_NONE: chr_epilogue<NUDF_CH_NONE, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_RELU: chr_epilogue<NUDF_CH_RELU, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_SIGMOIDN: chr_epilogue<NUDF_CH_SIGMOIDN, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_UDFHEAD: chr_epilogue<NUDF_CH_UDFHEAD, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_MULSP: if (XCLS >= 1) chr_epilogue<NUDF_CH_MULSP, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_MULMASK: if (XCLS >= 1) chr_epilogue<NUDF_CH_MULMASK, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_TANGENT: if (XCLS >= 2) chr_epilogue<NUDF_CH_TANGENT, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_BWD: if (XCLS >= 2) chr_epilogue<NUDF_CH_BWD, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_ADDMASK: if (XCLS >= 2) chr_epilogue<NUDF_CH_ADDMASK, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      case NUDF_CH_RELUADD: if (XCLS >= 2) chr_epilogue<NUDF_CH_RELUADD, W>(st, act, cs, NT, h, ln, acc, px1, px2); break;
      default: break;
    }
    if (dbg && lane == 0) dbg[4 + 4 * si] = dbg[5 + 4 * si] = __builtin_amdgcn_s_memtime();
    chr_wave_sync();
    if (st.pe_tail_col >= 0) {
      const int pe_end = st.pe_tail_col + 3 * (2 * p.pe_L + 1);
      chr_write_pe(act, xs, vs, p, lane, m0, st.pe_tail_col, st.pe_tail_scale, st.pe_dst, st.ld_pe, st.pe_tail_col,
                   min((pe_end + 15) & ~15, 288));
      chr_wave_sync();
    }
  }
}

// =====================================================================================================
// Workgroup-SHARED 64-point tile with the transposed product ("tq"): the tile geometry, barriers and occupancy of
// mlp_chain_kernel<64> (4 waves, each up to 2 x 2 tiles of 32 x 32, two workgroups per CU, one wave of each per
// SIMD) with this file's accumulator orientation -- lane = point, registers = 4 quads of consecutive features -- so
// that every epilogue access is 16 bytes per lane: 4 instead of 16 VMEM instructions per tile and operand.  A wave
// can have 64 VMEM operations in flight (6-bit vmcnt); mlp_chain_kernel issues ~256 b32 loads / stores per layer
// and wave, i.e. at least four HBM round trips per layer spent purely on that limit (scripts/chain_timeline.py:
// the epilogues' memory operations are 4 / 11 / 28 / 18 % of the four UDF sweeps).  Same contract as the wave-private
// kernel above (nudf_chain_rows_class), same step tables; the bias is added after the products, as in
// mlp_chain_kernel, so the two shared-tile kernels agree to the bit wherever the compiler contracts the same way.
// =====================================================================================================
struct ChainTqSmem {
  float act[64 * CH_LD];   // 74 752 B
  float bias[2][256];      //  2 048 B (step s reads [s & 1], stages [~s & 1])
  float xs[64 * 3];
  float vs[64 * 3];
};

template <int NRT, int NCT, class Tail>
__device__ __forceinline__ void tq_mma(const float* __restrict__ arow, const f32x4* __restrict__ bptr, size_t bstride,
                                       int G, f32x16 (&acc)[2][2], Tail&& tail) {
  f32x4 a0[NRT], a1[NRT], b0[NCT], b1[NCT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
  for (int i = 0; i < NRT; ++i) a0[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD);
#pragma unroll
  for (int j = 0; j < NCT; ++j) b0[j] = bptr[j * 64];
#if NUDF_TQ_SPREAD
  // the next group's operand requests spread over the four quarters of this group's MFMAs (weights first: the longer
  // latency) instead of all in front of it: the wave's own request burst no longer sits between two MFMA blocks
  auto grp = [&](const f32x4 (&ac)[NRT], const f32x4 (&bc)[NCT], f32x4 (&an)[NRT], f32x4 (&bn)[NCT], int gn)
      __attribute__((always_inline)) {
    const f32x4* bq = bptr + (size_t)gn * bstride;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (jj == 0) bn[0] = bq[0];
      if (jj == 1 && NCT > 1) bn[NCT - 1] = bq[(NCT - 1) * 64];
      if (jj == 2) an[0] = *reinterpret_cast<const f32x4*>(arow + gn * 8);
      if (jj == 3 && NRT > 1) an[NRT - 1] = *reinterpret_cast<const f32x4*>(arow + (NRT - 1) * 32 * CH_LD + gn * 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(bc[j][jj], ac[i][jj], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int g = 0; g < G - 2; g += 2) {
    grp(a0, b0, a1, b1, g + 1);
    grp(a1, b1, a0, b0, g + 2);
  }
#else
#pragma unroll 1
  for (int g = 0; g < G - 2; g += 2) {
    {
      const f32x4* bq = bptr + (size_t)(g + 1) * bstride;
#pragma unroll
      for (int j = 0; j < NCT; ++j) b1[j] = bq[j * 64];
#pragma unroll
      for (int i = 0; i < NRT; ++i) a1[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + (g + 1) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(b0[j][jj], a0[i][jj], acc[i][j]);
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
        for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(b1[j][jj], a1[i][jj], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
  {
    const f32x4* bq = bptr + (size_t)(G - 1) * bstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) b1[j] = bq[j * 64];
#pragma unroll
    for (int i = 0; i < NRT; ++i) a1[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + (G - 1) * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(b0[j][jj], a0[i][jj], acc[i][j]);
  __builtin_amdgcn_sched_barrier(0);
  tail();      // b0 / a0 are dead here
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(b1[j][jj], a1[i][jj], acc[i][j]);
  __builtin_amdgcn_sched_barrier(0);
}

// tq_mma with a ring of D register sets (D - 1 k groups of LDS / L2 reads in flight instead of one; see ch_mma_ring in
// mlp_chain.hip).  Same MFMA order: bit-identical results.
template <int NRT, int NCT, int D, class Tail>
__device__ __forceinline__ void tq_mma_ring(const float* __restrict__ arow, const f32x4* __restrict__ bptr, size_t bstride,
                                            int G, f32x16 (&acc)[2][2], Tail&& tail) {
  f32x4 a[D][NRT], b[D][NCT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int glast = G - 1;
  auto load = [&](f32x4 (&as)[NRT], f32x4 (&bs)[NCT], int g) {
    g = (g < glast) ? g : glast;
    const f32x4* bq = bptr + (size_t)g * bstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) bs[j] = bq[j * 64];
#pragma unroll
    for (int i = 0; i < NRT; ++i) as[i] = *reinterpret_cast<const f32x4*>(arow + i * 32 * CH_LD + g * 8);
  };
  auto mma = [&](const f32x4 (&as)[NRT], const f32x4 (&bs)[NCT]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int i = 0; i < NRT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(bs[j][jj], as[i][jj], acc[i][j]);
  };
#pragma unroll
  for (int d = 0; d < D - 1; ++d) load(a[d], b[d], d);
  int g = 0;
#pragma unroll 1
  for (; g + D <= G; g += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#if NUDF_TQ_SPREAD
      const int nx = (d + D - 1) % D;
      int gn = g + d + D - 1;
      gn = (gn < glast) ? gn : glast;
      const f32x4* bq = bptr + (size_t)gn * bstride;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if (jj == 0) b[nx][0] = bq[0];
        if (jj == 1 && NCT > 1) b[nx][NCT - 1] = bq[(NCT - 1) * 64];
        if (jj == 2) a[nx][0] = *reinterpret_cast<const f32x4*>(arow + gn * 8);
        if (jj == 3 && NRT > 1) a[nx][NRT - 1] = *reinterpret_cast<const f32x4*>(arow + (NRT - 1) * 32 * CH_LD + gn * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int j = 0; j < NCT; ++j) acc[i][j] = ch_mfma(b[d][j][jj], a[d][i][jj], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      load(a[(d + D - 1) % D], b[(d + D - 1) % D], g + d + D - 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a[d], b[d]);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
  const int rem = G - g;
  // tail: up to D - 1 groups whose operands are already requested; the epilogue's stored-state operands go out behind
  // them, before the last group's MFMAs
#pragma unroll
  for (int d = 0; d < D - 1; ++d) {
    if (d == rem - 1 || (rem == 0 && d == 0)) {
      tail();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (d < rem) mma(a[d], b[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the wave's tiles as straight-line code; X2 of tile t + 1 is requested before tile t is computed and stored.
// S1: so is X1 -- only tile 0's X1 (px1[0][0]) and X2 (px2_0) were requested under the K loop, 32 registers instead of the
// 64 of a whole-step X1 prefetch; the registers pay for a third operand set in the K loop (tq_mma_ring).
template <int EPI, int NRT, int NCT, bool S1 = false>
__device__ __forceinline__ void tq_epilogue(const NudfChainStep& st, float* act, const ChrStep (&cs)[2], int rt0, int ct0,
                                            int h, int ln, f32x16 (&acc)[2][2], float (&px1)[2][2][16], const float* bias_lds,
                                            float (*px2_0)[16] = nullptr) {
  constexpr bool U1 = CH_USES_X1(EPI), U2 = CH_USES_X2(EPI);
  constexpr int NTL = NRT * NCT;
  float xa[2][16], xb[2][16];
  auto issue = [&](float (&x)[16], int i, int j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (cs[i].X2) v = chr_load_quad(cs[i].X2, cs[i].x2, 32 * (ct0 + j) + 4 * h + 8 * q, cs[i].nq, cs[i].m2);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * q + e] = v[e];
    }
  };
  auto issue1 = [&](float (&x)[16], int i, int j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = chr_load_quad(cs[i].X1, cs[i].x1, 32 * (ct0 + j) + 4 * h + 8 * q, cs[i].nq, cs[i].m1);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * q + e] = v[e];
    }
  };
  if (U2) {
    if (S1 && px2_0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[0][r] = (*px2_0)[r];
    } else {
      issue(xb[0], 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (S1 && U1 && t + 1 < NTL) issue1(xa[(t + 1) & 1], (t + 1) / NCT, (t + 1) % NCT);
    if (U2 && t + 1 < NTL) issue(xb[(t + 1) & 1], (t + 1) / NCT, (t + 1) % NCT);
    const int i = t / NCT, j = t % NCT;
    chr_epi_tile<EPI>(st, act + (rt0 + i) * 32 * CH_LD, cs[i], ct0 + j, h, ln, acc[i][j],
                      (S1 && t > 0) ? xa[t & 1] : px1[S1 ? 0 : i][S1 ? 0 : j], xb[t & 1], false, 0, bias_lds);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int EPI, bool S1 = false>
__device__ __forceinline__ void tq_epilogue_any(const NudfChainStep& st, float* act, const ChrStep (&cs)[2], int rt0, int ct0,
                                                int nrt, int nct, int h, int ln, f32x16 (&acc)[2][2],
                                                float (&px1)[2][2][16], const float* bias_lds, float (*px2_0)[16] = nullptr) {
  if (nrt == 2 && nct == 2) tq_epilogue<EPI, 2, 2, S1>(st, act, cs, rt0, ct0, h, ln, acc, px1, bias_lds, px2_0);
  else if (nrt == 2) tq_epilogue<EPI, 2, 1, S1>(st, act, cs, rt0, ct0, h, ln, acc, px1, bias_lds, px2_0);
  else if (nct == 2) tq_epilogue<EPI, 1, 2, S1>(st, act, cs, rt0, ct0, h, ln, acc, px1, bias_lds, px2_0);
  else tq_epilogue<EPI, 1, 1, S1>(st, act, cs, rt0, ct0, h, ln, acc, px1, bias_lds, px2_0);
}

// S1: the epilogues' X1 operand is streamed one tile ahead (tq_epilogue) instead of prefetched for the whole step
template <int XCLS, int RING = 0, bool S1 = false>
__global__ __launch_bounds__(256, 2) void mlp_chain_tq_kernel(NudfChain p_arg) {
  (void)p_arg;   // read where it lies (see mlp_chain_kernel): no private copy for the optimiser to prove away
  const NudfChain& p = *(const NudfChain*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) ChainTqSmem sm;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ln = lane & 31;
  const int m0 = blockIdx.x * 64;

  // ---- tile initialisation (as mlp_chain_kernel) ---------------------------------------------------
  if (p.x) {
    for (int e = tid; e < 64 * 3; e += 256) {
      int r = m0 + e / 3;
      if (r > p.P - 1) r = p.P - 1;
      sm.xs[e] = p.x[(size_t)(r / p.x_div) * 3 + (e % 3)];
      sm.vs[e] = p.v ? p.v[(size_t)r * 3 + (e % 3)] : 0.0f;
    }
  }
  {
    const NudfChainStep& s0 = p.step[0];
    sm.bias[0][tid] = (s0.bias && tid < s0.N) ? s0.bias[tid] : 0.0f;
  }
  __syncthreads();
  if (p.init == NUDF_CH_INIT_LOAD) {
    const int k4 = p.k0 >> 2;
    for (int e = tid; e < 64 * k4; e += 256) {
      const int r = e / k4, c4 = e - r * k4;
      int gr = m0 + r;
      if (gr > p.P - 1) gr = p.P - 1;
      const f32x4 val = *reinterpret_cast<const f32x4*>(p.A0 + (size_t)gr * p.lda0 + c4 * 4);
      *reinterpret_cast<f32x4*>(sm.act + r * CH_LD + c4 * 4) = val;
    }
  } else if (p.init == NUDF_CH_INIT_POSENC) {
    ch_write_pe_rows<256>(sm.act, sm.xs, sm.vs, 64, tid, p, m0, 0, 1.0f, p.G0, p.ldg0, 0, p.k0);
  } else if (p.init == NUDF_CH_INIT_SEED) {
    const int C = p.k0;
    for (int e = tid; e < 64 * C; e += 256) {
      const int r = e / C, c = e - r * C;
      int gr = m0 + r;
      const bool live = gr < p.P;
      if (!live) gr = p.P - 1;
      float s, om;
      const float hst = (p.init_state16 & 4) ? p.A0[ch_blk_off(gr, c, p.lda0)] : p.A0[(size_t)gr * p.lda0 + c];
      ch_sp_derivs(hst, p.seed_xscale, s, om);
      const float val = p.seed_sign[gr] * p.seed_wrow[c] * p.seed_scale * s;
      sm.act[r * CH_LD + c] = val;
      if (p.G0 && live) {
        if (p.init_state16 & 8) p.G0[ch_blk_off(gr, c, p.ldg0)] = val;
        else p.G0[(size_t)gr * p.ldg0 + c] = val;
      }
    }
  }
  __syncthreads();

  // de-phase the two workgroups of a CU once (see mlp_chain_kernel)
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID.wave_id
#if NUDF_TQ_SLEEP
  if (gridDim.x > 256)
    for (unsigned d = 0; d < (slot & 1u) * 2u; ++d) __builtin_amdgcn_s_sleep(127);
#endif

  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * 4 + wave) * 64 : nullptr;
  if (dbg && lane == 0) {
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    dbg[1] = __builtin_amdgcn_s_memtime();
    dbg[62] = wall_clock64();      // 100 MHz reference: the shader clock the sweep ran at
  }

  for (int si = 0; si < p.n_steps; ++si) {
    const NudfChainStep& st = p.step[si];
#if NUDF_TQ_PRIO == 1
    if ((si + slot) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#elif NUDF_TQ_PRIO == 2
    if (si == 0) __builtin_amdgcn_s_setprio(slot & 1u);
#endif
    const int G = st.K >> 3;
    const int NT = (st.N + 31) >> 5;
    int rt0, ct0, nrt, nct;
    if (NT <= 2) { rt0 = wave >> 1; ct0 = wave & 1; nrt = 1; nct = (ct0 < NT) ? 1 : 0; }
    else if (NT <= 4) { rt0 = wave >> 1; ct0 = 2 * (wave & 1); nrt = 1; nct = min(2, max(0, NT - ct0)); }
    else { rt0 = 0; ct0 = 2 * wave; nrt = 2; nct = min(2, max(0, NT - ct0)); }

    ChrStep cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ChrStep& c = cs[i];
      c.row = (unsigned)(m0 + (rt0 + i) * 32 + ln);
      const int lay = chr_pin(st.layout);
      auto base = [&](int ld, bool blk) {   // blocked: [32-row block][quad of features][row in block][4]
        return blk ? (c.row - (unsigned)ln) * (unsigned)ld + 4u * (unsigned)ln : c.row * (unsigned)ld;
      };
      c.x1 = base(st.ldx1, lay & NUDF_CH_BLK_X1); c.m1 = (lay & NUDF_CH_BLK_X1) ? 32u : 1u;
      c.x2 = base(st.ldx2, lay & NUDF_CH_BLK_X2); c.m2 = (lay & NUDF_CH_BLK_X2) ? 32u : 1u;
      c.c1 = base(st.ldc1, lay & NUDF_CH_BLK_C1); c.mc1 = (lay & NUDF_CH_BLK_C1) ? 32u : 1u;
      c.c2 = base(st.ldc2, lay & NUDF_CH_BLK_C2); c.mc2 = (lay & NUDF_CH_BLK_C2) ? 32u : 1u;
      c.X1 = (chr_gcp)chr_pin(st.X1);
      c.X2 = (chr_gcp)chr_pin(st.X2);
      c.C1 = (chr_gp)chr_pin(st.C1);
      c.C2 = (chr_gp)chr_pin(st.C2);
      c.N = chr_pin(st.N);
      c.nq = chr_pin(((st.N + 3) & ~3) - 4);
      c.iparam = chr_pin(st.iparam);
      c.act_col0 = chr_pin(st.act_col0);
      c.act_write = chr_pin(st.act_write);
      const int flim = (st.epi == NUDF_CH_MULSP && st.iparam > 0) ? min(st.N, st.iparam) : st.N;
      c.lim1 = chr_pin(min((flim + 3) & ~3, st.ldc1));
      c.lim2 = chr_pin(min((st.N + 3) & ~3, st.ldc2));
      c.scale = chr_pin(st.scale);
      c.xscale = chr_pin(st.xscale);
    }

    f32x16 acc[2][2];
    float px1[2][2][16];
    float px2[16];      // S1: X2 of the wave's first tile
    // next step's bias: requested now, staged into LDS after the K loop (thread t <-> feature t)
    float nbias = 0.0f;
    if (si + 1 < p.n_steps) {
      const NudfChainStep& sn = p.step[si + 1];
      if (sn.bias && tid < sn.N) nbias = sn.bias[tid];
    }
    if (nct > 0) {
      const float* arow = sm.act + (rt0 * 32 + ln) * CH_LD + 4 * h;
      const f32x4* bptr = reinterpret_cast<const f32x4*>(st.Bp) + (size_t)ct0 * 64 + lane;
      const size_t bstride = (size_t)NT * 64;
      const bool u1 = XCLS >= 1 && CH_USES_X1(st.epi);
      const bool u2 = XCLS >= 2 && CH_USES_X2(st.epi) && cs[0].X2;
      auto tail = [&]() {
        if (XCLS == 0) return;
#pragma unroll
        for (int i = 0; i < (S1 ? 1 : 2); ++i)
#pragma unroll
          for (int j = 0; j < (S1 ? 1 : 2); ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
              if (u1 && i < nrt && j < nct) v = chr_load_quad(cs[i].X1, cs[i].x1, 32 * (ct0 + j) + 4 * h + 8 * q, cs[i].nq, cs[i].m1);
#pragma unroll
              for (int e = 0; e < 4; ++e) px1[i][j][4 * q + e] = v[e];
            }
        if (S1 && XCLS >= 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (u2) v = chr_load_quad(cs[0].X2, cs[0].x2, 32 * ct0 + 4 * h + 8 * q, cs[0].nq, cs[0].m2);
#pragma unroll
            for (int e = 0; e < 4; ++e) px2[4 * q + e] = v[e];
          }
        }
      };
      if (RING > 1) {
        constexpr int D = RING > 1 ? RING : 2;
        if (nrt == 2 && nct == 2) tq_mma_ring<2, 2, D>(arow, bptr, bstride, G, acc, tail);
        else if (nrt == 2) tq_mma_ring<2, 1, D>(arow, bptr, bstride, G, acc, tail);
        else if (nct == 2) tq_mma_ring<1, 2, D>(arow, bptr, bstride, G, acc, tail);
        else tq_mma_ring<1, 1, D>(arow, bptr, bstride, G, acc, tail);
      }
      else if (nrt == 2 && nct == 2) tq_mma<2, 2>(arow, bptr, bstride, G, acc, tail);
      else if (nrt == 2) tq_mma<2, 1>(arow, bptr, bstride, G, acc, tail);
      else if (nct == 2) tq_mma<1, 2>(arow, bptr, bstride, G, acc, tail);
      else tq_mma<1, 1>(arow, bptr, bstride, G, acc, tail);
    }
    sm.bias[(si + 1) & 1][tid] = nbias;
    if (dbg && lane == 0) dbg[2 + 4 * si] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // every wave is done reading the activation tile
    if (dbg && lane == 0) dbg[3 + 4 * si] = __builtin_amdgcn_s_memtime();

    if (nct > 0) {
      switch (st.epi) {
        case NUDF_CH_SOFTPLUS: tq_epilogue_any<NUDF_CH_SOFTPLUS, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_NONE: tq_epilogue_any<NUDF_CH_NONE, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_RELU: tq_epilogue_any<NUDF_CH_RELU, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_SIGMOIDN: tq_epilogue_any<NUDF_CH_SIGMOIDN, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_UDFHEAD: tq_epilogue_any<NUDF_CH_UDFHEAD, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_MULSP: if (XCLS >= 1) tq_epilogue_any<NUDF_CH_MULSP, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_MULMASK: if (XCLS >= 1) tq_epilogue_any<NUDF_CH_MULMASK, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_TANGENT: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_TANGENT, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_BWD: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_BWD, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_ADDMASK: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_ADDMASK, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        case NUDF_CH_RELUADD: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_RELUADD, S1>(st, sm.act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, sm.bias[si & 1], (S1 && XCLS >= 2) ? &px2 : nullptr); break;
        default: break;
      }
    }
    if (dbg && lane == 0) dbg[4 + 4 * si] = __builtin_amdgcn_s_memtime();
    if (st.pe_tail_col >= 0) {
      __syncthreads();
      const int pe_end = st.pe_tail_col + 3 * (2 * p.pe_L + 1);
      ch_write_pe_rows<256>(sm.act, sm.xs, sm.vs, 64, tid, p, m0, st.pe_tail_col, st.pe_tail_scale, st.pe_dst, st.ld_pe,
                            st.pe_tail_col, min((pe_end + 15) & ~15, 288), false, (st.layout & NUDF_CH_BLK_PE) != 0);
    }
    __syncthreads();
    if (dbg && lane == 0) dbg[5 + 4 * si] = __builtin_amdgcn_s_memtime();
  }
  if (dbg && lane == 0) dbg[63] = wall_clock64();
}

// =====================================================================================================
// PAIRED 64-point tiles ("pair"): ONE workgroup of 8 waves owns two 64-point tiles (two LDS activation tiles, 150 KB,
// one workgroup per CU) and runs them in ANTI-PHASE by construction: in every phase one half (4 waves, one per SIMD)
// runs the K loop of a step -- alone on the matrix pipe -- while the other half runs the epilogue of its previous K loop;
// one __syncthreads() per phase swaps the roles.
//
// Why (scripts/chain_timeline.py, round 3): two independent 64-point workgroups per CU drift INTO phase -- while both
// are in their K loops they share the matrix pipe and slow down together, so they also reach their epilogues together
// and the pipe idles: K-loop efficiency 45-48 % (50 % = two waves sharing a saturated pipe) with only 55 % of the
// pipe's time used on the 128-wide colour-net sweeps, 76 % on the UDF forward.  In-phase is the stable state of that
// design (the wave that falls behind gets the whole pipe and catches up); the one-off start-up delay cannot hold the
// workgroups apart.  Here the phase relation is enforced: per step and pair the pipe is busy 2 M of max(M, E) + max(M, E).
//
// Same tiles, step tables, K loop, epilogues and summation order as mlp_chain_tq_kernel (the halves' code IS that
// kernel's): results are bit-identical to it.  A step whose epilogue is followed by a positional-encoding tail (the skip
// layer) needs one barrier between the two inside its half; both halves execute it (the other one after its K loop).
// =====================================================================================================
struct ChainPairSmem {
  float act[2][64 * CH_LD];   // 149 504 B
  float bias[2][2][256];      //   4 096 B
  float xs[2][64 * 3];
  float vs[2][64 * 3];
};

template <int XCLS>
__global__ __launch_bounds__(512, 2) void mlp_chain_pair_kernel(NudfChain p_arg) {
  (void)p_arg;   // read where it lies (see mlp_chain_kernel): no private copy for the optimiser to prove away
  const NudfChain& p = *(const NudfChain*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) ChainPairSmem sm;
  const int tid8 = threadIdx.x;
  const int half = __builtin_amdgcn_readfirstlane(tid8 >> 8);
  const int tid = tid8 & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ln = lane & 31;
  const int m0 = (blockIdx.x * 2 + half) * 64;
  const bool live = m0 < p.P;          // the odd tile of the last pair: that half only keeps the barriers company
  float* act = sm.act[half];
  float* xs = sm.xs[half];
  float* vs = sm.vs[half];

  // ---- tile initialisation (as mlp_chain_tq_kernel, each half its own tile) ------------------------
  if (p.x && live) {
    for (int e = tid; e < 64 * 3; e += 256) {
      int r = m0 + e / 3;
      if (r > p.P - 1) r = p.P - 1;
      xs[e] = p.x[(size_t)(r / p.x_div) * 3 + (e % 3)];
      vs[e] = p.v ? p.v[(size_t)r * 3 + (e % 3)] : 0.0f;
    }
  }
  {
    const NudfChainStep& s0 = p.step[0];
    sm.bias[half][0][tid] = (s0.bias && tid < s0.N) ? s0.bias[tid] : 0.0f;
  }
  __syncthreads();
  if (live) {
    if (p.init == NUDF_CH_INIT_LOAD) {
      const int k4 = p.k0 >> 2;
      for (int e = tid; e < 64 * k4; e += 256) {
        const int r = e / k4, c4 = e - r * k4;
        int gr = m0 + r;
        if (gr > p.P - 1) gr = p.P - 1;
        const f32x4 val = *reinterpret_cast<const f32x4*>(p.A0 + (size_t)gr * p.lda0 + c4 * 4);
        *reinterpret_cast<f32x4*>(act + r * CH_LD + c4 * 4) = val;
      }
    } else if (p.init == NUDF_CH_INIT_POSENC) {
      ch_write_pe_rows<256>(act, xs, vs, 64, tid, p, m0, 0, 1.0f, p.G0, p.ldg0, 0, p.k0);
    } else if (p.init == NUDF_CH_INIT_SEED) {
      const int C = p.k0;
      for (int e = tid; e < 64 * C; e += 256) {
        const int r = e / C, c = e - r * C;
        int gr = m0 + r;
        const bool lv = gr < p.P;
        if (!lv) gr = p.P - 1;
        float sg, om;
        const float hst = (p.init_state16 & 4) ? p.A0[ch_blk_off(gr, c, p.lda0)] : p.A0[(size_t)gr * p.lda0 + c];
        ch_sp_derivs(hst, p.seed_xscale, sg, om);
        const float val = p.seed_sign[gr] * p.seed_wrow[c] * p.seed_scale * sg;
        act[r * CH_LD + c] = val;
        if (p.G0 && lv) {
          if (p.init_state16 & 8) p.G0[ch_blk_off(gr, c, p.ldg0)] = val;
          else p.G0[(size_t)gr * p.ldg0 + c] = val;
        }
      }
    }
  }
  __syncthreads();

  unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * 8 + half * 4 + wave) * 64 : nullptr;
  if (dbg && lane == 0) {
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    dbg[1] = __builtin_amdgcn_s_memtime();
    dbg[62] = wall_clock64();
  }

  const int L = p.n_steps;
  // Phase ph = 0 .. 2 L, one __syncthreads() each: half 0 runs K(0) E(0) K(1) E(1) ... and idles in the last phase, half 1
  // idles in the first and runs the same sequence one phase later -- so that K loops and epilogues of the two halves
  // alternate.  Written per step (K loop, barrier, epilogue, barrier: the straight-line body of mlp_chain_tq_kernel, whose
  // register allocation it keeps) with the idle phases as one extra barrier in front (half 1) / behind (half 0).
  if (half == 1) __syncthreads();

  for (int si = 0; si < L; ++si) {
    const NudfChainStep& st = p.step[si];
    const int G = st.K >> 3;
    const int NT = (st.N + 31) >> 5;
    int rt0, ct0, nrt, nct;
    if (NT <= 2) { rt0 = wave >> 1; ct0 = wave & 1; nrt = 1; nct = (ct0 < NT) ? 1 : 0; }
    else if (NT <= 4) { rt0 = wave >> 1; ct0 = 2 * (wave & 1); nrt = 1; nct = min(2, max(0, NT - ct0)); }
    else { rt0 = 0; ct0 = 2 * wave; nrt = 2; nct = min(2, max(0, NT - ct0)); }
    if (!live) nct = 0;

    ChrStep cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ChrStep& c = cs[i];
      c.row = (unsigned)(m0 + (rt0 + i) * 32 + ln);
      const int lay = chr_pin(st.layout);
      auto base = [&](int ld, bool blk) {
        return blk ? (c.row - (unsigned)ln) * (unsigned)ld + 4u * (unsigned)ln : c.row * (unsigned)ld;
      };
      c.x1 = base(st.ldx1, lay & NUDF_CH_BLK_X1); c.m1 = (lay & NUDF_CH_BLK_X1) ? 32u : 1u;
      c.x2 = base(st.ldx2, lay & NUDF_CH_BLK_X2); c.m2 = (lay & NUDF_CH_BLK_X2) ? 32u : 1u;
      c.c1 = base(st.ldc1, lay & NUDF_CH_BLK_C1); c.mc1 = (lay & NUDF_CH_BLK_C1) ? 32u : 1u;
      c.c2 = base(st.ldc2, lay & NUDF_CH_BLK_C2); c.mc2 = (lay & NUDF_CH_BLK_C2) ? 32u : 1u;
      c.X1 = (chr_gcp)chr_pin(st.X1);
      c.X2 = (chr_gcp)chr_pin(st.X2);
      c.C1 = (chr_gp)chr_pin(st.C1);
      c.C2 = (chr_gp)chr_pin(st.C2);
      c.N = chr_pin(st.N);
      c.nq = chr_pin(((st.N + 3) & ~3) - 4);
      c.iparam = chr_pin(st.iparam);
      c.act_col0 = chr_pin(st.act_col0);
      c.act_write = chr_pin(st.act_write);
      const int flim = (st.epi == NUDF_CH_MULSP && st.iparam > 0) ? min(st.N, st.iparam) : st.N;
      c.lim1 = chr_pin(min((flim + 3) & ~3, st.ldc1));
      c.lim2 = chr_pin(min((st.N + 3) & ~3, st.ldc2));
      c.scale = chr_pin(st.scale);
      c.xscale = chr_pin(st.xscale);
    }

    f32x16 acc[2][2];
    float px1[2][2][16];
    float nbias = 0.0f;
    if (si + 1 < L) {
      const NudfChainStep& sn = p.step[si + 1];
      if (sn.bias && tid < sn.N) nbias = sn.bias[tid];
    }
    // ---------------- K loop of step si: this half alone on the matrix pipe ----------------
    if (nct > 0) {
      const float* arow = act + (rt0 * 32 + ln) * CH_LD + 4 * h;
      const f32x4* bptr = reinterpret_cast<const f32x4*>(st.Bp) + (size_t)ct0 * 64 + lane;
      const size_t bstride = (size_t)NT * 64;
      const bool u1 = XCLS >= 1 && CH_USES_X1(st.epi);
      auto tail = [&]() {
        if (XCLS == 0) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
              if (u1 && i < nrt && j < nct) v = chr_load_quad(cs[i].X1, cs[i].x1, 32 * (ct0 + j) + 4 * h + 8 * qd, cs[i].nq, cs[i].m1);
#pragma unroll
              for (int e = 0; e < 4; ++e) px1[i][j][4 * qd + e] = v[e];
            }
      };
      if (nrt == 2 && nct == 2) tq_mma<2, 2>(arow, bptr, bstride, G, acc, tail);
      else if (nrt == 2) tq_mma<2, 1>(arow, bptr, bstride, G, acc, tail);
      else if (nct == 2) tq_mma<1, 2>(arow, bptr, bstride, G, acc, tail);
      else tq_mma<1, 1>(arow, bptr, bstride, G, acc, tail);
    }
    sm.bias[half][(si + 1) & 1][tid] = nbias;
    if (dbg && lane == 0) dbg[2 + 4 * si] = __builtin_amdgcn_s_memtime();
    // the epilogue the OTHER half runs in this phase (half 0 sees E(si - 1) of half 1, half 1 sees E(si) of half 0): if
    // it ends with a positional-encoding tail, it has one barrier inside -- keep it company
    {
      const int so = (half == 0) ? si - 1 : si;
      if (so >= 0 && p.step[so].pe_tail_col >= 0) __syncthreads();
    }
    __syncthreads();  // phase boundary: the other half's epilogue is done, its K loop may start; ours starts its epilogue
    if (dbg && lane == 0) dbg[3 + 4 * si] = __builtin_amdgcn_s_memtime();

    // ---------------- epilogue of step si: beside the other half's K loop ----------------
    if (nct > 0) {
      const float* bl = sm.bias[half][si & 1];
      switch (st.epi) {
        case NUDF_CH_SOFTPLUS: tq_epilogue_any<NUDF_CH_SOFTPLUS>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_NONE: tq_epilogue_any<NUDF_CH_NONE>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_RELU: tq_epilogue_any<NUDF_CH_RELU>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_SIGMOIDN: tq_epilogue_any<NUDF_CH_SIGMOIDN>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_UDFHEAD: tq_epilogue_any<NUDF_CH_UDFHEAD>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_MULSP: if (XCLS >= 1) tq_epilogue_any<NUDF_CH_MULSP>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_MULMASK: if (XCLS >= 1) tq_epilogue_any<NUDF_CH_MULMASK>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_TANGENT: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_TANGENT>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_BWD: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_BWD>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_ADDMASK: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_ADDMASK>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        case NUDF_CH_RELUADD: if (XCLS >= 2) tq_epilogue_any<NUDF_CH_RELUADD>(st, act, cs, rt0, ct0, nrt, nct, h, ln, acc, px1, bl); break;
        default: break;
      }
    }
    if (dbg && lane == 0) dbg[4 + 4 * si] = __builtin_amdgcn_s_memtime();
    if (st.pe_tail_col >= 0) {
      __syncthreads();
      if (live) {
        const int pe_end = st.pe_tail_col + 3 * (2 * p.pe_L + 1);
        ch_write_pe_rows<256>(act, xs, vs, 64, tid, p, m0, st.pe_tail_col, st.pe_tail_scale, st.pe_dst, st.ld_pe,
                              st.pe_tail_col, min((pe_end + 15) & ~15, 288), false, (st.layout & NUDF_CH_BLK_PE) != 0);
      }
    }
    __syncthreads();  // phase boundary
    if (dbg && lane == 0) dbg[5 + 4 * si] = __builtin_amdgcn_s_memtime();
  }
  if (half == 0) {    // last phase: half 1 runs E(L - 1)
    if (p.step[L - 1].pe_tail_col >= 0) __syncthreads();
    __syncthreads();
  }
  if (dbg && lane == 0) dbg[63] = wall_clock64();
}

// NUDF_CHAIN_PAIR: 0 (default) = independent 64-point workgroups; 1 = launches of at least 32 768 points that already go
// to the transposed-product kernel (blocked state: the UDF sweeps) run as paired tiles; 2 = every fp32 launch of at least
// 32 768 points that meets that kernel's contract does (colour / NeRF chains, coarse forward).  MEASURED (round 3,
// profiles/r03_chain_pair.txt): not faster -- a K loop alone on the pipe next to the partner's epilogue reaches only 77 %
// of the pipe rate (two in-phase K loops together 95 %), and the epilogue beside a K loop takes 1.8x as long; the
// kernel stays as the measured counter-example to "enforce anti-phase".
int nudf_chain_pair_mode() {
  static const int mode = [] {
    const char* e = getenv("NUDF_CHAIN_PAIR");
    return e ? atoi(e) : 0;
  }();
  return mode;
}

int nudf_mlp_chain_tq_launch(const NudfChain& p, int cls, hipStream_t st, int force_pair) {
  const dim3 grid((p.P + 63) / 64), block(256);
  // forward sweeps (no stored-state operand): three register sets in the K loop, two k groups of operand reads in flight
  // (tq_mma_ring; measured 629 -> 605 us at 65 536 points).  NUDF_TQ_RING=0: two sets.
  static const int ring = [] {
    const char* e = getenv("NUDF_TQ_RING");
    return e ? atoi(e) : 3;
  }();
  static const int s1 = [] {
    const char* e = getenv("NUDF_TQ_S1");
    return e ? atoi(e) : 1;
  }();
  if (force_pair || (nudf_chain_pair_mode() > 0 && p.P >= 32768)) {   // paired tiles: tile_rows 130 / NUDF_CHAIN_PAIR
    const dim3 pgrid((p.P + 127) / 128), pblock(512);
    if (cls == 0) hipLaunchKernelGGL((mlp_chain_pair_kernel<0>), pgrid, pblock, 0, st, p);
    else if (cls == 1) hipLaunchKernelGGL((mlp_chain_pair_kernel<1>), pgrid, pblock, 0, st, p);
    else hipLaunchKernelGGL((mlp_chain_pair_kernel<2>), pgrid, pblock, 0, st, p);
    NUDF_CHECK_LAUNCH("nudf_mlp_chain(pair)");
    return 0;
  }
  if (cls == 0 && ring == 3) hipLaunchKernelGGL((mlp_chain_tq_kernel<0, 3>), grid, block, 0, st, p);
  else if (cls == 0) hipLaunchKernelGGL((mlp_chain_tq_kernel<0>), grid, block, 0, st, p);
  // input-gradient sweeps: X1 streamed one tile ahead (NUDF_TQ_S1, default 1) frees the registers for the third operand
  // set (197 VGPRs, 646-652 -> 637 us at 65 536 points).  The tangent / adjoint instantiation gets SLOWER with a streamed
  // X1, with two or three operand sets (tangent 675 -> 703 us, adjoint 692-750 -> 727-775: two more exposed operand
  // requests per tile): it keeps the whole-step prefetch; NUDF_TQ_S1=2 / 3 select those builds for measurements.
  else if (cls == 1 && s1 && ring == 3) hipLaunchKernelGGL((mlp_chain_tq_kernel<1, 3, true>), grid, block, 0, st, p);
  else if (cls == 1 && s1) hipLaunchKernelGGL((mlp_chain_tq_kernel<1, 0, true>), grid, block, 0, st, p);
  else if (cls == 1) hipLaunchKernelGGL((mlp_chain_tq_kernel<1>), grid, block, 0, st, p);
  else if (s1 == 3 && ring == 3) hipLaunchKernelGGL((mlp_chain_tq_kernel<2, 3, true>), grid, block, 0, st, p);
  else if (s1 >= 2) hipLaunchKernelGGL((mlp_chain_tq_kernel<2, 0, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((mlp_chain_tq_kernel<2>), grid, block, 0, st, p);
  NUDF_CHECK_LAUNCH("nudf_mlp_chain(tq)");
  return 0;
}

// The launch-time contract above; also picks the operand class.  Returns -1 when the workgroup-shared kernel must run.
int nudf_chain_rows_class(const NudfChain& p, bool allow_blocked) {
  int cls = 0;
  if ((p.init_state16 & 12) && !allow_blocked) return -1;
  auto vec_ok = [](const void* q, int ld) { return ((((uintptr_t)q) | ((unsigned)ld << 2)) & 15) == 0; };
  for (int i = 0; i < p.n_steps; ++i) {
    const NudfChainStep& s = p.step[i];
    if (s.prec != 0 || (s.layout & (NUDF_CH_STATE16 | NUDF_CH_P4_X1 | NUDF_CH_P4_C1))) return -1;
    if ((s.layout & 31) && !allow_blocked) return -1;
    const int e = s.epi;
    if (CH_USES_X1(e)) {
      if (!s.X1 || !vec_ok(s.X1, s.ldx1)) return -1;
      cls = cls < 1 ? 1 : cls;
    }
    if (CH_USES_X2(e)) {
      if (s.X2 && !vec_ok(s.X2, s.ldx2)) return -1;
      cls = 2;
    }
    if (e != NUDF_CH_UDFHEAD && e != NUDF_CH_SIGMOIDN) {
      if (s.C1 && !vec_ok(s.C1, s.ldc1)) return -1;
      if (s.act_write && (s.act_col0 & 3)) return -1;
    }
    if ((e == NUDF_CH_TANGENT || e == NUDF_CH_RELU) && s.C2 && !vec_ok(s.C2, s.ldc2)) return -1;
    if (s.r1_row && ((e != NUDF_CH_BWD && e != NUDF_CH_MULMASK) || !s.r1_col)) return -1;
  }
  return cls;
}

// launch (argument checks are done by nudf_mlp_chain): one workgroup = 4 waves = 128 points
int nudf_mlp_chain_rows_launch(const NudfChain& p, int cls, hipStream_t st) {
  const dim3 grid((p.P + CHR_WAVES * CHR_ROWS - 1) / (CHR_WAVES * CHR_ROWS)), block(CHR_WAVES * 64);
  // operand window (tiles in flight per stored-state operand): 4 with one operand; 2 with two operands (3 makes the
  // register allocator spill ~100 values per step; NUDF_CHAIN_WIN2=3 selects that build for measurements)
  static const int win2 = [] {
    const char* e = getenv("NUDF_CHAIN_WIN2");
    return (e && e[0] == '3') ? 3 : 2;
  }();
  if (cls == 0) hipLaunchKernelGGL((mlp_chain_rows_kernel<0, 4>), grid, block, 0, st, p);
  else if (cls == 1) hipLaunchKernelGGL((mlp_chain_rows_kernel<1, 4>), grid, block, 0, st, p);
  else if (win2 == 3) hipLaunchKernelGGL((mlp_chain_rows_kernel<2, 3>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((mlp_chain_rows_kernel<2, 2>), grid, block, 0, st, p);
  NUDF_CHECK_LAUNCH("nudf_mlp_chain(rows)");
  return 0;
}
