// Weight-gradient contraction of the MLP chains (gfx950):  C[NA,NB] += A[M,NA]^T B[M,NB]  (+ dbias[NA] += column sums
// of A), reduction over the M sample points, several problems (all layers of a network) in ONE launch.
//
// Replaces the weight / bias gradient halves of autograd's F.linear backward in models/fields.py:192-231 (UDFNetwork,
// first and second order), :452-495 (ResidualRenderingNetwork), :599-628 (NeRF).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 == an fmaf chain; 157.3 TFLOP/s nominal, ~140 sustained) on 128 x 128 x 32 block
// tiles, 4 waves, double-buffered LDS, 2 workgroups per CU.  What makes this kernel's schedule:
//   * the reduction (M = 65 536 points) is split over row chunks so that ONE resident wave of ~512 workgroups covers
//     all tiles of all problems (a second, partial wave of workgroups costs a whole extra pass);
//   * layer widths are ragged (39, 217, 257 ...): a tile's live 32 x 32 sub-tiles are dealt to the four waves ALONG THE
//     SHORTER LIVE SIDE (wave w owns column sub-tile w and loops over the n live row sub-tiles, or the transpose), so a
//     tile costs n in 1..4 units instead of always 4, and the row chunks are sized per tile so that every workgroup
//     gets the same number of MFMAs (cost-weighted split);
//   * partial tiles go to a workspace with plain stores and a second kernel adds them up in a fixed order: run-to-run
//     identical gradients, no fp32 atomics (atomics remain the fallback when the caller passes no workspace);
//   * full tiles with fp32 operands (85 % of a step's work) run kloop_full: one LDS read behind every MFMA, the LDS writes
//     and global loads spread over the MFMA groups, the barrier inside the MFMA block -- a wave never issues a long run of
//     non-MFMA instructions (scripts/ubench/mfma_pair.hip; 79 % MFMA-busy, profiles/r02_pmc_gemm_tn.txt);
//   * blockIdx -> (tile, chunk) keeps the tiles that read the same rows on one XCD / L2 (tn_decode_block);
//   * the 16-bit MFMA mode (config 5) has its own kernel, gemm_tn16_group_kernel: LDS image = bf16 k-pairs.
#include "nudf_common.h"
#include "../../include/nudf.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#ifndef NUDF_MMA16_REPEAT
#define NUDF_MMA16_REPEAT 1  // timing probe (see mlp_chain.hip)
#endif
#define BM 128
#define BN 128
#define BK 32
#define LDT (BM + 4)   // both operand tiles are straight [k][column] copies; +4 keeps rows 16-byte aligned
#define T_TILE (BK * LDT)
#define TN_MAX_TILES 64
#define TN_WS_TILE (BM * BN + BM)   // floats per workspace slot: 4 waves x 4 sub-tiles x 64 lanes x 16 accumulators + bias
#define TNF_NO_EPILOGUE 2           // timing experiments only: results are dropped
#define TNF_NO_BIAS 4
#define TNF_ATOMICS 8
#define TNF_UNIFORM_CHUNKS 16
#define TNF_NO_QUADRANTS 32
#define TNF_NO_XCD_MAP 64           // plain tile-major blockIdx (A/B of the XCD-aware order)
#define TNF_NO_INTERLEAVE 128       // full fp32 tiles through the generic k-loop (A/B of kloop_full)
#define TNF_NO_PACK16 256           // 16-bit MFMA mode through the generic kernel's fp32 LDS image (A/B of gemm_tn16_group_kernel)
#define TNF_NO_SPLIT_IMAGE 512      // bf16x3 mode through the generic kernel (split on the way out of the fp32 image; A/B of gemm_tn3_group_kernel)
#define TNF_GENERIC_STAGE 2048      // bf16x3 split-image kernel: every k-step through the generic (clamped-address) staging: A/B of the
                                    // loop-invariant addressing of round 5; bit-identical
#define TNF_WIDE 1024               // bf16x3 mode: the wide double-buffered kernel (gemm_tn3w_group_kernel) instead of the 128 x 128
                                    // two-barrier one -- opt-in: bit-identical on equal row chunks, measured 8-12 % SLOWER
                                    // (profiles/r05_tn_wide.txt)

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct TnTile {
  int blk_start;        // first workgroup of this tile; its row chunks are consecutive workgroups
  int rows_per_block;
  short prob, ti, tj;
  short layout;         // 0: wave w <-> column sub-tile w, n live row sub-tiles; 1: wave w <-> row sub-tile w, n column
                        // ones; 2 (full tiles): waves 2 x 2, each a 64 x 64 quadrant = 2 x 2 sub-tiles (n = 4)
  short n;
  short gfirst;         // tile group = consecutive tiles of one problem with the same row chunks: its first tile ...
  short gn;             // ... and its size (the group's workgroups are one contiguous range of blockIdx)
  short rot;            // the group's lane rotation of the XCD-aware order (tn_decode_block)
  int grid_start;       // first blockIdx of the tile's GROUP (the same on every tile of the group)
};
struct TnPlan {
  NudfGemmTNProblem prob[NUDF_TN_MAX_PROBLEMS];
  TnTile tile[TN_MAX_TILES + 1];   // tile[n_tiles].blk_start = total workgroups
  int n_tiles, M, prec, flags;
  int grid_blocks;                 // workgroups launched (>= tile[n_tiles].blk_start: the XCD-aware order leaves holes, which exit)
  short partner[TN_MAX_TILES];     // wide bf16x3 kernel (gemm_tn3w_group_kernel): >= 0 = this tile leads a pair with the tile one
                                   // tile row below (same problem, same tile column, same chunks); -1 = a leader without partner;
                                   // -2 = follower (its workgroups exit: the leader's workgroup computes both tiles)
  int assign;                      // reduce kernel: C = 0 + sum instead of C += sum
  int wide;                        // the wide bf16x3 kernel runs this plan (partner[] is set)
  short unit_tile[TN_MAX_TILES];   // wide kernel: workgroup unit u = the leader (or unpaired) tile unit_tile[u] ...
  int n_units, unit_chunks;        // ... and its row chunks: blockIdx = u * unit_chunks + chunk.  n_units * unit_chunks <= 256
                                   // workgroups, one per CU and 32 per XCD whatever the chunk count (the 128 x 128 kernels'
                                   // XCD-aware order puts a chunk's tiles on ONE XCD: 13 chunks x 19 units would load five
                                   // XCDs with 38 one-per-CU workgroups and three with 19 -- two rounds, measured 2x slower)
  float* ws;
  long long* dbg;                  // tuning: per workgroup {start, end} of wall_clock64 (100 MHz), layout, n
  const float* amax_a;             // f16x2 mode (prec 4): device scalars holding max |A| / max |B| over the group's operands
  const float* amax_b;             // (NULL: that side is used unscaled)
};

static_assert(sizeof(TnPlan) <= 4096, "TnPlan is passed by value: kernel arguments are limited to 4 KB");

// 32 x 32 sub-tile (row index, column index) held by accumulator s of a wave
__device__ __forceinline__ int tn_isub(int layout, int wave, int s) {
  return layout == 0 ? s : layout == 1 ? wave : (wave >> 1) * 2 + (s >> 1);
}
__device__ __forceinline__ int tn_jsub(int layout, int wave, int s) {
  return layout == 0 ? wave : layout == 1 ? s : (wave & 1) * 2 + (s & 1);
}

// blockIdx -> (tile, row chunk).  Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own L2:
// the tiles of one problem that read the same rows (same chunk; tiles of a tile row share the A panel, of a tile column
// the B panel) are put on the SAME XCD.  Within a tile group of T > 1 tiles, blockIdx = grid_start + (c / 8) * 8 T + tile * 8 +
// x with lane x = (c + rot) % 8: EVERY chunk's tiles share blockIdx % 8.  The last row of a group whose chunk count C is not a
// multiple of 8 has 8 - C % 8 lanes without a chunk: those workgroups are HOLES (t = -1, they exit at once); the group's
// rotation `rot` is chosen by the planner so that the occupied lanes of all groups load the 8 XCDs evenly.  (Round 2-4 packed
// the last row without holes -- its tiles landed on different XCDs: with C = 13, 5 of 13 chunks read their panels twice from
// HBM, 1.91 GB fetched for 1.21 GB of operands in the UDF adjoint group, and the kernel ran at the HBM rate,
// profiles/r05_tn_l2_sharing.txt.)  Workspace slots stay tile-major (tile.blk_start + chunk).
__host__ __device__ __forceinline__ void tn_decode_block(const TnPlan& g, int block, int& t, int& chunk) {
  int gf = 0;
  while (gf + g.tile[gf].gn < g.n_tiles && g.tile[gf + g.tile[gf].gn].grid_start <= block) gf += g.tile[gf].gn;
  const int T = g.tile[gf].gn;
  const int o = block - g.tile[gf].grid_start;
  t = gf;
  chunk = o;
  if (T > 1) {
    const int C = g.tile[gf + 1].blk_start - g.tile[gf].blk_start;
    const int c_hi = o / (8 * T), rem = o - c_hi * 8 * T;
    chunk = c_hi * 8 + (((rem & 7) - g.tile[gf].rot) & 7);
    t = chunk < C ? gf + (rem >> 3) : -1;
  }
}
__device__ __forceinline__ void tn_decode(const TnPlan& g, int& t, int& chunk) { tn_decode_block(g, (int)blockIdx.x, t, chunk); }

__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(TnPlan g) {
  __shared__ __attribute__((aligned(16))) float smem[4 * T_TILE];
  float* As = smem;
  float* Bs = smem + 2 * T_TILE;

  const long long t_begin = g.dbg ? (long long)wall_clock64() : 0;
  const long long c_begin = g.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
  int t, chunk;
  tn_decode(g, t, chunk);
  if (t < 0) return;               // a hole of the XCD-aware order
  const TnTile tl = g.tile[t];
  const NudfGemmTNProblem& q = g.prob[tl.prob];
  const int slot_id = tl.blk_start + chunk;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i0 = tl.ti * BM, j0 = tl.tj * BN;
  const int lda = q.lda1, ldb = q.ldb1;
  const int mbeg = chunk * tl.rows_per_block;
  const int mend = min(mbeg + tl.rows_per_block, g.M);
  const int nk = (mend - mbeg + BK - 1) / BK;
  const int nfull = (mend - mbeg) / BK;

  // An operand stored as fp32: thread (t_k = tid / 32, t_c = 4 (tid % 32)) copies rows t_k + 8 ps (4 passes), 4 columns
  // each; stored as bf16 (config-5 mode, NUDF_TN_A16 / _B16): thread (tid / 16, 8 (tid % 16)) copies rows t_k + 16 ps
  // (2 passes), 8 columns each -- half the HBM bytes, widened to fp32 on the way into the same LDS tile.
  // Column indices are clamped into the buffer (columns past NA / NB only feed outputs that are never stored).  A FULL
  // k-step (all 32 rows exist -- every step but the last of the last row chunk) is plain loads off per-thread base
  // pointers and plain LDS stores; only the ragged step clamps rows and zeroes rows >= mend with selects.
  const bool a16 = (q.flags & NUDF_TN_A16) != 0, b16 = (q.flags & NUDF_TN_B16) != 0;
  // base pointer (bytes), row stride (bytes), byte offset of each of this thread's passes, bytes per k-step, thread row /
  // first column, LDS offset (floats) between passes
  struct Op { const char* p; size_t rowb, stepb; unsigned poff[4]; int tk, tc, lds_pass; bool blk; };
  auto mk = [&](const float* base, int ld, int c0, bool h, bool blk) {
    Op o;
    const int esz = h ? 2 : 4;
    o.rowb = (size_t)ld * esz;
    o.stepb = (size_t)BK * o.rowb;
    o.blk = blk;
    if (blk) {
      // fp32, blocked: [32-row block][quad of columns][row in block][4]; row chunks start on block boundaries.  Lane ->
      // row of the block, so that a wave instruction reads two whole quads = 1 KB of contiguous memory; the passes walk
      // the quads (8 per pass), i.e. 32 tile columns
      o.tk = tid & 31;
      o.tc = (tid >> 5) * 4;
      o.lds_pass = 32;
      o.p = reinterpret_cast<const char*>(base) + (size_t)mbeg * o.rowb + (size_t)o.tk * 16;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) o.poff[ps] = (unsigned)(min(c0 + o.tc + 32 * ps, ld - 4) >> 2) * 512u;
    } else {
      o.tk = h ? tid >> 4 : tid >> 5;
      o.tc = h ? (tid & 15) * 8 : (tid & 31) * 4;
      o.lds_pass = (h ? 16 : 8) * LDT;
      const int col = min(c0 + o.tc, ld - (h ? 8 : 4));
      o.p = reinterpret_cast<const char*>(base) + (size_t)(mbeg + o.tk) * o.rowb + (size_t)col * esz;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) o.poff[ps] = (unsigned)((size_t)ps * (h ? 16 : 8) * o.rowb);
    }
    return o;
  };
  const Op oa = mk(q.A1, lda, i0, a16, (q.flags & NUDF_TN_A_BLK) != 0), ob = mk(q.B1, ldb, j0, b16, (q.flags & NUDF_TN_B_BLK) != 0);
  f32x4 ra[4], rb[4];
  int ld_rows = 0;
  // the operand kind is a COMPILE-TIME parameter of these (dispatched once per workgroup below): with run-time kinds the
  // compiler put every load into its own branch and waited vmcnt(0) right behind it (1821 us instead of 650)
  auto load_op = [&](const Op& o, auto H, f32x4 (&r)[4], int kt, bool full) {
    constexpr bool h = decltype(H)::value;
    constexpr int np = h ? 2 : 4, rs = h ? 16 : 8;
    if (full || o.blk) {   // blocked buffers are padded to whole 32-row blocks: the ragged step needs no row clamp
      const char* b = o.p + (size_t)kt * o.stepb;
#pragma unroll
      for (int ps = 0; ps < np; ++ps) r[ps] = *reinterpret_cast<const f32x4*>(b + o.poff[ps]);
    } else {
#pragma unroll
      for (int ps = 0; ps < np; ++ps) {
        const ptrdiff_t row = min((ptrdiff_t)kt * BK + ps * rs, (ptrdiff_t)(g.M - 1 - mbeg - o.tk));   // clamp into the buffer
        r[ps] = *reinterpret_cast<const f32x4*>(o.p + row * (ptrdiff_t)o.rowb);
      }
    }
  };
  // bias gradient = column sums of A, taken from the registers on their way to LDS (tiles of the first tile column)
  const bool do_bias = (q.dbias != nullptr) && (tl.tj == 0) && !(g.flags & TNF_NO_BIAS);
  f32x4 bacc = {0.f, 0.f, 0.f, 0.f}, bacc2 = {0.f, 0.f, 0.f, 0.f};   // bacc2: columns 4..7 of a bf16 operand's 8
  f32x4 bacc3 = {0.f, 0.f, 0.f, 0.f}, bacc4 = {0.f, 0.f, 0.f, 0.f};   // blocked fp32 operand: one accumulator per pass
  auto widen = [](const f32x4& raw, f32x4& lo, f32x4& hi) {   // 8 bf16 -> 8 fp32 (memory order)
    const uint4 u = __builtin_bit_cast(uint4, raw);
    lo = f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
               __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    hi = f32x4{__builtin_bit_cast(float, u.z << 16), __builtin_bit_cast(float, u.z & 0xffff0000u),
               __builtin_bit_cast(float, u.w << 16), __builtin_bit_cast(float, u.w & 0xffff0000u)};
  };
  auto store_op = [&](const Op& o, auto H, const f32x4 (&r)[4], float* tile, bool bias) {
    constexpr bool h = decltype(H)::value;
    float* dst = tile + o.tk * LDT + o.tc;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!h) {
      if (ld_rows >= BK) {   // the common case: no selects
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<f32x4*>(dst + ps * o.lds_pass) = r[ps];
        if (bias) {
          if (o.blk) { bacc += r[0]; bacc2 += r[1]; bacc3 += r[2]; bacc4 += r[3]; }   // a pass = other columns
          else bacc += (r[0] + r[1]) + (r[2] + r[3]);                               // a pass = other rows
        }
        return;
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const f32x4 v = ((o.blk ? o.tk : ps * 8 + o.tk) < ld_rows) ? r[ps] : z;
        *reinterpret_cast<f32x4*>(dst + ps * o.lds_pass) = v;
        if (bias) {
          if (!o.blk || ps == 0) bacc += v;
          else if (ps == 1) bacc2 += v;
          else if (ps == 2) bacc3 += v;
          else bacc4 += v;
        }
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        f32x4 lo, hi;
        widen(r[ps], lo, hi);
        if (ld_rows < BK && (ps * 16 + o.tk) >= ld_rows) lo = hi = z;
        *reinterpret_cast<f32x4*>(dst + ps * 16 * LDT) = lo;
        *reinterpret_cast<f32x4*>(dst + ps * 16 * LDT + 4) = hi;
        if (bias) { bacc += lo; bacc2 += hi; }
      }
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;

  const bool live = tl.layout == 2 ? true : tl.layout == 0 ? (j0 + 32 * wave) < q.NB : (i0 + 32 * wave) < q.NA;
  const int n_w = live ? tl.n : 0;

  // one k-step of a wave: its FIXED operand (the sub-tile column it owns) against N sub-tiles of the other operand, two
  // reduction indices per MFMA, LDS reads software-pipelined four MFMA groups ahead
  auto mma = [&](auto N, auto LAY, int cur) {
    constexpr int kN = decltype(N)::value, kLay = decltype(LAY)::value;
    if constexpr (kLay == 2) {   // 64 x 64 quadrant: 2 + 2 operand reads per 4 MFMAs
      const float* as = As + cur * T_TILE + (lane >> 5) * LDT + (wave >> 1) * 64 + (lane & 31);
      const float* bs = Bs + cur * T_TILE + (lane >> 5) * LDT + (wave & 1) * 64 + (lane & 31);
      float av[2][4][2], bv[2][4][2];
      auto rd = [&](int set, int c) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int kk = c * 4 + qd;
#pragma unroll
          for (int i = 0; i < 2; ++i) av[set][qd][i] = as[(2 * kk) * LDT + 32 * i];
#pragma unroll
          for (int j = 0; j < 2; ++j) bv[set][qd][j] = bs[(2 * kk) * LDT + 32 * j];
        }
      };
      rd(0, 0);
#pragma unroll
      for (int c = 0; c < BK / 8; ++c) {
        if (c + 1 < BK / 8) rd((c + 1) & 1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][qd][i], bv[c & 1][qd][j], acc[i * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    const float* fx = (kLay == 0 ? Bs : As) + cur * T_TILE + (lane >> 5) * LDT + 32 * wave + (lane & 31);
    const float* vr = (kLay == 0 ? As : Bs) + cur * T_TILE + (lane >> 5) * LDT + (lane & 31);
    float fv[2][4], vv[2][4][4];
    auto rd = [&](int set, int c) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int kk = c * 4 + qd;
        fv[set][qd] = fx[(2 * kk) * LDT];
#pragma unroll
        for (int s = 0; s < kN; ++s) vv[set][qd][s] = vr[(2 * kk) * LDT + 32 * s];
      }
    };
    rd(0, 0);
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      if (c + 1 < BK / 8) rd((c + 1) & 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int s = 0; s < kN; ++s)
          acc[s] = kLay == 0 ? __builtin_amdgcn_mfma_f32_32x32x2f32(vv[c & 1][qd][s], fv[c & 1][qd], acc[s], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x2f32(fv[c & 1][qd], vv[c & 1][qd][s], acc[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // config-5 mode: bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Lane (i, h) of an operand holds the 8
  // reduction indices 8h .. 8h+7 of its column: 8 strided ds_read_b32 of the SAME fp32 LDS tiles, converted on the fly
  // (RNE).  16x the fp32 MFMA rate, so this loop is bound by the LDS reads / HBM, not by the matrix pipe.
  auto mma16 = [&](auto N, auto LAY, int cur) {
    constexpr int kN = decltype(N)::value, kLay = decltype(LAY)::value;
    if constexpr (kLay == 2) {
      const float* as = As + cur * T_TILE + (8 * (lane >> 5)) * LDT + (wave >> 1) * 64 + (lane & 31);
      const float* bs = Bs + cur * T_TILE + (8 * (lane >> 5)) * LDT + (wave & 1) * 64 + (lane & 31);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 a16[2], b16[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = as[(16 * kk + e) * LDT + 32 * i];
          a16[i] = __builtin_convertvector(v, bf16x8);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = bs[(16 * kk + e) * LDT + 32 * j];
          b16[j] = __builtin_convertvector(v, bf16x8);
        }
#pragma unroll
        for (int rep = 0; rep < NUDF_MMA16_REPEAT; ++rep)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16[i], b16[j], acc[i * 2 + j], 0, 0, 0);
      }
      return;
    }
    const float* fx = (kLay == 0 ? Bs : As) + cur * T_TILE + (8 * (lane >> 5)) * LDT + 32 * wave + (lane & 31);
    const float* vr = (kLay == 0 ? As : Bs) + cur * T_TILE + (8 * (lane >> 5)) * LDT + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f32x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fx[(16 * kk + e) * LDT];
      const bf16x8 f16 = __builtin_convertvector(v, bf16x8);
#pragma unroll
      for (int s = 0; s < kN; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vr[(16 * kk + e) * LDT + 32 * s];
        const bf16x8 v16 = __builtin_convertvector(v, bf16x8);
        for (int rep = 0; rep < NUDF_MMA16_REPEAT; ++rep)
        acc[s] = kLay == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(v16, f16, acc[s], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(f16, v16, acc[s], 0, 0, 0);
      }
    }
  };
  // bf16x3 mode (prec == 3, see NudfChainStep.prec): fp32 emulated on the bf16 pipe.  Both operand fragments are split
  // exactly into three bf16 parts on their way out of the SAME fp32 LDS image (8 ds_read_b32 + the split per fragment),
  // six partial products per sub-tile and k step, smallest first, fp32 accumulate.  Quadrant layout only (tn_plan).
  auto split3 = [](const f32x8& x, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    p0 = __builtin_convertvector(x, bf16x8);
    const f32x8 r1 = x - __builtin_convertvector(p0, f32x8);
    p1 = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(p1, f32x8);
    p2 = __builtin_convertvector(r2, bf16x8);
  };
  auto mma16x3 = [&](int cur) {
    const float* as = As + cur * T_TILE + (8 * (lane >> 5)) * LDT + (wave >> 1) * 64 + (lane & 31);
    const float* bs = Bs + cur * T_TILE + (8 * (lane >> 5)) * LDT + (wave & 1) * 64 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 a3[2][3], b3[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = as[(16 * kk + e) * LDT + 32 * i];
        split3(v, a3[i][0], a3[i][1], a3[i][2]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bs[(16 * kk + e) * LDT + 32 * j];
        split3(v, b3[j][0], b3[j][1], b3[j][2]);
      }
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pa = (t == 0 || t == 3 || t == 5) ? 0 : (t == 1 ? 2 : 1);     // h l m h m h
        const int pb = (t == 0) ? 2 : ((t == 2 || t == 3) ? 1 : 0);              // l h m m h h
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][pa], b3[j][pb], acc[i * 2 + j], 0, 0, 0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  // The loop of a FULL tile with fp32 operands and fp32 MFMAs -- 85 % of the step's weight-gradient work -- with every
  // wave's instruction stream arranged so that it never issues a long run of non-MFMA instructions
  // (scripts/ubench/mfma_pair.hip: the 64 operand reads in blocks of 16, the 8 LDS writes + wait in front of the barrier
  // and the 8 global loads at the top cost a wave pair 22 % of the matrix pipe; interleaved, 12 %):
  //   groups 0 .. 2 of 16 MFMAs, one LDS operand read (of the next group) behind every MFMA; the LDS writes of the NEXT
  //   k-step's operands spread over group 2; the barrier; group 3 -- its operands are in registers -- with the next
  //   k-step's first reads and the global loads of the k-step AFTER the next spread over it.
  // The steady-state body is branch-free (full k-steps only; the bias sums are always accumulated, one accumulator per
  // pass); the last iterations run the same pipeline through the generic load / store helpers.
  auto kloop_full = [&]() {
    const float* as0 = As + (lane >> 5) * LDT + (wave >> 1) * 64 + (lane & 31);
    const float* bs0 = Bs + (lane >> 5) * LDT + (wave & 1) * 64 + (lane & 31);
    float av[2][4][2], bv[2][4][2];
    const char* pa = oa.p;                       // this thread's operand rows of the k-step to LOAD next
    const char* pb = ob.p;
    float* const da0 = As + oa.tk * LDT + oa.tc;
    float* const db0 = Bs + ob.tk * LDT + ob.tc;
    auto gload = [&](int kt) {
      const bool full = kt < nfull;
      load_op(oa, std::false_type{}, ra, kt, full);
      load_op(ob, std::false_type{}, rb, kt, full);
      ld_rows = full ? BK : mend - (mbeg + kt * BK);
    };
    auto sstore = [&](int buf) {
      store_op(oa, std::false_type{}, ra, As + buf * T_TILE, do_bias);
      store_op(ob, std::false_type{}, rb, Bs + buf * T_TILE, false);
    };
    auto group = [&](auto C, auto FAST, int cur) {
      constexpr int c = decltype(C)::value;
      constexpr bool fast = decltype(FAST)::value;
      constexpr int ns = (c + 1) & 1, nc = (c + 1) & 3;
      const float* as = as0 + (c == 3 ? cur ^ 1 : cur) * T_TILE;
      const float* bs = bs0 + (c == 3 ? cur ^ 1 : cur) * T_TILE;
      float* da = da0 + (cur ^ 1) * T_TILE;
      float* db = db0 + (cur ^ 1) * T_TILE;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][qd][i], bv[c & 1][qd][j], acc[i * 2 + j], 0, 0, 0);
            const int r = j * 2 + i, kk = nc * 4 + qd;
            if (r < 2) av[ns][qd][r] = as[(2 * kk) * LDT + 32 * r];
            else bv[ns][qd][r - 2] = bs[(2 * kk) * LDT + 32 * (r - 2)];
            const int m = qd * 4 + r;
            if (fast && c == 2 && (m & 1)) {
              const int ps = (m >> 1) & 3;
              if (m < 8) {
                *reinterpret_cast<f32x4*>(da + ps * oa.lds_pass) = ra[ps];
                if (ps == 0) bacc += ra[0]; else if (ps == 1) bacc2 += ra[1]; else if (ps == 2) bacc3 += ra[2]; else bacc4 += ra[3];
              } else {
                *reinterpret_cast<f32x4*>(db + ps * ob.lds_pass) = rb[ps];
              }
            }
            if (fast && c == 3 && (m & 1)) {
              const int ps = (m >> 1) & 3;
              if (m < 8) ra[ps] = *reinterpret_cast<const f32x4*>(pa + oa.poff[ps]);
              else rb[ps] = *reinterpret_cast<const f32x4*>(pb + ob.poff[ps]);
            }
          }
      if constexpr (fast) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (c == 2 && (m & 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          if (c == 3 && (m & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    using F = std::true_type;
    using S = std::false_type;
    if (nk > 1) gload(1);   // run_kind has put k-step 0 into LDS buffer 0 (and passed the barrier behind it)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[0][qd][i] = as0[(2 * qd) * LDT + 32 * i];
        bv[0][qd][i] = bs0[(2 * qd) * LDT + 32 * i];
      }
    int kt = 0;
    if (2 < nfull) {
      pa += 2 * oa.stepb;
      pb += 2 * ob.stepb;
      for (; kt + 2 < nfull; ++kt) {   // k-steps kt + 1 (stored) and kt + 2 (loaded) are full
        const int cur = kt & 1;
        __builtin_amdgcn_sched_barrier(0);
        group(I0{}, F{}, cur);
        group(I1{}, F{}, cur);
        group(I2{}, F{}, cur);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        group(I3{}, F{}, cur);
        pa += oa.stepb;
        pb += ob.stepb;
      }
      ld_rows = BK;
    }
    for (; kt < nk; ++kt) {
      const int cur = kt & 1;
      group(I0{}, S{}, cur);
      group(I1{}, S{}, cur);
      group(I2{}, S{}, cur);
      if (kt + 1 < nk) sstore(cur ^ 1);
      __syncthreads();
      if (kt + 2 < nk) gload(kt + 2);
      group(I3{}, S{}, cur);
    }
    if (!oa.blk) {   // row-major A: the four passes are rows of the same columns
      bacc = (bacc + bacc2) + (bacc3 + bacc4);
      bacc2 = bacc3 = bacc4 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // Everything from the first operand load to the last k-step, for one (A kind, B kind).  The k-loop inside is
  // instantiated ONCE PER (sub-tile count, layout, precision), chosen outside the loop: a shape switch inside it made
  // the compiler copy all 64 accumulator registers (behind an MFMA drain) on every k-step.  bf16 operands and the 16-bit
  // MFMAs only ever meet the quadrant layout (tn_plan), so those kinds instantiate two loops each.
  auto run_kind = [&](auto KA, auto KB) {
    constexpr bool kA = decltype(KA)::value, kB = decltype(KB)::value;
    auto gload = [&](int kt) {
      const bool full = kt < nfull;
      load_op(oa, KA, ra, kt, full);
      load_op(ob, KB, rb, kt, full);
      ld_rows = full ? BK : mend - (mbeg + kt * BK);   // rows of this k-step that exist (the others are zeroed at the LDS store)
    };
    auto sstore = [&](int buf) {
      store_op(oa, KA, ra, As + buf * T_TILE, do_bias);
      store_op(ob, KB, rb, Bs + buf * T_TILE, false);
    };
    if (nk > 0) {
      gload(0);
      sstore(0);
    }
    __syncthreads();
    auto kloop = [&](auto N, auto LAY, auto P16) {
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the global loads above the MFMA block
        if constexpr (decltype(N)::value > 0) {
          if constexpr (decltype(P16)::value == 3) mma16x3(cur);
          else if constexpr (decltype(P16)::value != 0) mma16(N, LAY, cur);
          else mma(N, LAY, cur);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
      }
    };
    auto by_prec = [&](auto N, auto LAY) {
      if (g.prec == 3) kloop(N, LAY, I3{});
      else if (g.prec != 0) kloop(N, LAY, I1{});
      else kloop(N, LAY, I0{});
    };
    if (n_w == 0) { by_prec(I0{}, I0{}); return; }
    if constexpr (!kA && !kB) {
      if (tl.layout == 2 && g.prec == 0 && !(g.flags & TNF_NO_INTERLEAVE)) { kloop_full(); return; }
    }
    if (tl.layout == 2) { by_prec(I4{}, I2{}); return; }
    if constexpr (!kA && !kB) {
      if (tl.layout == 0) {
        if (n_w == 4) kloop(I4{}, I0{}, I0{});
        else if (n_w == 3) kloop(I3{}, I0{}, I0{});
        else if (n_w == 2) kloop(I2{}, I0{}, I0{});
        else kloop(I1{}, I0{}, I0{});
      } else {
        if (n_w == 3) kloop(I3{}, I1{}, I0{});
        else if (n_w == 2) kloop(I2{}, I1{}, I0{});
        else kloop(I1{}, I1{}, I0{});                 // n = 4 never takes this layout (ties go to layout 0)
      }
    }
  };
  if (!a16 && !b16) run_kind(std::false_type{}, std::false_type{});
  else if (a16 && b16) run_kind(std::true_type{}, std::true_type{});
  else if (a16) run_kind(std::true_type{}, std::false_type{});
  else run_kind(std::false_type{}, std::true_type{});

  if (g.dbg && tid == 0) {
    long long* d = g.dbg + 4 * (size_t)blockIdx.x;
    d[0] = t_begin; d[1] = (long long)wall_clock64();
    // tile kind | CU identity (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]) -> which workgroups shared a CU
    d[2] = (tl.layout * 16 + tl.n) | ((long long)(__builtin_amdgcn_s_getreg(63492) & 0xff00) << 8) |
           ((long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 32);
    d[3] = nk | (((long long)__builtin_amdgcn_s_memtime() - c_begin) << 16);   // shader-clock ticks of the same span
  }
  if (g.flags & TNF_NO_EPILOGUE) return;
  float* slot = g.ws ? g.ws + (size_t)slot_id * TN_WS_TILE : nullptr;
  if (do_bias) {   // the loop's last barrier has passed: the operand tiles are free
    int ngroups;
    if (oa.blk) {   // thread = (row of the block, quad): 32 partial rows x 128 columns
      ngroups = 32;
      float* d = smem + oa.tk * BM + oa.tc;
      *reinterpret_cast<f32x4*>(d) = bacc;
      *reinterpret_cast<f32x4*>(d + 32) = bacc2;
      *reinterpret_cast<f32x4*>(d + 64) = bacc3;
      *reinterpret_cast<f32x4*>(d + 96) = bacc4;
    } else {
      ngroups = a16 ? 16 : 8;
      *reinterpret_cast<f32x4*>(smem + oa.tk * BM + oa.tc) = bacc;
      if (a16) *reinterpret_cast<f32x4*>(smem + oa.tk * BM + oa.tc + 4) = bacc2;
    }
    __syncthreads();
    if (tid < BM) {
      float s = 0.0f;
      for (int k = 0; k < ngroups; ++k) s += smem[k * BM + tid];
      if (slot) slot[BM * BN + tid] = s;
      else if (i0 + tid < q.NA) atomicAdd(q.dbias + i0 + tid, s);
    }
  }
  if (n_w == 0) return;
  if (slot) {   // accumulator register order, 64 contiguous bytes per lane
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s >= n_w) break;
      float* w = slot + ((wave * 4 + s) * 64 + lane) * 16;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[s][4 * qd], acc[s][4 * qd + 1], acc[s][4 * qd + 2], acc[s][4 * qd + 3]};
        *reinterpret_cast<f32x4*>(w + 4 * qd) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s >= n_w) break;
    const int isub = tn_isub(tl.layout, wave, s), jsub = tn_jsub(tl.layout, wave, s);
    const int col = j0 + 32 * jsub + (lane & 31);
    if (col >= q.NB) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + 32 * isub + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < q.NA) atomicAdd(q.C + (size_t)row * q.ldc + col, acc[s][r]);
    }
  }
}

// second pass of the workspace path: C[tile] += sum over the tile's row chunks of the partial tiles, in chunk order
// (fixed association -> run-to-run identical), one float4 of accumulator registers per thread; slice 16 = the bias.
__global__ __launch_bounds__(256) void tn_reduce_kernel(TnPlan g) {
  const int t = blockIdx.x / 17, slice = blockIdx.x % 17;
  const TnTile tl = g.tile[t];
  const NudfGemmTNProblem& q = g.prob[tl.prob];
  const int chunks = g.tile[t + 1].blk_start - tl.blk_start;
  const int i0 = tl.ti * BM, j0 = tl.tj * BN;
  const float* ws = g.ws + (size_t)tl.blk_start * TN_WS_TILE;
  if (slice == 16) {
    if (!q.dbias || tl.tj != 0 || (g.flags & TNF_NO_BIAS)) return;
    const int col = threadIdx.x;
    if (col >= BM || i0 + col >= q.NA) return;
    float s = 0.0f;
    for (int c = 0; c < chunks; ++c) s += ws[(size_t)c * TN_WS_TILE + BM * BN + col];
    if (g.assign) q.dbias[i0 + col] = 0.0f + s;
    else q.dbias[i0 + col] += s;
    return;
  }
  const int e = (slice * 256 + threadIdx.x) * 4;            // first of 4 accumulator registers
  const int r0 = e & 15, lane = (e >> 4) & 63, s = (e >> 10) & 3, wave = e >> 12;
  const int isub = tn_isub(tl.layout, wave, s), jsub = tn_jsub(tl.layout, wave, s);
  const int row = i0 + 32 * isub + 8 * (r0 >> 2) + 4 * (lane >> 5);
  const int col = j0 + 32 * jsub + (lane & 31);
  // same liveness rule as the producer: sub-tiles wholly past NA / NB were never written
  if (i0 + 32 * isub >= q.NA || j0 + 32 * jsub >= q.NB || col >= q.NB) return;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int c = 0; c < chunks; ++c) sum += *reinterpret_cast<const f32x4*>(ws + (size_t)c * TN_WS_TILE + e);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (row + k < q.NA) {
      float* cp = q.C + (size_t)(row + k) * q.ldc + col;
      *cp = (g.assign ? 0.0f : *cp) + sum[k];
    }
}


// =======================================================================================================
// 16-bit MFMA mode (NudfGemmTNGroup.prec != 0, config 5): the same 128 x 128 tiles, quadrant layout, workspace slots
// and reduce, but k-steps of 64 rows and an LDS image that is ALREADY the MFMA operand: one dword = the bf16 values of
// rows (2 p, 2 p + 1) of one column, 32 such k-pair rows of LD16 dwords per operand and buffer.  Lane (i, h) of
// v_mfma_f32_32x32x16_bf16 needs rows 16 kk + 8 h .. + 7 of column i = 4 dwords a k-pair row apart: 4 ds_read_b32 (two
// ds_read2_b32) per sub-tile and MFMA, no conversion in the loop.  The generic kernel's 16-bit loop read the fp32 image
// instead: 8 ds_read_b32 + 4 v_cvt_pk per sub-tile and MFMA, and that -- not HBM -- bound it (2.07 ms at 1024 x 256).
// Rounding is unchanged: fp32 operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on the way INTO the image instead of
// out of it, bf16 operands are copied bit for bit, the MFMAs run over the same rows in the same order: bit-identical C.
// Thread (r = tid / 16, c = 8 (tid % 16)) stages k-pairs r and r + 16, 8 columns each: 2 x 2 rows of 16 B (bf16 operand)
// or 32 B (fp32 operand).
// A 4-POINT PACKED bf16 operand (NUDF_TN_A_P4 / _B_P4: the chains' 16-bit stored state) already holds the image's dwords:
// the 8 bytes of (point quad, column) are the k-pairs (2 quad, 2 quad + 1) of that column.  Thread (quad = tid / 32 + 8 ps,
// t = tid % 32) loads columns 2 t, 2 t + 1 and 64 + 2 t, 65 + 2 t (16 B each: a wave instruction reads 2 x 512 contiguous
// bytes) and stores the even dwords to one image row, the odd ones to the next -- no bit shuffling at all (the row-major
// bf16 operand needs 16 and/shift/or per 8 values).
// =======================================================================================================
#define BK16 64
#define LD16 132
#define T16 (32 * LD16)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int KIND> struct Tn16Stage { f32x4 v[2][2][KIND ? 1 : 2]; };   // [pass][row of the pair (P4: column pair)][16-byte piece]

__global__ __launch_bounds__(256, 2) void gemm_tn16_group_kernel(TnPlan g) {
  __shared__ __attribute__((aligned(16))) unsigned smem[4 * T16];
  unsigned* As = smem;
  unsigned* Bs = smem + 2 * T16;

  const long long t_begin = g.dbg ? (long long)wall_clock64() : 0;
  const long long c_begin = g.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
  int t, chunk;
  tn_decode(g, t, chunk);
  if (t < 0) return;               // a hole of the XCD-aware order
  const TnTile tl = g.tile[t];
  const NudfGemmTNProblem& q = g.prob[tl.prob];
  const int slot_id = tl.blk_start + chunk;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i0 = tl.ti * BM, j0 = tl.tj * BN;
  const int mbeg = chunk * tl.rows_per_block;
  const int mend = min(mbeg + tl.rows_per_block, g.M);
  const int nk = (mend - mbeg + BK16 - 1) / BK16;
  const int pr = tid >> 4, pc = (tid & 15) * 8;
  const bool do_bias = (q.dbias != nullptr) && (tl.tj == 0) && !(g.flags & TNF_NO_BIAS);
  f32x4 bias_lo = {0.f, 0.f, 0.f, 0.f}, bias_hi = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;

  // operand kinds: 0 = fp32 row-major, 1 = bf16 row-major, 2 = bf16 4-point packed
  auto run_kind = [&](auto KA, auto KB) {
    constexpr int kA = decltype(KA)::value, kB = decltype(KB)::value;
    // this thread's 8 columns of an operand, clamped into the buffer (columns past NA / NB only feed outputs that are
    // never stored); rows are clamped per load
    // (a bf16 operand's leading dimension is a multiple of 8, an fp32 one's of 4: the two 4-column pieces of an fp32
    // row are clamped separately)
    const int ca = i0 + pc, cb = j0 + pc;
    // (P4: this thread's column pairs p4c and 64 + p4c, 8 bytes per column and point quad; a packed operand's leading
    // dimension is a multiple of 8, so clamping a pair to ld - 2 keeps it inside the row)
    const int p4c = (tid & 31) * 2, p4q = tid >> 5;
    const char* pa = (kA == 2) ? reinterpret_cast<const char*>(q.A1) + (size_t)min(i0 + p4c, q.lda1 - 2) * 8
                               : reinterpret_cast<const char*>(q.A1) + (size_t)min(ca, q.lda1 - (kA ? 8 : 4)) * (kA ? 2 : 4);
    const char* pb = (kB == 2) ? reinterpret_cast<const char*>(q.B1) + (size_t)min(j0 + p4c, q.ldb1 - 2) * 8
                               : reinterpret_cast<const char*>(q.B1) + (size_t)min(cb, q.ldb1 - (kB ? 8 : 4)) * (kB ? 2 : 4);
    const int pa2x = (kA == 2) ? (min(i0 + p4c + 64, q.lda1 - 2) - min(i0 + p4c, q.lda1 - 2)) * 8 : 0;
    const int pb2x = (kB == 2) ? (min(j0 + p4c + 64, q.ldb1 - 2) - min(j0 + p4c, q.ldb1 - 2)) * 8 : 0;
    const int pa2 = (min(ca + 4, q.lda1 - 4) - min(ca, q.lda1 - 4)) * 4;   // byte offset of an fp32 row's second piece
    const int pb2 = (min(cb + 4, q.ldb1 - 4) - min(cb, q.ldb1 - 4)) * 4;
    // bytes per row (P4: per point quad)
    const size_t rowa = (size_t)q.lda1 * (kA == 2 ? 8 : kA ? 2 : 4), rowb = (size_t)q.ldb1 * (kB == 2 ? 8 : kB ? 2 : 4);
    Tn16Stage<kA> sa;
    Tn16Stage<kB> sb;
    auto load = [&](auto Hc, const char* p, int p2, size_t rowbytes, auto& st, int kt) {
      constexpr int h = decltype(Hc)::value;
      if constexpr (h == 2) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int quad = min(((mbeg + kt * BK16) >> 2) + p4q + 8 * ps, (g.M - 1) >> 2);
          const char* src = p + (size_t)quad * rowbytes;
          st.v[ps][0][0] = *reinterpret_cast<const f32x4*>(src);
          st.v[ps][1][0] = *reinterpret_cast<const f32x4*>(src + p2);
        }
        return;
      }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int row = min(mbeg + kt * BK16 + 2 * (pr + 16 * ps) + rr, g.M - 1);
          const char* src = p + (size_t)row * rowbytes;
          st.v[ps][rr][0] = *reinterpret_cast<const f32x4*>(src);
          if constexpr (h == 0) st.v[ps][rr][1] = *reinterpret_cast<const f32x4*>(src + p2);
        }
    };
    auto widen = [](const f32x4& raw, f32x4& lo, f32x4& hi) {
      const uint4 u = __builtin_bit_cast(uint4, raw);
      lo = f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                 __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
      hi = f32x4{__builtin_bit_cast(float, u.z << 16), __builtin_bit_cast(float, u.z & 0xffff0000u),
                 __builtin_bit_cast(float, u.w << 16), __builtin_bit_cast(float, u.w & 0xffff0000u)};
    };
    auto pack2 = [](float a, float b) {
      return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    };
    auto store = [&](auto Hc, const auto& st, int kt, unsigned* tile, bool bias) {
      constexpr int h = decltype(Hc)::value;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if constexpr (h == 2) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int qd = p4q + 8 * ps;                                // point quad of the k-step = image rows 2 qd, 2 qd + 1
          const int nv = mend - (mbeg + kt * BK16 + 4 * qd);          // live points of the quad (rows >= mend are zeros)
          const unsigned m01 = nv >= 2 ? 0xffffffffu : (nv == 1 ? 0x0000ffffu : 0u);
          const unsigned m23 = nv >= 4 ? 0xffffffffu : (nv == 3 ? 0x0000ffffu : 0u);
          const uint4 a = __builtin_bit_cast(uint4, st.v[ps][0][0]), b = __builtin_bit_cast(uint4, st.v[ps][1][0]);
          // columns p4c, p4c + 1 | 64 + p4c, 65 + p4c
          const u32x4 even = {a.x & m01, a.z & m01, b.x & m01, b.z & m01};
          const u32x4 odd = {a.y & m23, a.w & m23, b.y & m23, b.w & m23};
          if (bias) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              bias_lo[c] += (__builtin_bit_cast(float, even[c] << 16) + __builtin_bit_cast(float, even[c] & 0xffff0000u)) +
                            (__builtin_bit_cast(float, odd[c] << 16) + __builtin_bit_cast(float, odd[c] & 0xffff0000u));
          }
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          unsigned* dst = tile + (2 * qd) * LD16 + p4c;
          *reinterpret_cast<u32x2*>(dst) = u32x2{even[0], even[1]};
          *reinterpret_cast<u32x2*>(dst + 64) = u32x2{even[2], even[3]};
          *reinterpret_cast<u32x2*>(dst + LD16) = u32x2{odd[0], odd[1]};
          *reinterpret_cast<u32x2*>(dst + LD16 + 64) = u32x2{odd[2], odd[3]};
        }
        return;
      }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int k0 = mbeg + kt * BK16 + 2 * (pr + 16 * ps);
        const bool v0 = k0 < mend, v1 = k0 + 1 < mend;
        u32x4 olo, ohi;
        if constexpr (h == 1) {
          const f32x4 r0 = v0 ? st.v[ps][0][0] : z, r1 = v1 ? st.v[ps][1][0] : z;
          const uint4 a = __builtin_bit_cast(uint4, r0), b = __builtin_bit_cast(uint4, r1);
          olo = u32x4{(a.x & 0xffffu) | (b.x << 16), (a.x >> 16) | (b.x & 0xffff0000u),
                      (a.y & 0xffffu) | (b.y << 16), (a.y >> 16) | (b.y & 0xffff0000u)};
          ohi = u32x4{(a.z & 0xffffu) | (b.z << 16), (a.z >> 16) | (b.z & 0xffff0000u),
                      (a.w & 0xffffu) | (b.w << 16), (a.w >> 16) | (b.w & 0xffff0000u)};
          if (bias) {
            f32x4 l0, h0, l1, h1;
            widen(r0, l0, h0);
            widen(r1, l1, h1);
            bias_lo += l0 + l1;
            bias_hi += h0 + h1;
          }
        } else {
          const f32x4 r0l = v0 ? st.v[ps][0][0] : z, r0h = v0 ? st.v[ps][0][1] : z;
          const f32x4 r1l = v1 ? st.v[ps][1][0] : z, r1h = v1 ? st.v[ps][1][1] : z;
          olo = u32x4{pack2(r0l[0], r1l[0]), pack2(r0l[1], r1l[1]), pack2(r0l[2], r1l[2]), pack2(r0l[3], r1l[3])};
          ohi = u32x4{pack2(r0h[0], r1h[0]), pack2(r0h[1], r1h[1]), pack2(r0h[2], r1h[2]), pack2(r0h[3], r1h[3])};
          if (bias) {
            bias_lo += r0l + r1l;
            bias_hi += r0h + r1h;
          }
        }
        unsigned* dst = tile + (pr + 16 * ps) * LD16 + pc;
        *reinterpret_cast<u32x4*>(dst) = olo;
        *reinterpret_cast<u32x4*>(dst + 4) = ohi;
      }
    };
    // one k-step: 4 groups of 16 rows, per group 2 + 2 operand sub-tiles and 4 MFMAs; operand reads one group ahead
    auto mma = [&](int cur) {
      const unsigned* as = As + cur * T16 + (4 * (lane >> 5)) * LD16 + (wave >> 1) * 64 + (lane & 31);
      const unsigned* bs = Bs + cur * T16 + (4 * (lane >> 5)) * LD16 + (wave & 1) * 64 + (lane & 31);
      u32x4 a[2][2], b[2][2];
      auto rd = [&](int set, int kk) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[set][s2][e] = as[(8 * kk + e) * LD16 + 32 * s2];
            b[set][s2][e] = bs[(8 * kk + e) * LD16 + 32 * s2];
          }
      };
      rd(0, 0);
#pragma unroll
      for (int kk = 0; kk < BK16 / 16; ++kk) {
        if (kk + 1 < BK16 / 16) rd((kk + 1) & 1, kk + 1);
#pragma unroll
        for (int rep = 0; rep < NUDF_MMA16_REPEAT; ++rep)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk & 1][i]),
                                                                     __builtin_bit_cast(bf16x8, b[kk & 1][j]), acc[i * 2 + j], 0, 0, 0);
      }
    };
    if (nk > 0) {
      load(KA, pa, kA == 2 ? pa2x : pa2, rowa, sa, 0);
      load(KB, pb, kB == 2 ? pb2x : pb2, rowb, sb, 0);
      store(KA, sa, 0, As, do_bias);
      store(KB, sb, 0, Bs, false);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) {
        load(KA, pa, kA == 2 ? pa2x : pa2, rowa, sa, kt + 1);
        load(KB, pb, kB == 2 ? pb2x : pb2, rowb, sb, kt + 1);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the global loads above the MFMA block
      mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        store(KA, sa, kt + 1, As + (cur ^ 1) * T16, do_bias);
        store(KB, sb, kt + 1, Bs + (cur ^ 1) * T16, false);
      }
      __syncthreads();
    }
  };
  const int ka = (q.flags & NUDF_TN_A16) ? ((q.flags & NUDF_TN_A_P4) ? 2 : 1) : 0;
  const int kb = (q.flags & NUDF_TN_B16) ? ((q.flags & NUDF_TN_B_P4) ? 2 : 1) : 0;
  typedef std::integral_constant<int, 0> K0;
  typedef std::integral_constant<int, 1> K1;
  typedef std::integral_constant<int, 2> K2;
  // (row-major and packed bf16 meet only across problems of a group; tn_plan rejects the pair inside one problem)
  if (ka == 2 && kb == 2) run_kind(K2{}, K2{});
  else if (ka == 2) run_kind(K2{}, K0{});
  else if (kb == 2) run_kind(K0{}, K2{});
  else if (ka == 1 && kb == 1) run_kind(K1{}, K1{});
  else if (ka == 0 && kb == 0) run_kind(K0{}, K0{});
  else if (ka == 1) run_kind(K1{}, K0{});
  else run_kind(K0{}, K1{});
  const bool a_p4 = ka == 2;

  if (g.dbg && tid == 0) {
    long long* d = g.dbg + 4 * (size_t)blockIdx.x;
    d[0] = t_begin; d[1] = (long long)wall_clock64(); d[2] = 2 * 16 + 4;
    d[3] = nk | (((long long)__builtin_amdgcn_s_memtime() - c_begin) << 16);
  }
  if (g.flags & TNF_NO_EPILOGUE) return;
  float* slot = g.ws ? g.ws + (size_t)slot_id * TN_WS_TILE : nullptr;
  if (do_bias) {   // the loop's last barrier has passed: the operand image is free
    float* red = reinterpret_cast<float*>(smem);
    if (a_p4) {   // 8 thread rows x (2 + 2) columns each: partial sums in rows 0..7, zeros in rows 8..15 of the 16-row table
      float* r0 = red + (tid >> 5) * BM + (tid & 31) * 2;
      r0[0] = bias_lo[0]; r0[1] = bias_lo[1]; r0[64] = bias_lo[2]; r0[65] = bias_lo[3];
      r0[8 * BM] = 0.0f; r0[8 * BM + 1] = 0.0f; r0[8 * BM + 64] = 0.0f; r0[8 * BM + 65] = 0.0f;
    } else {
      *reinterpret_cast<f32x4*>(red + pr * BM + pc) = bias_lo;
      *reinterpret_cast<f32x4*>(red + pr * BM + pc + 4) = bias_hi;
    }
    __syncthreads();
    if (tid < BM) {
      float sum = 0.0f;
      for (int k = 0; k < 16; ++k) sum += red[k * BM + tid];
      if (slot) slot[BM * BN + tid] = sum;
      else if (i0 + tid < q.NA) atomicAdd(q.dbias + i0 + tid, sum);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (slot) {   // accumulator register order, 64 contiguous bytes per lane (tn_reduce_kernel decodes it)
      float* w = slot + ((wave * 4 + s) * 64 + lane) * 16;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[s][4 * qd], acc[s][4 * qd + 1], acc[s][4 * qd + 2], acc[s][4 * qd + 3]};
        *reinterpret_cast<f32x4*>(w + 4 * qd) = v;
      }
    } else {
      const int col = j0 + 32 * tn_jsub(2, wave, s) + (lane & 31);
      if (col >= q.NB) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 32 * tn_isub(2, wave, s) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < q.NA) atomicAdd(q.C + (size_t)row * q.ldc + col, acc[s][r]);
      }
    }
  }
}

// =======================================================================================================
// bf16x3 mode (NudfGemmTNGroup.prec == 3): fp32 EMULATED on the bf16 matrix pipe, see NudfChainStep.prec.  Same tiles,
// quadrant layout, workspace slots and reduce as above; both operands are fp32 row-major (the bf16x3 chains keep fp32
// state).  The LDS image is the MFMA operand THREE TIMES -- planes hi / mid / lo -- filled by splitting every element ONCE on
// its way in (x = hi + mid + lo exactly; 11 VALU operations per pair of values), so the k loop holds no conversion at all:
// per 16 rows and sub-tile pair 3 + 3 operand fragments (one ds_read_b128 each, layout in the kernel) and 6 MFMAs (hi lo,
// lo hi, mid mid, hi mid, mid hi, hi hi; fp32 accumulate).  k-steps of 32 rows, ONE 48 KB buffer: the global loads of step
// k + 2 are in flight under step k's 48 MFMAs per wave, the split + LDS stores sit between two barriers, and the other
// workgroup of the CU runs its MFMA phase meanwhile.  A wave-specialised form (one 8-wave workgroup per CU: four matrix
// waves, four staging waves, double-buffered image, one barrier per step) was built and measured slower in every variant
// (1.30-1.51 ms against 1.01-1.15 ms per step; profiles/r04_bf16x3_experiments.txt item 7) and removed.
// =======================================================================================================
#ifndef NUDF_TN3_DIST2
#define NUDF_TN3_DIST2 1
#endif
#ifndef NUDF_TN3_PIPE
#define NUDF_TN3_PIPE 1      // A/B build switch: 1 = the split of the next k-step interleaved with this step's MFMAs (see pstep)
#endif
#ifndef NUDF_TN3_LDSPREAD
#define NUDF_TN3_LDSPREAD 0  // A/B build switch (with NUDF_TN3_PIPE): 1 = the row requests of step kt + 2 issued between the MFMA groups (measured 8 % slower: the requests are throttled by the memory path wherever they are issued)
#endif
#ifndef NUDF_TN3_BUFLOAD
#define NUDF_TN3_BUFLOAD 1   // A/B build switch: 1 = the steady state's row requests are buffer loads (scalar row offset, 32-bit lane offset)
#endif
#ifndef NUDF_TN3_STAMPS
#define NUDF_TN3_STAMPS 0
#endif
#ifndef NUDF_TN3_WGS
#define NUDF_TN3_WGS 2       // workgroups per CU the split-image kernel is register-allocated for (3: 13 spilled registers)
#endif
#define BK3 32
#define LD3 132
#define T3 (16 * LD3)
#define T3Q (4 * 128 * 4)    // dwords per plane and operand of the [k-pair group][column][4] image
__device__ __forceinline__ void tn_split3_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, p0 << 16);
  const float r1 = x1 - __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, p1 << 16);
  const float s1 = r1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

__global__ __launch_bounds__(256, NUDF_TN3_WGS) void gemm_tn3_group_kernel(TnPlan g) {
  // image: [operand 2][plane 3][k-pair group 4][column 128][4 dwords] -- the four k-pairs (8 rows) a lane's MFMA operand
  // needs for one column are ONE 16-byte unit: a fragment is one conflict-free ds_read_b128 (the [k-pair][column] layout of
  // the 16-bit kernel costs four ds_read_b32 at half the LDS rate, and its 32-byte-per-lane staging writes hit every bank
  // twice: at six planes the loop was LDS-bound, ~2 000 LDS cycles per k-step against 1 536 of MFMA).  A staging thread owns
  // ONE column and 16 rows (two k-pair groups): 16 coalesced 4-byte loads per operand (a wave = 256 contiguous bytes of a
  // row), 8 pair splits, and per plane two 16-byte stores at a 16-byte lane stride.
  __shared__ __attribute__((aligned(16))) unsigned smem[6 * T3Q];
  unsigned* As = smem;             // planes at As + pl * T3Q
  unsigned* Bs = smem + 3 * T3Q;

  const long long t_begin = g.dbg ? (long long)wall_clock64() : 0;
  const long long c_begin = g.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
  int t, chunk;
  tn_decode(g, t, chunk);
  if (t < 0) return;               // a hole of the XCD-aware order
  const TnTile tl = g.tile[t];
  const NudfGemmTNProblem& q = g.prob[tl.prob];
  const int slot_id = tl.blk_start + chunk;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i0 = tl.ti * BM, j0 = tl.tj * BN;
  const int mbeg = chunk * tl.rows_per_block;
  const int mend = min(mbeg + tl.rows_per_block, g.M);
  const int nk = (mend - mbeg + BK3 - 1) / BK3;
  const int sc = tid & 127, sg = tid >> 7;          // staging: column of the tile, half of the k-step (rows 16 sg .. + 15)
  const bool do_bias = (q.dbias != nullptr) && (tl.tj == 0) && !(g.flags & TNF_NO_BIAS);
  float bias_acc = 0.0f;

  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;

  // this thread's column of each operand, clamped into the buffer (columns past NA / NB only feed outputs that are never
  // stored); rows are clamped per load and zeroed past mend at the split
  const float* pa = q.A1 + min(i0 + sc, q.lda1 - 1);
  const float* pb = q.B1 + min(j0 + sc, q.ldb1 - 1);
  const size_t lda = (size_t)q.lda1, ldb = (size_t)q.ldb1;
  auto load = [&](const float* p, size_t ld, float (&st)[16], int kt) {
    const int r0 = mbeg + kt * BK3 + 16 * sg;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = p[(size_t)min(r0 + r, g.M - 1) * ld];
  };
  auto store = [&](const float (&st)[16], int kt, unsigned* tile, bool bias) {
    const int r0 = mbeg + kt * BK3 + 16 * sg;
    u32x4 hq[2], mq[2], lq[2];
    float ps[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      const float x0 = (r0 + 2 * pp < mend) ? st[2 * pp] : 0.0f;
      const float x1 = (r0 + 2 * pp + 1 < mend) ? st[2 * pp + 1] : 0.0f;
      unsigned a, b, d;
      tn_split3_pair(x0, x1, a, b, d);
      hq[pp >> 2][pp & 3] = a; mq[pp >> 2][pp & 3] = b; lq[pp >> 2][pp & 3] = d;
      ps[pp] = x0 + x1;
    }
    // bias gradient: the step's 16 rows summed as a tree, then one addition into the running sum (a plain running sum over
    // all rows of the chunk was 10x the exact kernel's error against float64)
    if (bias) bias_acc += ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
    unsigned* dst = tile + ((2 * sg) * 128 + sc) * 4;
    *reinterpret_cast<u32x4*>(dst) = hq[0];
    *reinterpret_cast<u32x4*>(dst + 512) = hq[1];
    *reinterpret_cast<u32x4*>(dst + T3Q) = mq[0];
    *reinterpret_cast<u32x4*>(dst + T3Q + 512) = mq[1];
    *reinterpret_cast<u32x4*>(dst + 2 * T3Q) = lq[0];
    *reinterpret_cast<u32x4*>(dst + 2 * T3Q + 512) = lq[1];
  };
  // one k-step: 2 groups of 16 rows; per group 2 + 2 operand sub-tiles x 3 planes (one ds_read_b128 each) and 24 MFMAs
  auto mma = [&]() {
    const unsigned* as = As + ((lane >> 5) * 128 + (wave >> 1) * 64 + (lane & 31)) * 4;
    const unsigned* bs = Bs + ((lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31)) * 4;
#pragma unroll
    for (int kk = 0; kk < BK3 / 16; ++kk) {
      u32x4 a[2][3], b[2][3];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          a[s2][pl] = *reinterpret_cast<const u32x4*>(as + pl * T3Q + (2 * kk) * 512 + 128 * s2);
          b[s2][pl] = *reinterpret_cast<const u32x4*>(bs + pl * T3Q + (2 * kk) * 512 + 128 * s2);
        }
#pragma unroll
      for (int tt = 0; tt < 6; ++tt) {
        const int qa = (tt == 0 || tt == 3 || tt == 5) ? 0 : (tt == 1 ? 2 : 1);     // h l m h m h
        const int qb = (tt == 0) ? 2 : ((tt == 2 || tt == 3) ? 1 : 0);              // l h m m h h
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][qa]),
                                                                     __builtin_bit_cast(bf16x8, b[j][qb]), acc[i * 2 + j], 0, 0, 0);
      }
    }
  };
  // the operand rows of k-step kt + 2 are requested before step kt's MFMAs (two staging sets): a panel's first reader
  // among the tiles of its XCD pays an HBM round trip, which is longer than one step's 48 MFMAs per wave
  float sa[16], sb[16], sa2[16], sb2[16];
  if (nk > 0) {
    load(pa, lda, sa, 0);
    load(pb, ldb, sb, 0);
    if (nk > 1) {
      load(pa, lda, sa2, 1);
      load(pb, ldb, sb2, 1);
    }
    store(sa, 0, As, do_bias);
    store(sb, 0, Bs, false);
  }
  __syncthreads();
  auto kstep = [&](int kt, float (&la)[16], float (&lb)[16], float (&ua)[16], float (&ub)[16]) {
    // la / lb: free set, receives step kt + 2; ua / ub: holds step kt + 1 (requested one step ago), stored after the MFMAs
#if NUDF_TN3_DIST2
    if (kt + 2 < nk) {
      load(pa, lda, la, kt + 2);
      load(pb, ldb, lb, kt + 2);
    }
#else      // A/B: requests one step ahead only (into the set that is stored after this step's MFMAs)
    if (kt + 1 < nk && kt > 0) {
      load(pa, lda, ua, kt + 1);
      load(pb, ldb, ub, kt + 1);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);   // keep the global loads above the MFMA block
    mma();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // every wave is done reading the image
    if (kt + 1 < nk) {
      store(ua, kt + 1, As, do_bias);
      store(ub, kt + 1, Bs, false);
    }
    __syncthreads();
  };
  // ---- steady state (round 5): FULL k-steps with loop-invariant addressing.  The generic `load` forms every address with a
  // row clamp, a 64-bit multiply-add and a 64-bit shift-add per element (4 VALU operations per 4-byte load, 128 per thread and
  // step -- as many as the split), and hipcc, recycling those address registers, put an s_waitcnt vmcnt(0) in front of every
  // step's requests: the "two steps ahead" prefetch waited for the previous step's rows first.  Here a step's rows start at a
  // wave-uniform offset formed by the scalar unit and a thread's only address register is its column offset: a load is ONE
  // instruction (buffer_load_dword v, v_col, s[descriptor], s_row offen; NUDF_TN3_BUFLOAD), no vector address arithmetic, nothing
  // to wait for; full steps also need no row-validity selects in the split.  (The first form of this loop used flat loads from
  // `base + r * ld`: hipcc kept those bases in VECTOR registers -- a chain of 30 v_lshl_add_u64 per step, still 39 % fewer VALU
  // operations than the generic staging; the descriptor form is the one without any.)  Same values, same order: C and dbias are
  // unchanged bit for bit.
  int kt0 = 0;
#if NUDF_TN3_STAMPS     // tuning build (scripts/tn3_phases.py): shader-clock ticks of waves 0 / 3 per segment of the pipelined steps
  long long tk_load = 0, tk_mma = 0, tk_bar1 = 0, tk_store = 0, tk_bar2 = 0;
#endif
  if (!(g.flags & TNF_GENERIC_STAGE)) {
    const bool last_ragged = ((mend - mbeg) % BK3) != 0 || mend > g.M;
    const int n_fast = nk - 2 - (last_ragged ? 1 : 0);          // steps kt with kt + 1 and kt + 2 full and inside the chunk
    // (buffer descriptors hold a 32-bit byte count)
    if (n_fast >= 2 && (size_t)(g.M - mbeg) * (size_t)max(q.lda1, q.ldb1) * 4 < ((size_t)1 << 31)) {
      // (16 sg is wave-uniform: waves 0-1 stage rows 0..15 of a step, waves 2-3 rows 16..31)
      const int sgu = __builtin_amdgcn_readfirstlane(sg);
      const unsigned ca = (unsigned)min(i0 + sc, q.lda1 - 1), cbb = (unsigned)min(j0 + sc, q.ldb1 - 1);
#if NUDF_TN3_BUFLOAD
      // BUFFER loads: address = descriptor base (the chunk's first row) + a 32-bit lane offset (the column) + a scalar offset (the
      // row): `buffer_load_dword v, v_col, s[rsrc], s_row offen` -- no vector address arithmetic at all.  (The flat form of the
      // same loop, `(base + r * ld)[col]`, compiled to a chain of 30 v_lshl_add_u64 and 64-bit per-lane addresses.)
      const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.A1 + (size_t)mbeg * lda), 0,
                                                                           (int)((size_t)(g.M - mbeg) * lda * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.B1 + (size_t)mbeg * ldb), 0,
                                                                           (int)((size_t)(g.M - mbeg) * ldb * 4), 0x00020000);
      const int lda4 = q.lda1 * 4, ldb4 = q.ldb1 * 4;
      int ba = (2 * BK3 + 16 * sgu) * lda4, bb = (2 * BK3 + 16 * sgu) * ldb4;       // scalar byte offsets of the rows of step kt + 2
      const int sa_step = BK3 * lda4, sb_step = BK3 * ldb4;
      auto load_f = [&](__amdgpu_buffer_rsrc_t rs, int so, int ld4, unsigned col, float (&st)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          st[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(col * 4), so + r * ld4, 0));
      };
#define TN3_LOAD_A(st) load_f(rsa, ba, lda4, ca, st)
#define TN3_LOAD_B(st) load_f(rsb, bb, ldb4, cbb, st)
#else
      auto load_f = [&](const float* __restrict__ base, size_t ld, unsigned col, float (&st)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = (base + (size_t)r * ld)[col];     // scalar row base + one 32-bit lane offset
      };
#define TN3_LOAD_A(st) load_f(ba, lda, ca, st)
#define TN3_LOAD_B(st) load_f(bb, ldb, cbb, st)
#endif
      auto store_f = [&](const float (&st)[16], unsigned* tile, bool bias) {
        u32x4 hq[2], mq[2], lq[2];
        float ps[8];
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
          unsigned a, b, d;
          tn_split3_pair(st[2 * pp], st[2 * pp + 1], a, b, d);
          hq[pp >> 2][pp & 3] = a; mq[pp >> 2][pp & 3] = b; lq[pp >> 2][pp & 3] = d;
          ps[pp] = st[2 * pp] + st[2 * pp + 1];
        }
        if (bias) bias_acc += ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
        unsigned* dst = tile + ((2 * sg) * 128 + sc) * 4;
        *reinterpret_cast<u32x4*>(dst) = hq[0];
        *reinterpret_cast<u32x4*>(dst + 512) = hq[1];
        *reinterpret_cast<u32x4*>(dst + T3Q) = mq[0];
        *reinterpret_cast<u32x4*>(dst + T3Q + 512) = mq[1];
        *reinterpret_cast<u32x4*>(dst + 2 * T3Q) = lq[0];
        *reinterpret_cast<u32x4*>(dst + 2 * T3Q + 512) = lq[1];
      };
      // uniform bases of the rows of step kt + 2 (scalar registers; the readfirstlane only tells the compiler so)
#if !NUDF_TN3_BUFLOAD
      const float* ba = q.A1 + (size_t)(mbeg + 2 * BK3 + 16 * sgu) * lda;
      const float* bb = q.B1 + (size_t)(mbeg + 2 * BK3 + 16 * sgu) * ldb;
      const size_t sa_step = (size_t)BK3 * lda, sb_step = (size_t)BK3 * ldb;
#endif
      auto fstep = [&](float (&la)[16], float (&lb)[16], float (&ua)[16], float (&ub)[16]) {
        TN3_LOAD_A(la);
        TN3_LOAD_B(lb);
        ba += sa_step;
        bb += sb_step;
        __builtin_amdgcn_sched_barrier(0);
        mma();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        store_f(ua, As, do_bias);
        store_f(ub, Bs, false);
        __syncthreads();
      };
      // PIPELINED form (build switch NUDF_TN3_PIPE, scripts/build_variants.sh): the split of step kt + 1 -- pure register arithmetic -- is
      // interleaved with step kt's MFMAs inside the wave (four VALU operations behind every MFMA, pinned with
      // sched_group_barrier), so that what is left between the two barriers is twelve LDS stores: the stretch in which this
      // workgroup cannot issue an MFMA shrinks from the whole split to the stores.  Same values, same order: bit-identical.
      auto store_r = [&](const u32x4 (&pl)[3][2], unsigned* tile) {
        unsigned* dst = tile + ((2 * sg) * 128 + sc) * 4;
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          *reinterpret_cast<u32x4*>(dst + p3 * T3Q) = pl[p3][0];
          *reinterpret_cast<u32x4*>(dst + p3 * T3Q + 512) = pl[p3][1];
        }
      };
      const unsigned* fas = As + ((lane >> 5) * 128 + (wave >> 1) * 64 + (lane & 31)) * 4;
      const unsigned* fbs = Bs + ((lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31)) * 4;
#if NUDF_TN3_STAMPS
#define TN3_STAMP(acc_) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); acc_ += n_ - tk_last; tk_last = n_; }
#else
#define TN3_STAMP(acc_)
#endif
      auto pstep = [&](float (&la)[16], float (&lb)[16], float (&ua)[16], float (&ub)[16]) {
#if NUDF_TN3_STAMPS
        long long tk_last = (long long)__builtin_amdgcn_s_memtime();
#endif
#if !NUDF_TN3_LDSPREAD
        TN3_LOAD_A(la);
        TN3_LOAD_B(lb);
        ba += sa_step;
        bb += sb_step;
        __builtin_amdgcn_sched_barrier(0);
#endif
        TN3_STAMP(tk_load)
        // 12 groups of 4 MFMAs (2 groups of 16 rows x 6 plane pairs); behind each of the first eight, the split of TWO row
        // pairs of the next step (22 VALU operations + the bias partial), pinned in source order by sched_barrier: a wave
        // issues its four MFMAs (4 x 32 pipe cycles) and splits while they execute
        u32x4 pa3[3][2], pb3[3][2];
        float ps[8];
#pragma unroll
        for (int kk = 0; kk < BK3 / 16; ++kk) {
          u32x4 a[2][3], b[2][3];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              a[s2][pl] = *reinterpret_cast<const u32x4*>(fas + pl * T3Q + (2 * kk) * 512 + 128 * s2);
              b[s2][pl] = *reinterpret_cast<const u32x4*>(fbs + pl * T3Q + (2 * kk) * 512 + 128 * s2);
            }
#pragma unroll
          for (int tt = 0; tt < 6; ++tt) {
            const int qa = (tt == 0 || tt == 3 || tt == 5) ? 0 : (tt == 1 ? 2 : 1);     // h l m h m h
            const int qb = (tt == 0) ? 2 : ((tt == 2 || tt == 3) ? 1 : 0);              // l h m m h h
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int i = 0; i < 2; ++i)
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][qa]),
                                                                         __builtin_bit_cast(bf16x8, b[j][qb]), acc[i * 2 + j], 0, 0, 0);
            const int c = kk * 6 + tt;           // chunk 0..11: chunks 0..3 split A's pairs 2c, 2c + 1, chunks 4..7 B's
            if (c < 8) {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int pp = 2 * (c & 3) + e;
                unsigned x, y, z;
                if (c < 4) {
                  tn_split3_pair(ua[2 * pp], ua[2 * pp + 1], x, y, z);
                  // (an opaque use HERE: the split is pure arithmetic and LLVM otherwise sinks it to its consumers, the LDS
                  // stores behind the barrier, before the machine scheduler ever sees the sched_barriers)
                  asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
                  pa3[0][pp >> 2][pp & 3] = x; pa3[1][pp >> 2][pp & 3] = y; pa3[2][pp >> 2][pp & 3] = z;
                  ps[pp] = ua[2 * pp] + ua[2 * pp + 1];
                } else {
                  tn_split3_pair(ub[2 * pp], ub[2 * pp + 1], x, y, z);
                  asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
                  pb3[0][pp >> 2][pp & 3] = x; pb3[1][pp >> 2][pp & 3] = y; pb3[2][pp >> 2][pp & 3] = z;
                }
              }
            }
#if NUDF_TN3_LDSPREAD && !NUDF_TN3_BUFLOAD
            // the 32 row requests of step kt + 2, spread over the twelve groups as well (3 3 3 3 3 3 3 3 2 2 2 2, in the order
            // the next step's split consumes them): issued in one burst at the top of the step they held the wave for ~1 800
            // cycles per step before its first MFMA (8 waves x 32 requests x 256 B against a 64 B / clk vector cache path)
            if (c < 8) {
              la[2 * c] = (ba + (size_t)(2 * c) * lda)[ca];
              la[2 * c + 1] = (ba + (size_t)(2 * c + 1) * lda)[ca];
              lb[c] = (bb + (size_t)c * ldb)[cbb];
            } else {
              lb[2 * c - 8] = (bb + (size_t)(2 * c - 8) * ldb)[cbb];
              lb[2 * c - 7] = (bb + (size_t)(2 * c - 7) * ldb)[cbb];
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#if NUDF_TN3_LDSPREAD
        ba += sa_step;
        bb += sb_step;
#endif
        // (a select, not a branch; the same tree and running sum as store_f)
        const float bsum = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
        bias_acc = do_bias ? bias_acc + bsum : bias_acc;
        __builtin_amdgcn_sched_barrier(0);
        TN3_STAMP(tk_mma)
        __syncthreads();
        TN3_STAMP(tk_bar1)
        store_r(pa3, As);
        store_r(pb3, Bs);
        TN3_STAMP(tk_store)
        __syncthreads();
        TN3_STAMP(tk_bar2)
      };
#undef TN3_STAMP
#undef TN3_LOAD_A
#undef TN3_LOAD_B
      // (n_fast >= 2 here: a do-while keeps the two steps of an iteration in straight-line code -- with a for loop and both
      // forms selectable at run time hipcc rotated the loop BETWEEN a step's MFMAs and its split, i.e. put them in different
      // basic blocks, where nothing can be interleaved)
#if NUDF_TN3_PIPE
      do {
        pstep(sa, sb, sa2, sb2);
        pstep(sa2, sb2, sa, sb);
        kt0 += 2;
      } while (kt0 + 1 < n_fast);
#else
      (void)pstep;
      do {
        fstep(sa, sb, sa2, sb2);
        fstep(sa2, sb2, sa, sb);
        kt0 += 2;
      } while (kt0 + 1 < n_fast);
#endif
    }
  }
  for (int kt = kt0; kt < nk; kt += 2) {
    kstep(kt, sa, sb, sa2, sb2);
    if (kt + 1 < nk) kstep(kt + 1, sa2, sb2, sa, sb);
  }

#if NUDF_TN3_STAMPS
  if (g.dbg && lane == 0 && (wave == 0 || wave == 3)) {
    long long* d = g.dbg + 16 * (size_t)blockIdx.x + (wave ? 8 : 0);
    d[0] = t_begin; d[1] = (long long)wall_clock64();
    d[2] = nk | ((long long)kt0 << 20) | (((long long)__builtin_amdgcn_s_memtime() - c_begin) << 40);
    d[3] = tk_load; d[4] = tk_mma; d[5] = tk_bar1; d[6] = tk_store; d[7] = tk_bar2;
  }
#else
  if (g.dbg && tid == 0) {
    long long* d = g.dbg + 4 * (size_t)blockIdx.x;
    d[0] = t_begin; d[1] = (long long)wall_clock64(); d[2] = 2 * 16 + 4;
    d[3] = nk | (((long long)__builtin_amdgcn_s_memtime() - c_begin) << 16);
  }
#endif
  if (g.flags & TNF_NO_EPILOGUE) return;
  float* slot = g.ws ? g.ws + (size_t)slot_id * TN_WS_TILE : nullptr;
  if (do_bias) {   // the loop's last barrier has passed: the operand image is free
    float* red = reinterpret_cast<float*>(smem);
    red[sg * BM + sc] = bias_acc;
    __syncthreads();
    if (tid < BM) {
      const float sum = red[tid] + red[BM + tid];
      if (slot) slot[BM * BN + tid] = sum;
      else if (i0 + tid < q.NA) atomicAdd(q.dbias + i0 + tid, sum);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (slot) {   // accumulator register order, 64 contiguous bytes per lane (tn_reduce_kernel decodes it)
      float* w = slot + ((wave * 4 + s) * 64 + lane) * 16;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[s][4 * qd], acc[s][4 * qd + 1], acc[s][4 * qd + 2], acc[s][4 * qd + 3]};
        *reinterpret_cast<f32x4*>(w + 4 * qd) = v;
      }
    } else {
      const int col = j0 + 32 * tn_jsub(2, wave, s) + (lane & 31);
      if (col >= q.NB) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 32 * tn_isub(2, wave, s) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < q.NA) atomicAdd(q.C + (size_t)row * q.ldc + col, acc[s][r]);
      }
    }
  }
}

// =======================================================================================================
// f16x2 mode (NudfGemmTNGroup.prec == 4, round 6): fp32 emulated with THREE fp16 MFMA products per fragment pair instead of
// bf16x3's six (NudfChainStep.prec 4 has the arithmetic: x = hi + 2^-11 lo in fp16, acc0 += hi hi', acc1 += lo hi' + hi lo',
// result acc0 + 2^-11 acc1).  The kernel above runs against the 1 400 W package limit, so the gain is the products not formed:
// measured on the step's three launches 412 / 365 / 216 -> 323 / 275 / 202 us, step 3.57 -> 3.44 ms as a pair on one box.
// (A timing probe that formed three of the bf16x3 kernel's six products had promised 3.74 -> 3.21 ms: its wrong gradients had
// degraded the weights it was then timed on -- 1 068 W -- so that number was not a bound of anything.)
// fp16's exponent range is the price: one operand of every weight-gradient problem holds adjoints of the LOSS (1e-6 ... 1e-9).
// Each side of a group therefore carries a power-of-two scale taken from a device scalar, max |x| over that side's operands
// (written by the chain sweeps that produce them, NudfChain.absmax_out): sigma = 2^(10 - floor(log2 max)), so the largest
// element lands in [1024, 2048), elements down to 2^-25 of it keep all 22 bits, and C is multiplied by 1 / (sigma_a sigma_b)
// on its way out -- exact.  A scaled element beyond +-60 000 (a producer that did not report its maximum) is clamped, not inf.
// Same tiles, image layout (two planes instead of three: 32 KB), k-steps, workspace slots and reduce as the bf16x3 kernel;
// the staging is its plain form (split + LDS stores between the barriers): two accumulator sets leave no registers for the
// interleaved one.
// =======================================================================================================
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void tn_split2_pair(float x0, float x1, float sc, unsigned& p0, unsigned& p1) {
  x0 = __builtin_fminf(__builtin_fmaxf(x0 * sc, -60000.0f), 60000.0f);
  x1 = __builtin_fminf(__builtin_fmaxf(x1 * sc, -60000.0f), 60000.0f);
  const f16x2_t h = __builtin_convertvector(f32x2{x0, x1}, f16x2_t);
  p0 = __builtin_bit_cast(unsigned, h);
  const float r0 = (x0 - (float)h[0]) * 2048.0f, r1 = (x1 - (float)h[1]) * 2048.0f;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2_t));
}
// 2^(10 - floor(log2 m)) and its reciprocal from the operand maximum (1 for NULL, 0, denormal or non-finite maxima)
__device__ __forceinline__ void tn_scale_of(const float* amax, float& sc, float& inv) {
  sc = 1.0f; inv = 1.0f;
  if (!amax) return;
  const unsigned e = (__builtin_bit_cast(unsigned, *amax) >> 23) & 0xffu;     // biased exponent of the maximum
  if (e == 0u || e == 255u) return;
  const int se = 127 + 10 - ((int)e - 127);                                    // biased exponent of sigma
  if (se < 1 || se > 254) return;
  sc = __builtin_bit_cast(float, (unsigned)se << 23);
  inv = __builtin_bit_cast(float, (unsigned)(254 - se) << 23);
}

#ifndef NUDF_TN2_DB
#define NUDF_TN2_DB 0      // A/B build switch, see the kernel's shared-memory comment (measured: no gain for the step)
#endif
__global__ __launch_bounds__(256, 2) void gemm_tn2_group_kernel(TnPlan g) {
  // image: [operand 2][plane 2][k-pair group 4][column 128][4 dwords], as gemm_tn3_group_kernel's with two planes
  // NUDF_TN2_DB=1 (A/B build, not the default): TWO images (64 KB, still two workgroups per CU) -- step kt's MFMAs read image
  // kt & 1 while step kt + 1 is split into the other one, so a k-step has ONE barrier and a wave's split / LDS stores run beside
  // the other waves' MFMAs; same arithmetic, bit-identical results, 250 VGPRs.  MEASURED (profiles/r06_tn2_double_buffer_ab.txt):
  // the kernel alone 5-10 % faster (331 / 319 / 200 vs 346-355 / 316 / 201 us), the train step SLOWER (3.56-3.57 vs 3.52 ms): the
  // card answers the denser kernel with a lower clock for the whole replayed step (2 008 vs 2 110 MHz at 1 276 vs 1 310 W).
  __shared__ __attribute__((aligned(16))) unsigned smem[(NUDF_TN2_DB ? 8 : 4) * T3Q];
  constexpr int IMG = NUDF_TN2_DB ? 4 * T3Q : 0;      // dwords from one image to the other
  unsigned* As = smem;
  unsigned* Bs = smem + 2 * T3Q;

  int t, chunk;
  tn_decode(g, t, chunk);
  if (t < 0) return;               // a hole of the XCD-aware order
  const TnTile tl = g.tile[t];
  const NudfGemmTNProblem& q = g.prob[tl.prob];
  const int slot_id = tl.blk_start + chunk;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i0 = tl.ti * BM, j0 = tl.tj * BN;
  const int mbeg = chunk * tl.rows_per_block;
  const int mend = min(mbeg + tl.rows_per_block, g.M);
  const int nk = (mend - mbeg + BK3 - 1) / BK3;
  const int sc = tid & 127, sg = tid >> 7;          // staging: column of the tile, half of the k-step (rows 16 sg .. + 15)
  const bool do_bias = (q.dbias != nullptr) && (tl.tj == 0) && !(g.flags & TNF_NO_BIAS);
  float bias_acc = 0.0f;
  float sca, inva, scb, invb;
  tn_scale_of(g.amax_a, sca, inva);
  tn_scale_of(g.amax_b, scb, invb);

  f32x16 acc[4], acc1[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[s4][r] = 0.0f; acc1[s4][r] = 0.0f; }

  const float* pa = q.A1 + min(i0 + sc, q.lda1 - 1);
  const float* pb = q.B1 + min(j0 + sc, q.ldb1 - 1);
  const size_t lda = (size_t)q.lda1, ldb = (size_t)q.ldb1;
  auto load = [&](const float* p, size_t ld, float (&st)[16], int kt) {
    const int r0 = mbeg + kt * BK3 + 16 * sg;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = p[(size_t)min(r0 + r, g.M - 1) * ld];
  };
  // rows >= mend are zeroed (FULL: a step that lies inside the chunk needs no row tests); bias sums from the UNSCALED values
  auto store = [&](auto FULLc, const float (&st)[16], int kt, unsigned* tile, float scale, bool bias) {
    constexpr bool FULL = decltype(FULLc)::value;
    const int r0 = mbeg + kt * BK3 + 16 * sg;
    u32x4 hq[2], lq[2];
    float ps[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      const float x0 = (FULL || r0 + 2 * pp < mend) ? st[2 * pp] : 0.0f;
      const float x1 = (FULL || r0 + 2 * pp + 1 < mend) ? st[2 * pp + 1] : 0.0f;
      unsigned a, b;
      tn_split2_pair(x0, x1, scale, a, b);
      hq[pp >> 2][pp & 3] = a; lq[pp >> 2][pp & 3] = b;
      ps[pp] = x0 + x1;
    }
    if (bias) bias_acc += ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
    unsigned* dst = tile + ((2 * sg) * 128 + sc) * 4;
    *reinterpret_cast<u32x4*>(dst) = hq[0];
    *reinterpret_cast<u32x4*>(dst + 512) = hq[1];
    *reinterpret_cast<u32x4*>(dst + T3Q) = lq[0];
    *reinterpret_cast<u32x4*>(dst + T3Q + 512) = lq[1];
  };
  // one k-step: 2 groups of 16 rows; per group 2 + 2 operand sub-tiles x 2 planes (one ds_read_b128 each) and 12 MFMAs
  auto mma = [&](int img) {
    const unsigned* as = As + img + ((lane >> 5) * 128 + (wave >> 1) * 64 + (lane & 31)) * 4;
    const unsigned* bs = Bs + img + ((lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31)) * 4;
#pragma unroll
    for (int kk = 0; kk < BK3 / 16; ++kk) {
      u32x4 a[2][2], b[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          a[s2][pl] = *reinterpret_cast<const u32x4*>(as + pl * T3Q + (2 * kk) * 512 + 128 * s2);
          b[s2][pl] = *reinterpret_cast<const u32x4*>(bs + pl * T3Q + (2 * kk) * 512 + 128 * s2);
        }
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {          // lo hi', hi lo' (their own accumulator), hi hi'
        const int qa = (tt == 0) ? 1 : 0, qb = (tt == 1) ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (tt < 2)
              acc1[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[i][qa]),
                                                                       __builtin_bit_cast(f16x8_t, b[j][qb]), acc1[i * 2 + j], 0, 0, 0);
            else
              acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[i][qa]),
                                                                      __builtin_bit_cast(f16x8_t, b[j][qb]), acc[i * 2 + j], 0, 0, 0);
          }
      }
    }
  };
  typedef std::integral_constant<bool, false> Ragged;
  typedef std::integral_constant<bool, true> Full;
  float sa[16], sb[16], sa2[16], sb2[16];
  if (nk > 0) {
    load(pa, lda, sa, 0);
    load(pb, ldb, sb, 0);
    if (nk > 1) {
      load(pa, lda, sa2, 1);
      load(pb, ldb, sb2, 1);
    }
    store(Ragged{}, sa, 0, As, sca, do_bias);
    store(Ragged{}, sb, 0, Bs, scb, false);
  }
  __syncthreads();
  auto kstep = [&](int kt, float (&la)[16], float (&lb)[16], float (&ua)[16], float (&ub)[16]) {
    if (kt + 2 < nk) {
      load(pa, lda, la, kt + 2);
      load(pb, ldb, lb, kt + 2);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the global loads above the MFMA block
    if (NUDF_TN2_DB) {
      const int nxt = ((kt + 1) & 1) * IMG;
      if (kt + 1 < nk) {                 // (the other image: every wave left it at the previous step's barrier)
        store(Ragged{}, ua, kt + 1, As + nxt, sca, do_bias);
        store(Ragged{}, ub, kt + 1, Bs + nxt, scb, false);
      }
      mma((kt & 1) * IMG);
      __syncthreads();
    } else {
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                     // every wave is done reading the image
      if (kt + 1 < nk) {
        store(Ragged{}, ua, kt + 1, As, sca, do_bias);
        store(Ragged{}, ub, kt + 1, Bs, scb, false);
      }
      __syncthreads();
    }
  };
  // steady state: FULL k-steps, buffer loads with a scalar row offset (gemm_tn3_group_kernel's loop-invariant addressing)
  int kt0 = 0;
  {
    const bool last_ragged = ((mend - mbeg) % BK3) != 0 || mend > g.M;
    const int n_fast = nk - 2 - (last_ragged ? 1 : 0);          // steps kt with kt + 1 and kt + 2 full and inside the chunk
    if (n_fast >= 2 && (size_t)(g.M - mbeg) * (size_t)max(q.lda1, q.ldb1) * 4 < ((size_t)1 << 31)) {
      const int sgu = __builtin_amdgcn_readfirstlane(sg);
      const unsigned ca = (unsigned)min(i0 + sc, q.lda1 - 1), cbb = (unsigned)min(j0 + sc, q.ldb1 - 1);
      const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.A1 + (size_t)mbeg * lda), 0,
                                                                           (int)((size_t)(g.M - mbeg) * lda * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.B1 + (size_t)mbeg * ldb), 0,
                                                                           (int)((size_t)(g.M - mbeg) * ldb * 4), 0x00020000);
      const int lda4 = q.lda1 * 4, ldb4 = q.ldb1 * 4;
      int ba = (2 * BK3 + 16 * sgu) * lda4, bb = (2 * BK3 + 16 * sgu) * ldb4;       // scalar byte offsets of the rows of step kt + 2
      const int sa_step = BK3 * lda4, sb_step = BK3 * ldb4;
      auto load_f = [&](__amdgpu_buffer_rsrc_t rs, int so, int ld4, unsigned col, float (&st)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          st[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(col * 4), so + r * ld4, 0));
      };
      auto fstep = [&](auto CURc, float (&la)[16], float (&lb)[16], float (&ua)[16], float (&ub)[16]) {
        constexpr int cur = decltype(CURc)::value * IMG, nxt = IMG - cur;       // the loop runs an even number of steps from kt0 = 0
        load_f(rsa, ba, lda4, ca, la);
        load_f(rsb, bb, ldb4, cbb, lb);
        ba += sa_step;
        bb += sb_step;
        __builtin_amdgcn_sched_barrier(0);
        if (NUDF_TN2_DB) {
          store(Full{}, ua, 0, As + nxt, sca, do_bias);
          store(Full{}, ub, 0, Bs + nxt, scb, false);
          mma(cur);
          __syncthreads();
        } else {
          mma(0);
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          store(Full{}, ua, 0, As, sca, do_bias);
          store(Full{}, ub, 0, Bs, scb, false);
          __syncthreads();
        }
      };
      do {
        fstep(std::integral_constant<int, 0>{}, sa, sb, sa2, sb2);
        fstep(std::integral_constant<int, 1>{}, sa2, sb2, sa, sb);
        kt0 += 2;
      } while (kt0 + 1 < n_fast);
    }
  }
  for (int kt = kt0; kt < nk; kt += 2) {
    kstep(kt, sa, sb, sa2, sb2);
    if (kt + 1 < nk) kstep(kt + 1, sa2, sb2, sa, sb);
  }

  if (g.flags & TNF_NO_EPILOGUE) return;
  float* slot = g.ws ? g.ws + (size_t)slot_id * TN_WS_TILE : nullptr;
  if (do_bias) {   // the loop's last barrier has passed: the operand image is free
    float* red = reinterpret_cast<float*>(smem);
    red[sg * BM + sc] = bias_acc;
    __syncthreads();
    if (tid < BM) {
      const float sum = red[tid] + red[BM + tid];
      if (slot) slot[BM * BN + tid] = sum;
      else if (i0 + tid < q.NA) atomicAdd(q.dbias + i0 + tid, sum);
    }
  }
  const float unscale = inva * invb;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s4][r] = __builtin_fmaf(acc1[s4][r], 1.0f / 2048.0f, acc[s4][r]) * unscale;
    if (slot) {   // accumulator register order, 64 contiguous bytes per lane (tn_reduce_kernel decodes it)
      float* w = slot + ((wave * 4 + s4) * 64 + lane) * 16;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[s4][4 * qd], acc[s4][4 * qd + 1], acc[s4][4 * qd + 2], acc[s4][4 * qd + 3]};
        *reinterpret_cast<f32x4*>(w + 4 * qd) = v;
      }
    } else {
      const int col = j0 + 32 * tn_jsub(2, wave, s4) + (lane & 31);
      if (col >= q.NB) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 32 * tn_isub(2, wave, s4) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < q.NA) atomicAdd(q.C + (size_t)row * q.ldc + col, acc[s4][r]);
      }
    }
  }
}

// =======================================================================================================
// bf16x3 mode, WIDE form (round 5; OPT-IN through NUDF_TN_FLAGS bit 1024 -- it measured slower): one
// 8-wave workgroup per CU computes TWO vertically adjacent 128 x 128 tiles of a problem -- a 256 x 128 block: 25 % less
// operand traffic per flop than two 128 x 128 workgroups (the B panel is staged once) -- from a DOUBLE-BUFFERED split image
// (2 x 72 KB), so a k-step has ONE barrier: step kt's MFMAs read buffer kt & 1 while step kt + 1 is split into the other one.
// The two waves of every SIMD run the two halves of a step in OPPOSITE order -- waves 0..3 multiply first and stage
// afterwards, waves 4..7 stage first and multiply afterwards (both orders are legal between the same two barriers) -- so that
// one wave's split / LDS stores run beside its partner's 48 MFMAs.
// Same 32-row k-steps, same MFMA order per accumulator, same workspace slots (a wave's 64 x 64 quadrant is written where the
// 128 x 128 kernel's wave writes it) and the same fixed-order reduce: on equal row chunks C and dbias are BIT-IDENTICAL to
// gemm_tn3_group_kernel's (tests/test_gpu_bf16x3.py).  The plan keeps its 128 x 128 tiles; TnPlan.partner pairs them and
// TnPlan.unit_tile enumerates the workgroups.
// MEASURED (profiles/r05_tn_wide.txt): 511-541 us against 467-495 us for the UDF adjoint group at 65 536 points.  Per k-step
// and wave (nudf_set_tn_debug, shader-clock ticks): MFMA segment 1.8-2.1 k (48 MFMAs = 1.5 k of pipe), split + LDS stores
// 2.0-3.3 k, load issue 0.8-1.7 k, barrier wait 1.1-3.1 k = 8.1 k per step where the pipe needs 3.1 k -- the staging of a
// step (24 four-byte loads with clamped 64-bit addresses, 12 pair splits of 11 VALU operations, 9 LDS stores per thread) costs
// a wave more issue time than its 48 MFMAs, and two such waves per SIMD do not hide each other's; the 128 x 128 kernel's two
// independent workgroups per CU interleave better than one barrier domain of eight waves.  What would move both kernels is
// less staging work per MFMA (a 256 x 256 block per workgroup needs 128 accumulator registers per wave: no room beside the
// staging sets), not a different overlap of the same work.
// =======================================================================================================
#define T3WA (4 * 256 * 4)   // dwords per plane of the A image [k-pair group 4][column 256][4]
#define T3WB (4 * 128 * 4)   // ... of the B image
#define T3W_BUF (3 * T3WA + 3 * T3WB)
__global__ __launch_bounds__(512, 1) void gemm_tn3w_group_kernel(TnPlan g) {
  __shared__ __attribute__((aligned(16))) unsigned smem_w[2 * T3W_BUF];   // 144 KB: one workgroup per CU
  const int t = g.unit_tile[blockIdx.x / g.unit_chunks];
  const int chunk = blockIdx.x % g.unit_chunks;
  const int partner = g.partner[t];
  const TnTile tl = g.tile[t];
  const NudfGemmTNProblem& q = g.prob[tl.prob];
  const bool two = partner >= 0;
  const int slot0 = tl.blk_start + chunk;
  const int slot1 = two ? g.tile[partner].blk_start + chunk : slot0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;                       // 0..7: A quadrant wave >> 1 (64 columns), B half wave & 1
  const int i0 = tl.ti * BM, j0 = tl.tj * BN;
  const int mbeg = chunk * tl.rows_per_block;
  const int mend = min(mbeg + tl.rows_per_block, g.M);
  const int nk = (mend - mbeg + BK3 - 1) / BK3;
  const int sca = tid & 255, sga = tid >> 8;       // A staging: column of the 256-wide panel, rows 16 sga .. + 15 of the step
  const int scb = tid & 127, sgb = tid >> 7;       // B staging: column, rows 8 sgb .. + 7 (one k-pair group)
  const bool bias_tile = (q.dbias != nullptr) && (tl.tj == 0) && !(g.flags & TNF_NO_BIAS);
  const bool do_bias = bias_tile && (two || sca < BM);
  float bias_acc = 0.0f;
  const bool mm_live = two || wave < 4;            // an unpaired tile: waves 4..7 only stage

  f32x16 acc[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s2][r] = 0.0f;

  const float* pa = q.A1 + min(i0 + sca, q.lda1 - 1);
  const float* pb = q.B1 + min(j0 + scb, q.ldb1 - 1);
  const size_t lda = (size_t)q.lda1, ldb = (size_t)q.ldb1;
  auto load_a = [&](float (&st)[16], int kt) {
    const int r0 = mbeg + kt * BK3 + 16 * sga;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = pa[(size_t)min(r0 + r, g.M - 1) * lda];
  };
  auto load_b = [&](float (&st)[8], int kt) {
    const int r0 = mbeg + kt * BK3 + 8 * sgb;
#pragma unroll
    for (int r = 0; r < 8; ++r) st[r] = pb[(size_t)min(r0 + r, g.M - 1) * ldb];
  };
  auto store_a = [&](const float (&st)[16], int kt, unsigned* img) {
    const int r0 = mbeg + kt * BK3 + 16 * sga;
    u32x4 hq[2], mq[2], lq[2];
    float ps[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      const float x0 = (r0 + 2 * pp < mend) ? st[2 * pp] : 0.0f;
      const float x1 = (r0 + 2 * pp + 1 < mend) ? st[2 * pp + 1] : 0.0f;
      unsigned a, b, d;
      tn_split3_pair(x0, x1, a, b, d);
      hq[pp >> 2][pp & 3] = a; mq[pp >> 2][pp & 3] = b; lq[pp >> 2][pp & 3] = d;
      ps[pp] = x0 + x1;
    }
    // (the same tree per 16 rows and the same running sum as gemm_tn3_group_kernel: identical bias gradients)
    if (do_bias) bias_acc += ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
    unsigned* dst = img + ((2 * sga) * 256 + sca) * 4;
    *reinterpret_cast<u32x4*>(dst) = hq[0];
    *reinterpret_cast<u32x4*>(dst + 1024) = hq[1];
    *reinterpret_cast<u32x4*>(dst + T3WA) = mq[0];
    *reinterpret_cast<u32x4*>(dst + T3WA + 1024) = mq[1];
    *reinterpret_cast<u32x4*>(dst + 2 * T3WA) = lq[0];
    *reinterpret_cast<u32x4*>(dst + 2 * T3WA + 1024) = lq[1];
  };
  auto store_b = [&](const float (&st)[8], int kt, unsigned* img) {
    const int r0 = mbeg + kt * BK3 + 8 * sgb;
    u32x4 hq, mq, lq;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const float x0 = (r0 + 2 * pp < mend) ? st[2 * pp] : 0.0f;
      const float x1 = (r0 + 2 * pp + 1 < mend) ? st[2 * pp + 1] : 0.0f;
      unsigned a, b, d;
      tn_split3_pair(x0, x1, a, b, d);
      hq[pp] = a; mq[pp] = b; lq[pp] = d;
    }
    unsigned* dst = img + 3 * T3WA + (sgb * 128 + scb) * 4;
    *reinterpret_cast<u32x4*>(dst) = hq;
    *reinterpret_cast<u32x4*>(dst + T3WB) = mq;
    *reinterpret_cast<u32x4*>(dst + 2 * T3WB) = lq;
  };
  // one k-step of MFMAs: the 12 operand fragments of the SECOND group of 16 rows are requested before the first group's 24
  // MFMAs (two register sets), so only the first group's LDS latency is exposed per step
  auto mma = [&](const unsigned* img) {
    const unsigned* as = img + ((lane >> 5) * 256 + (wave >> 1) * 64 + (lane & 31)) * 4;
    const unsigned* bs = img + 3 * T3WA + ((lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31)) * 4;
    u32x4 fa[2][2][3], fb[2][2][3];
    auto ldf = [&](int kk) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          fa[kk][s2][pl] = *reinterpret_cast<const u32x4*>(as + pl * T3WA + (2 * kk) * 1024 + 128 * s2);
          fb[kk][s2][pl] = *reinterpret_cast<const u32x4*>(bs + pl * T3WB + (2 * kk) * 512 + 128 * s2);
        }
    };
    ldf(0);
#pragma unroll
    for (int kk = 0; kk < BK3 / 16; ++kk) {
      if (kk + 1 < BK3 / 16) ldf(kk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < 6; ++tt) {
        const int qa = (tt == 0 || tt == 3 || tt == 5) ? 0 : (tt == 1 ? 2 : 1);     // h l m h m h
        const int qb = (tt == 0) ? 2 : ((tt == 2 || tt == 3) ? 1 : 0);              // l h m m h h
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kk][i][qa]),
                                                                     __builtin_bit_cast(bf16x8, fb[kk][j][qb]), acc[i * 2 + j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  unsigned* buf0 = smem_w;
  unsigned* buf1 = smem_w + T3W_BUF;
  float sa[16], sb[8], sa2[16], sb2[8];
  if (nk > 0) {
    load_a(sa, 0);
    load_b(sb, 0);
    if (nk > 1) {
      load_a(sa2, 1);
      load_b(sb2, 1);
    }
    store_a(sa, 0, buf0);
    store_b(sb, 0, buf0);
  }
  const bool mma_first = wave < 4;
  // tuning (nudf_set_tn_debug): shader-clock ticks this wave spent in its three segments and waiting at the barrier
  long long tk_mma = 0, tk_store = 0, tk_load = 0, tk_bar = 0;
  const bool prof = g.dbg != nullptr;
  const long long tk_begin = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define TNW_STAMP(acc_) if (prof) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); acc_ += n_ - tk_last; tk_last = n_; }
  if (!mma_first && nk > 2) {     // the staging-first waves run one request further ahead (see kstep)
    load_a(sa, 2);
    load_b(sb, 2);
  }
  __syncthreads();
  auto kstep = [&](int kt, float (&la)[16], float (&lb)[8], float (&ua)[16], float (&ub)[8], const unsigned* cur, unsigned* nxt) {
    // la / lb: free staging set, receives step kt + 2; ua / ub: holds step kt + 1, split into `nxt` during this step.
    // Matrix-first waves: the rows of step kt + 2 are requested at the top of step kt and split at the END of step kt + 1.
    long long tk_last = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    if (mma_first) {
      if (kt + 2 < nk) {
        load_a(la, kt + 2);
        load_b(lb, kt + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      TNW_STAMP(tk_load)
      if (mm_live) mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      TNW_STAMP(tk_mma)
      if (kt + 1 < nk) {
        store_a(ua, kt + 1, nxt);
        store_b(ub, kt + 1, nxt);
      }
      __builtin_amdgcn_sched_barrier(0);
      TNW_STAMP(tk_store)
    } else {
      // staging-first waves: ua / ub (step kt + 1) is split right away and its registers take the rows of step kt + 3 --
      // la / lb keep step kt + 2, requested one step ago -- so every request has two whole steps to land here as well
      if (kt + 1 < nk) {
        store_a(ua, kt + 1, nxt);
        store_b(ub, kt + 1, nxt);
      }
      __builtin_amdgcn_sched_barrier(0);
      TNW_STAMP(tk_store)
      if (kt + 3 < nk) {
        load_a(ua, kt + 3);
        load_b(ub, kt + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      TNW_STAMP(tk_load)
      if (mm_live) mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      TNW_STAMP(tk_mma)
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    TNW_STAMP(tk_bar)
  };
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(kt, sa, sb, sa2, sb2, buf0, buf1);
    if (kt + 1 < nk) kstep(kt + 1, sa2, sb2, sa, sb, buf1, buf0);
  }

#undef TNW_STAMP
  if (prof && lane == 0 && (wave == 0 || wave == 4)) {   // 8 int64 per workgroup: waves 0 (matrix-first) and 4 (staging-first)
    long long* d = g.dbg + 8 * (size_t)blockIdx.x + (wave ? 4 : 0);
    d[0] = ((long long)__builtin_amdgcn_s_memtime() - tk_begin) | ((long long)nk << 40);
    d[1] = tk_mma; d[2] = tk_store | (tk_load << 32); d[3] = tk_bar;
  }
  if (g.flags & TNF_NO_EPILOGUE) return;
  const int tsel = wave >> 2;                                   // which tile of the pair this wave's quadrant belongs to
  const int vw = 2 * ((wave >> 1) & 1) + (wave & 1);            // its wave id in the 128 x 128 kernel's quadrant layout
  float* slot_t[2] = {g.ws ? g.ws + (size_t)slot0 * TN_WS_TILE : nullptr, g.ws ? g.ws + (size_t)slot1 * TN_WS_TILE : nullptr};
  if (bias_tile) {   // the loop's last barrier has passed: the image is free
    float* red = reinterpret_cast<float*>(smem_w);
    red[sga * 256 + sca] = bias_acc;
    __syncthreads();
    if (tid < 256 && (two || tid < BM)) {
      const float sum = red[tid] + red[256 + tid];
      float* sl = slot_t[tid >> 7];
      if (sl) sl[BM * BN + (tid & 127)] = sum;
      else if (i0 + tid < q.NA) atomicAdd(q.dbias + i0 + tid, sum);
    }
  }
  if (!mm_live) return;
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    float* slot = slot_t[tsel];
    if (slot) {
      float* w = slot + ((vw * 4 + s2) * 64 + lane) * 16;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {acc[s2][4 * qd], acc[s2][4 * qd + 1], acc[s2][4 * qd + 2], acc[s2][4 * qd + 3]};
        *reinterpret_cast<f32x4*>(w + 4 * qd) = v;
      }
    } else {
      const int col = j0 + 32 * tn_jsub(2, vw, s2) + (lane & 31);
      if (col >= q.NB) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 128 * tsel + 32 * tn_isub(2, vw, s2) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < q.NA) atomicAdd(q.C + (size_t)row * q.ldc + col, acc[s2][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side: the plan (tiles, layouts, cost-weighted row chunks) and the C ABI
// ---------------------------------------------------------------------------------------
static int g_tn_flags = -1;   // NUDF_TN_FLAGS / nudf_set_tn_flags
extern "C" int nudf_set_tn_flags(int f) {
  const int old = g_tn_flags < 0 ? 0 : g_tn_flags;
  g_tn_flags = f;
  return old;
}
static long long* g_tn_dbg = nullptr;
extern "C" int nudf_set_tn_debug(void* buf) {   // device buffer of >= 4 int64 per workgroup (or NULL): tuning only
  g_tn_dbg = (long long*)buf;
  return 0;
}
static int tn_flags() {
  if (g_tn_flags < 0) { const char* e = getenv("NUDF_TN_FLAGS"); g_tn_flags = e ? atoi(e) : 0; }
  return g_tn_flags;
}

// time of one k-step of a tile with n live sub-tiles per wave, in units where a full tile is 4: MFMA work is n, but
// every k-step also pays the fixed global-load -> LDS -> barrier latency (NUDF_TN_COSTS="c1,c2,c3,c4": tuning hook)
static double tn_cost(int n) {
  static double c[4] = {-1.0, 0, 0, 0};
  if (c[0] < 0) {
    double d[4] = {2.0, 3.0, 3.5, 4.0};   // measured (scripts/tn_group_bench.py, profiles/r02_tn_gemm.txt)
    const char* e = getenv("NUDF_TN_COSTS");
    if (e) sscanf(e, "%lf,%lf,%lf,%lf", &d[0], &d[1], &d[2], &d[3]);
    for (int i = 3; i >= 0; --i) c[i] = d[i];
  }
  return c[n - 1];
}

// returns the number of workgroups (0: nothing to do, < 0: invalid arguments, error text set)
static int tn_plan(const NudfGemmTNGroup& g, TnPlan& pl) {
  if (g.n_problems <= 0 || g.M <= 0) return 0;
  if (g.n_problems > NUDF_TN_MAX_PROBLEMS) {
    nudf_set_error("nudf_gemm_tn_grouped: too many problems", hipErrorInvalidValue);
    return -1;
  }
  const int flags = tn_flags();
  double cost[TN_MAX_TILES];
  int nt = 0;
  for (int i = 0; i < g.n_problems; ++i) {
    const NudfGemmTNProblem& q = g.prob[i];
    const int ma = (q.flags & NUDF_TN_A16) ? 8 : 4, mb = (q.flags & NUDF_TN_B16) ? 8 : 4;
    {
      const bool ap4 = (q.flags & NUDF_TN_A_P4) != 0, bp4 = (q.flags & NUDF_TN_B_P4) != 0;
      const bool a16 = (q.flags & NUDF_TN_A16) != 0, b16 = (q.flags & NUDF_TN_B16) != 0;
      if ((ap4 && !a16) || (bp4 && !b16) || ((ap4 || bp4) && (g.prec == 0 || g.prec == 3)) || (ap4 && b16 && !bp4) || (bp4 && a16 && !ap4) ||
          ((ap4 || bp4) && (flags & TNF_NO_PACK16)) || ((ap4 || bp4) && g.rows_per_block % 4)) {
        nudf_set_error("nudf_gemm_tn_grouped: a 4-point packed operand is a bf16 operand of the 16-bit MFMA mode "
                       "(NUDF_TN_x16 set, prec != 0, rows_per_block a multiple of 4); a problem's other operand is packed "
                       "as well or fp32", hipErrorInvalidValue);
        return -1;
      }
    }
    if (((q.flags & NUDF_TN_A_BLK) && (q.flags & NUDF_TN_A16)) || ((q.flags & NUDF_TN_B_BLK) && (q.flags & NUDF_TN_B16))) {
      nudf_set_error("nudf_gemm_tn_grouped: the blocked layout is defined for fp32 operands", hipErrorInvalidValue);
      return -1;
    }
    if ((q.lda1 % ma) || (q.ldb1 % mb) || q.NA <= 0 || q.NB <= 0 || q.lda1 < ma || q.ldb1 < mb ||
        (((uintptr_t)q.A1) & 15) || (((uintptr_t)q.B1) & 15)) {
      nudf_set_error("nudf_gemm_tn_grouped: leading dimensions must be multiples of 4, operands 16-byte aligned",
                     hipErrorInvalidValue);
      return -1;
    }
    pl.prob[i] = q;
    const int ti_n = (q.NA + BM - 1) / BM, tj_n = (q.NB + BN - 1) / BN;
    for (int ti = 0; ti < ti_n; ++ti)
      for (int tj = 0; tj < tj_n; ++tj) {
        if (nt >= TN_MAX_TILES) {
          nudf_set_error("nudf_gemm_tn_grouped: more than 64 output tiles of 128 x 128 in one group",
                         hipErrorInvalidValue);
          return -1;
        }
        int li = (q.NA - ti * BM + 31) / 32, lj = (q.NB - tj * BN + 31) / 32;   // live 32-wide sub-tiles per side
        if (li > 4) li = 4;
        if (lj > 4) lj = 4;
        TnTile& tl = pl.tile[nt];
        tl.prob = (short)i; tl.ti = (short)ti; tl.tj = (short)tj; tl.rot = 0;
        tl.layout = (short)(li <= lj ? 0 : 1);
        tl.n = (short)(li <= lj ? li : lj);
        if (li == 4 && lj == 4 && !(flags & TNF_NO_QUADRANTS)) tl.layout = 2;
        if (g.prec != 0 || (q.flags & (NUDF_TN_A16 | NUDF_TN_B16))) {   // 16-bit MFMAs / bf16 operands: load-bound k-steps,
          tl.layout = 2;                                                // one loop shape (whole quadrants; zero-padded
          tl.n = 4;                                                     // operand columns make the dead sub-tiles exact 0)
        }
        cost[nt++] = tn_cost(tl.n);
      }
  }
  // With a workspace the reduce kernel adds the partial tiles into C / dbias with plain (non-atomic) read-modify-writes,
  // one workgroup per output tile: two problems of a group that write overlapping C rows or dbias entries would race.
  if (g.workspace && !(flags & TNF_ATOMICS)) {
    for (int i = 0; i < g.n_problems; ++i)
      for (int j = i + 1; j < g.n_problems; ++j) {
        const NudfGemmTNProblem &a = g.prob[i], &b = g.prob[j];
        const float *a0 = a.C, *a1 = a.C + (size_t)(a.NA - 1) * a.ldc + a.NB;
        const float *b0 = b.C, *b1 = b.C + (size_t)(b.NA - 1) * b.ldc + b.NB;
        const bool c_overlap = a0 < b1 && b0 < a1;
        const bool d_overlap = a.dbias && b.dbias && a.dbias < b.dbias + b.NA && b.dbias < a.dbias + a.NA;
        if (c_overlap || d_overlap) {
          nudf_set_error("nudf_gemm_tn_grouped: with a workspace the problems of a group must write disjoint C / dbias "
                         "ranges (the fixed-order reduction is not atomic)", hipErrorInvalidValue);
          return -1;
        }
      }
  }
  pl.n_tiles = nt;
  pl.M = g.M;
  pl.prec = g.prec;
  pl.flags = flags;
  // bf16x3 groups of fp32 row-major operands: pair every tile with the tile one tile row below it (same problem, same tile
  // column) for the wide kernel -- one 8-wave workgroup per CU computes both.  Worth it when most tiles find a partner (the
  // UDF network's 256-wide layers; the colour net's 128-wide ones do not).
  pl.wide = 0;
  int n_units = nt;
  for (int t = 0; t < nt; ++t) pl.partner[t] = -1;
  {
    bool ok = g.prec == 3 && (flags & TNF_WIDE) && !(flags & TNF_NO_SPLIT_IMAGE);
    for (int i = 0; i < g.n_problems && ok; ++i)
      if (g.prob[i].flags & (NUDF_TN_A16 | NUDF_TN_B16 | NUDF_TN_A_BLK | NUDF_TN_B_BLK | NUDF_TN_A_P4 | NUDF_TN_B_P4)) ok = false;
    if (ok) {
      int pairs = 0;
      for (int t = 0; t < nt; ++t) {
        if (pl.partner[t] != -1 || (pl.tile[t].ti & 1)) continue;
        const NudfGemmTNProblem& q = g.prob[pl.tile[t].prob];
        const int tj_n = (q.NB + BN - 1) / BN;
        const int u = t + tj_n;                        // tiles of a problem are enumerated tile row by tile row
        if (u < nt && pl.tile[u].prob == pl.tile[t].prob && pl.tile[u].ti == pl.tile[t].ti + 1 && pl.tile[u].tj == pl.tile[t].tj) {
          pl.partner[t] = (short)u;
          pl.partner[u] = -2;
          ++pairs;
        }
      }
      if (4 * pairs >= nt) {                           // at least half of the tiles are in pairs
        pl.wide = 1;
        n_units = nt - pairs;
      } else {
        for (int t = 0; t < nt; ++t) pl.partner[t] = -1;
      }
    }
  }
  const int nkt = (g.M + BK - 1) / BK;                 // k-steps over all points
  int chunks_of[TN_MAX_TILES];
  int blocks = 0;
  // workgroup order for a given chunking: workspace slots (tile-major), tile groups, the XCD-aware grid (tn_decode_block);
  // returns the largest number of LIVE workgroups any XCD receives
  auto layout = [&]() {
    blocks = 0;
    for (int t = 0; t < nt; ++t) {
      pl.tile[t].blk_start = blocks;
      blocks += (g.M + pl.tile[t].rows_per_block - 1) / pl.tile[t].rows_per_block;
    }
    pl.tile[nt].blk_start = blocks;
    pl.tile[nt].gfirst = (short)nt; pl.tile[nt].gn = 1; pl.tile[nt].rot = 0;
    for (int t = 0; t < nt;) {   // tile groups: same problem, same chunking (TNF_NO_XCD_MAP: every tile alone = tile-major order)
      int e = t + 1;
      while (!(flags & TNF_NO_XCD_MAP) && e < nt && pl.tile[e].prob == pl.tile[t].prob &&
             pl.tile[e].rows_per_block == pl.tile[t].rows_per_block) ++e;
      for (int k = t; k < e; ++k) { pl.tile[k].gfirst = (short)t; pl.tile[k].gn = (short)(e - t); }
      t = e;
    }
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int G = 0;
    for (int t = 0; t < nt; t += pl.tile[t].gn) {
      const int T = pl.tile[t].gn, C = pl.tile[t + 1].blk_start - pl.tile[t].blk_start;
      int rot = 0, size = C;
      if (T == 1) {
        for (int i = 0; i < C; ++i) ++load[(G + i) & 7];
      } else {
        const int w = C & 7;
        for (int x = 0; x < 8; ++x) load[x] += T * (C >> 3);
        if (w) {   // the rotation whose occupied lanes are the least loaded XCDs so far
          int best = 1 << 30;
          for (int r = 0; r < 8; ++r) {
            int m = 0, sum = 0;
            for (int cl = 0; cl < w; ++cl) { const int l = load[(G + ((cl + r) & 7)) & 7]; m = l > m ? l : m; sum += l; }
            if (m * 4096 + sum < best) { best = m * 4096 + sum; rot = r; }
          }
          for (int cl = 0; cl < w; ++cl) load[(G + ((cl + rot) & 7)) & 7] += T;
        }
        size = ((C + 7) >> 3) * 8 * T;
      }
      for (int k = t; k < t + T; ++k) { pl.tile[k].grid_start = G; pl.tile[k].rot = (short)rot; }
      G += size;
    }
    pl.tile[nt].grid_start = G;
    pl.grid_blocks = G;
    int m = 0;
    for (int x = 0; x < 8; ++x) m = load[x] > m ? load[x] : m;
    return m;
  };
  if (g.rows_per_block > 0) {
    bool any_blk = false;
    for (int i = 0; i < g.n_problems; ++i) any_blk = any_blk || (g.prob[i].flags & (NUDF_TN_A_BLK | NUDF_TN_B_BLK));
    if (any_blk && (g.rows_per_block % 32)) {
      nudf_set_error("nudf_gemm_tn_grouped: rows_per_block must be a multiple of 32 with blocked operands",
                     hipErrorInvalidValue);
      return -1;
    }
    for (int t = 0; t < nt; ++t) pl.tile[t].rows_per_block = g.rows_per_block;
    layout();
  } else {
    // exactly one resident wave of workgroups (2 per CU x 256 CUs; NUDF_TNG_BLOCKS: tuning hook), at least 8 k-steps
    // per workgroup.  Tile t costs cost[t] MFMA units per k-step: find the smallest per-workgroup budget T for which
    // sum_t ceil(cost[t] * nkt / T) fits, i.e. every workgroup does (nearly) the same number of MFMAs.
    static int target0 = -1;
    if (target0 < 0) { const char* e = getenv("NUDF_TNG_BLOCKS"); target0 = e ? atoi(e) : 512; }
    const int max_chunks = nkt / 8 > 0 ? nkt / 8 : 1;
    if (pl.wide) {
      // one resident wave of WIDE workgroups: a pair (or an unpaired tile) per CU, every tile the same number of chunks
      static int target_w = -1;
      if (target_w < 0) { const char* e = getenv("NUDF_TNW_BLOCKS"); target_w = e ? atoi(e) : 256; }
      int n = target_w / n_units;
      if (n < 1) n = 1;
      if (n > max_chunks) n = max_chunks;
      for (int t = 0; t < nt; ++t) chunks_of[t] = n;
      for (int t = 0; t < nt; ++t) pl.tile[t].rows_per_block = ((nkt + chunks_of[t] - 1) / chunks_of[t]) * BK;
      layout();
    } else {
      // (the XCD-aware order keeps whole chunks of a tile group on one XCD, so the XCDs' shares are not exactly equal: when one
      // XCD would receive more live workgroups than it has slots -- a second round on its CUs -- the total is lowered)
      for (int target = target0;; target -= 8) {
        auto count = [&](double T, bool store) {
          long total = 0;
          for (int t = 0; t < nt; ++t) {
            // 16-bit operands: the k-step is bound by the loads / LDS traffic of the (always full-size) operand tiles, not by
            // the live MFMAs -- every tile costs the same
            const double c = ((flags & TNF_UNIFORM_CHUNKS) || g.prec != 0) ? 4.0 : cost[t];
            long n = (long)((c * (double)nkt + T - 1e-9) / T);
            if (n < 1) n = 1;
            if (n > max_chunks) n = max_chunks;
            if (store) chunks_of[t] = (int)n;
            total += n;
          }
          return total;
        };
        double lo = 0.0, hi = 4.0 * nkt;                   // hi: one chunk per tile (always fits: nt <= 64 <= target)
        if (count(hi, false) <= target) {
          for (int it = 0; it < 60; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (count(mid, false) <= target) hi = mid; else lo = mid;
          }
        }
        count(hi, true);
        for (int t = 0; t < nt; ++t) pl.tile[t].rows_per_block = ((nkt + chunks_of[t] - 1) / chunks_of[t]) * BK;
        const int most = layout();
        if (most <= target0 / 8 || target - 8 < nt || target - 8 < target0 / 2) break;
      }
    }
  }
  pl.n_units = 0;
  pl.unit_chunks = nt > 0 ? pl.tile[1].blk_start - pl.tile[0].blk_start : 1;
  if (pl.wide) {
    for (int t = 0; t < nt; ++t) {
      if (pl.tile[t + 1].blk_start - pl.tile[t].blk_start != pl.unit_chunks) pl.wide = 0;   // (cannot happen: uniform chunks)
      if (pl.partner[t] != -2) pl.unit_tile[pl.n_units++] = (short)t;
    }
  }
  return blocks;
}

extern "C" int64_t nudf_gemm_tn_grouped_workspace(const NudfGemmTNGroup* args) {
  TnPlan pl;
  const int blocks = tn_plan(*args, pl);
  pl.dbg = nullptr;
  return blocks <= 0 ? (int64_t)blocks : (int64_t)blocks * TN_WS_TILE;
}

// host mirror of the kernels' blockIdx -> (tile, chunk) decode, for tests of the plan: out[4 b ..] = {problem, tile row *
// 256 + tile column, row chunk, workspace slot} of workgroup b (-1 x 4: a hole of the XCD-aware order); returns the number of
// workgroups LAUNCHED (<= capacity written)
extern "C" int nudf_gemm_tn_grouped_plan(const NudfGemmTNGroup* args, int32_t* out, int capacity) {
  TnPlan pl;
  const int blocks = tn_plan(*args, pl);
  if (blocks <= 0) return blocks;
  for (int b = 0; b < pl.grid_blocks && b < capacity; ++b) {
    int t, chunk;
    tn_decode_block(pl, b, t, chunk);
    if (t < 0) { out[4 * b] = out[4 * b + 1] = out[4 * b + 2] = out[4 * b + 3] = -1; continue; }   // a hole: exits at once
    out[4 * b] = pl.tile[t].prob;
    out[4 * b + 1] = pl.tile[t].ti * 256 + pl.tile[t].tj;
    out[4 * b + 2] = chunk;
    out[4 * b + 3] = pl.tile[t].blk_start + chunk;
  }
  return pl.grid_blocks;
}

extern "C" int nudf_gemm_tn_grouped(const NudfGemmTNGroup* args, void* stream) {
  TnPlan pl;
  const int blocks = tn_plan(*args, pl);
  if (blocks == 0) return 0;
  if (blocks < 0) return (int)hipErrorInvalidValue;
  pl.ws = (pl.flags & TNF_ATOMICS) ? nullptr : args->workspace;
  pl.dbg = g_tn_dbg;
  pl.assign = args->assign ? 1 : 0;
  if (pl.assign && (!pl.ws || (pl.flags & TNF_NO_EPILOGUE))) {
    nudf_set_error("nudf_gemm_tn_grouped: assign needs the workspace (two-pass) path", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  if (pl.ws && ((((uintptr_t)pl.ws) & 15) || args->workspace_floats < (int64_t)blocks * TN_WS_TILE)) {
    nudf_set_error("nudf_gemm_tn_grouped: workspace too small or not 16-byte aligned "
                   "(nudf_gemm_tn_grouped_workspace gives the size)", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  // 16-bit MFMA mode: the packed-image kernel, unless an operand is in the blocked fp32 layout (generic kernel only) or
  // most operands are stored as fp32 (its staging of an fp32 operand is heavier: 262 -> 278 us with all-fp32 operands,
  // 228 -> 183 us with all-bf16 ones at 65 536 points)
  bool packed16 = pl.prec != 0 && pl.prec != 3 && !(pl.flags & TNF_NO_PACK16);   // (bf16x3 splits from the fp32 image)
  int n16 = 0;
  for (int i = 0; i < args->n_problems && packed16; ++i) {
    const int f = args->prob[i].flags;
    if (f & (NUDF_TN_A_BLK | NUDF_TN_B_BLK)) packed16 = false;
    n16 += ((f & NUDF_TN_A16) ? 1 : 0) + ((f & NUDF_TN_B16) ? 1 : 0);
  }
  if (n16 < args->n_problems) packed16 = false;
  bool any_p4 = false;
  for (int i = 0; i < args->n_problems; ++i) any_p4 = any_p4 || (args->prob[i].flags & (NUDF_TN_A_P4 | NUDF_TN_B_P4));
  if (any_p4) {   // only the packed-image kernel reads 4-point packed operands
    for (int i = 0; i < args->n_problems; ++i)
      if (args->prob[i].flags & (NUDF_TN_A_BLK | NUDF_TN_B_BLK)) {
        nudf_set_error("nudf_gemm_tn_grouped: 4-point packed and blocked operands cannot share a group", hipErrorInvalidValue);
        return (int)hipErrorInvalidValue;
      }
    packed16 = true;
  }
  // bf16x3 mode: the split-image kernel when every operand is fp32 row-major (what the bf16x3 chains store); anything else
  // goes through the generic kernel, which splits on the way OUT of its fp32 image (NUDF_TN_FLAGS & 512 forces that: A/B)
  // f16x2 mode (prec 4): fp32 row-major operands only (what the split-mode chains store); anything else is an error -- the
  // caller decides per group (mlp.py) and falls back to prec 3 itself
  if (pl.prec == 4) {
    for (int i = 0; i < args->n_problems; ++i)
      if (args->prob[i].flags & (NUDF_TN_A16 | NUDF_TN_B16 | NUDF_TN_A_BLK | NUDF_TN_B_BLK | NUDF_TN_A_P4 | NUDF_TN_B_P4)) {
        nudf_set_error("nudf_gemm_tn_grouped: prec 4 (f16x2) takes fp32 row-major operands", hipErrorInvalidValue);
        return (int)hipErrorInvalidValue;
      }
    pl.amax_a = args->amax_a;
    pl.amax_b = args->amax_b;
    hipLaunchKernelGGL(gemm_tn2_group_kernel, dim3(pl.grid_blocks), dim3(256), 0, (hipStream_t)stream, pl);
    NUDF_CHECK_LAUNCH("nudf_gemm_tn_grouped");
    if (pl.ws && !(pl.flags & TNF_NO_EPILOGUE)) {
      hipLaunchKernelGGL(tn_reduce_kernel, dim3(pl.n_tiles * 17), dim3(256), 0, (hipStream_t)stream, pl);
      NUDF_CHECK_LAUNCH("nudf_gemm_tn_grouped (reduce)");
    }
    return 0;
  }
  bool split3 = pl.prec == 3 && !(pl.flags & TNF_NO_SPLIT_IMAGE);
  for (int i = 0; i < args->n_problems && split3; ++i)
    if (args->prob[i].flags & (NUDF_TN_A16 | NUDF_TN_B16 | NUDF_TN_A_BLK | NUDF_TN_B_BLK | NUDF_TN_A_P4 | NUDF_TN_B_P4)) split3 = false;
  if (split3 && pl.wide)
    hipLaunchKernelGGL(gemm_tn3w_group_kernel, dim3(pl.n_units * pl.unit_chunks), dim3(512), 0, (hipStream_t)stream, pl);
  else if (split3) hipLaunchKernelGGL(gemm_tn3_group_kernel, dim3(pl.grid_blocks), dim3(256), 0, (hipStream_t)stream, pl);
  else if (packed16) hipLaunchKernelGGL(gemm_tn16_group_kernel, dim3(pl.grid_blocks), dim3(256), 0, (hipStream_t)stream, pl);
  else hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(pl.grid_blocks), dim3(256), 0, (hipStream_t)stream, pl);
  NUDF_CHECK_LAUNCH("nudf_gemm_tn_grouped");
  if (pl.ws && !(pl.flags & TNF_NO_EPILOGUE)) {
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(pl.n_tiles * 17), dim3(256), 0, (hipStream_t)stream, pl);
    NUDF_CHECK_LAUNCH("nudf_gemm_tn_grouped (reduce)");
  }
  return 0;
}

// the two-pair form (dW = dY^T X + DA^T R of one layer): two problems of a group accumulating into the same C
extern "C" int nudf_gemm_tn(const NudfGemmTN* args, void* stream) {
  const NudfGemmTN& p = *args;
  if (p.M <= 0 || p.NA <= 0 || p.NB <= 0) return 0;
  NudfGemmTNGroup g;
  memset(&g, 0, sizeof(g));
  g.M = p.M; g.rows_per_block = p.rows_per_block; g.prec = p.prec;
  g.n_problems = p.A2 ? 2 : 1;
  NudfGemmTNProblem& q0 = g.prob[0];
  q0.A1 = p.A1; q0.B1 = p.B1; q0.C = p.C; q0.dbias = p.dbias;
  q0.lda1 = p.lda1; q0.ldb1 = p.ldb1; q0.ldc = p.ldc; q0.NA = p.NA; q0.NB = p.NB;
  if (p.A2) {
    NudfGemmTNProblem& q1 = g.prob[1];
    q1 = q0;
    q1.A1 = p.A2; q1.B1 = p.B2; q1.lda1 = p.lda2; q1.ldb1 = p.ldb2; q1.dbias = nullptr;
  }
  return nudf_gemm_tn_grouped(&g, stream);
}
