// fp32 MFMA GEMMs for the MLP chains of the NeuralUDF hot path (gfx950).
//
//   gemm_nn : C[M,N]   = epilogue( A[M,K] * B[K,N] )         A,B row-major, K % 32 == 0
//   (the weight-gradient contraction gemm_tn lives in gemm_tn_f32_mfma.hip)
//
// v_mfma_f32_32x32x2_f32 (exact fp32, == an fmaf chain), block tiles up to 128x128x32,
// 4 waves (2x2), each wave up to a 64x64 sub-tile = 2x2 MFMA tiles, double-buffered LDS.
// These replace the chains of F.linear / weight_norm / Softplus / ReLU / autograd ops of
// models/fields.py:192-231 (UDFNetwork), :452-495 (ResidualRenderingNetwork), :599-628 (NeRF)
// and their (double-)backward.  MFMA peak for this instruction: 157.3 TFLOP/s.
#include "nudf_common.h"
#include "../../include/nudf.h"
#include <stdlib.h>
#include <type_traits>

#define BM 128
#define BN 128
#define BK 32
#define LDA_S (BM + 1)  // As[k][m]: +1 pad makes the transposing ds_write_b32 conflict-free
#define LDB_S (BN + 4)  // Bs[k][n]: rows stay 16-byte aligned for ds_write_b128
#define A_TILE (BK * LDA_S)
#define B_TILE (BK * LDB_S)
#define LDT_S (BM + 4)  // TN kernel: both operands are straight copies
#define T_TILE (BK * LDT_S)

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int xcd_swizzle(int b, int nblk) {
  // blocks are dealt round-robin to the 8 XCDs; make consecutive logical tiles (which share
  // an A row-tile) land on the same XCD's L2.  Speed only.
  return ((nblk & 7) == 0) ? (b & 7) * (nblk >> 3) + (b >> 3) : b;
}

// softplus'(a) = s and 1 - s recovered from the STORED activation h = softplus100(a) * (1/xscale):
//   1 - s = exp(-100 h), s = -expm1(-100 h)  (exact identities; no 1 - s cancellation near s = 1, which
//   matters for softplus'' = 100 s (1 - s) in the second-order backward).  Above torch's threshold
//   (100 a > 20 <=> h > 0.2) autograd uses s = 1, s' = 0.
__device__ __forceinline__ void sp_derivs_from_h(float hstored, float xscale, float& s, float& om) {
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

template <int EPI>
__device__ __forceinline__ void epilogue_store(const NudfGemmNN& p, int row, int col, float acc) {
  float v = acc;
  if (p.bias) v += p.bias[col];
  const size_t r = (size_t)row;
  if (EPI == NUDF_EPI_NONE) {
    p.C1[r * p.ldc1 + col] = v * p.scale;
  } else if (EPI == NUDF_EPI_SOFTPLUS) {
    // nn.Softplus(beta=100, threshold=20) and its derivative from ONE hardware exp (v_exp_f32):
    //   z = e^{100 v}; h = log(1+z)/100; h' = z/(1+z)   (above the threshold: h = v, h' = 1)
    const float t = 100.0f * v;
    float h = v;
    if (t <= 20.0f) {
      const float z = __expf(t);
      h = (z < 1e-4f) ? (z - 0.5f * z * z) * 0.01f : __logf(1.0f + z) * 0.01f;   // log1p series for tiny z
    }
    p.C1[r * p.ldc1 + col] = h * p.scale;
    if (p.C2) p.C2[r * p.ldc2 + col] = (t <= 20.0f) ? __fdividef(__expf(t), 1.0f + __expf(t)) : 1.0f;
  } else if (EPI == NUDF_EPI_RELU) {
    p.C1[r * p.ldc1 + col] = fmaxf(v, 0.0f) * p.scale;
  } else if (EPI == NUDF_EPI_MUL) {
    p.C1[r * p.ldc1 + col] = v * p.X1[r * p.ldx1 + col] * p.scale;
  } else if (EPI == NUDF_EPI_MULMASK) {
    p.C1[r * p.ldc1 + col] = (p.X1[r * p.ldx1 + col] > 0.0f) ? v * p.scale : 0.0f;
  } else if (EPI == NUDF_EPI_MULSP) {
    float sg, om;
    sp_derivs_from_h(p.X1[r * p.ldx1 + col], p.xscale, sg, om);
    p.C1[r * p.ldc1 + col] = v * sg * p.scale;
  } else if (EPI == NUDF_EPI_TANGENT) {
    float sg, om;
    sp_derivs_from_h(p.X1[r * p.ldx1 + col], p.xscale, sg, om);
    p.C1[r * p.ldc1 + col] = v * sg * p.scale;
    // X2 = da = delta * s, so  acc * delta * softplus''  =  acc * da * 100 (1 - s)
    p.C2[r * p.ldc2 + col] = v * p.X2[r * p.ldx2 + col] * 100.0f * om;
  } else if (EPI == NUDF_EPI_BWD) {
    float sg, om;
    sp_derivs_from_h(p.X1[r * p.ldx1 + col], p.xscale, sg, om);
    p.C1[r * p.ldc1 + col] = v * p.scale * sg + (p.X2 ? p.X2[r * p.ldx2 + col] : 0.0f);
  } else if (EPI == NUDF_EPI_SIGMOID) {
    // first iparam columns through a sigmoid (optionally mirrored into C2), the rest raw
    if (col < p.iparam) {
      float s = sigmoidf_(v);
      p.C1[r * p.ldc1 + col] = s;
      if (p.C2) p.C2[r * p.ldc2 + col] = s;
    } else if (p.C3) {
      p.C3[r * p.ldc3 + (col - p.iparam)] = v;
    } else {
      p.C1[r * p.ldc1 + col] = v;
    }
  } else if (EPI == NUDF_EPI_UDFHEAD) {
    // channel 0 = |x| * scale (the 'abs' UDF head, fields.py:184-190, 210) -> C2[row], its sign ->
    // C3[row] (kept for the backward); channels 1.. (the appearance feature) -> C1[row, col-1]
    if (col == 0) {
      // udf_out and its derivative by p.iparam: 0 'abs', 1 'square', 2 'sdf' (fields.py:184-190)
      const float hv = (p.iparam == 1) ? v * v : ((p.iparam == 2) ? v : fabsf(v));
      const float hm = (p.iparam == 1) ? 2.0f * v : ((p.iparam == 2) ? 1.0f : ((v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f)));
      if (p.C2) p.C2[r] = hv * p.scale;
      if (p.C3) p.C3[r] = hm;
    } else if (p.C1) {
      p.C1[r * p.ldc1 + (col - 1)] = v;
    }
  } else if (EPI == NUDF_EPI_SKIPSPLIT) {
    // reverse sweep through the skip concat: columns < iparam belong to the hidden branch
    // (times softplus' and scale), the rest go to the embedding branch (times scale)
    if (col < p.iparam) {
      float sg, om;
      sp_derivs_from_h(p.X1[r * p.ldx1 + col], p.xscale, sg, om);
      p.C1[r * p.ldc1 + col] = v * sg * p.scale;
    } else {
      p.C2[r * p.ldc2 + (col - p.iparam)] = v * p.scale;
    }
  } else if (EPI == NUDF_EPI_RELU_DUAL) {
    // relu output to two destinations (hidden tap of the colour net, fields.py:472-473)
    float h = fmaxf(v, 0.0f);
    p.C1[r * p.ldc1 + col] = h;
    if (p.C2) p.C2[r * p.ldc2 + col] = h;
  } else if (EPI == NUDF_EPI_ADDMASK) {
    p.C1[r * p.ldc1 + col] = (p.X1[r * p.ldx1 + col] > 0.0f) ? (v + p.X2[r * p.ldx2 + col]) * p.scale : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------
// C[M,N] = epilogue(A[M,K] B[K,N]).  Block = 4 waves (2x2); each wave owns (WM*32)x(WN*32) outputs as
// WMxWN MFMA tiles, so the block tile is (64*WM)x(64*WN).  NBUF = 1: single LDS buffer + register
// prefetch, two barriers per k-step, small enough (33 KB at 128x128) for 3-4 co-resident blocks per CU
// whose epilogues (HBM-bound) overlap the other blocks' MFMA phases; NBUF = 2: double-buffered LDS.
// ---------------------------------------------------------------------------------------
template <int EPI, int WM, int WN, int NBUF, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_nn_kernel(NudfGemmNN p) {
  constexpr int TBM = 64 * WM, TBN = 64 * WN;
  constexpr int LDA = TBM + 1, LDB = TBN + 4;
  constexpr int ATILE = BK * LDA, BTILE = BK * LDB;
  constexpr int APASS = TBM / 32, BPASS = TBN / 32;  // float4 loads per thread per k-step
  __shared__ __attribute__((aligned(16))) float smem[NBUF * (ATILE + BTILE)];
  float* As = smem;
  float* Bs = smem + NBUF * ATILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + TBN - 1) / TBN;
  const int lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (lb / tiles_n) * TBM;
  const int n0 = (lb % tiles_n) * TBN;
  const int nk = p.K / BK;

  int a_off[APASS], b_off[BPASS];
  int a_lds[APASS], b_lds[BPASS];
  bool b_ok[BPASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int idx = ps * 256 + tid;
    const int row = idx >> 3, kq = idx & 7;
    int gr = m0 + row;
    if (gr > p.M - 1) gr = p.M - 1;
    a_off[ps] = gr * p.lda + kq * 4;
    a_lds[ps] = (kq * 4) * LDA + row;
  }
#pragma unroll
  for (int ps = 0; ps < BPASS; ++ps) {
    const int idx = ps * 256 + tid;
    const int kr = idx / (TBN / 4), c4 = idx % (TBN / 4);
    const int gc = n0 + c4 * 4;
    b_ok[ps] = gc < p.ldb;
    b_off[ps] = kr * p.ldb + (b_ok[ps] ? gc : 0);
    b_lds[ps] = kr * LDB + c4 * 4;
  }

  f32x4 ra[APASS], rb[BPASS];
  auto gload = [&](int kt) {
    const float* Ak = p.A + kt * BK;
    const float* Bk = p.B + (size_t)kt * BK * p.ldb;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) ra[ps] = *reinterpret_cast<const f32x4*>(Ak + a_off[ps]);
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps)
      rb[ps] = b_ok[ps] ? *reinterpret_cast<const f32x4*>(Bk + b_off[ps]) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto sstore = [&](int buf) {
    float* as = As + buf * ATILE;
    float* bs = Bs + buf * BTILE;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      float* d = as + a_lds[ps];
      d[0] = ra[ps].x;
      d[LDA] = ra[ps].y;
      d[2 * LDA] = ra[ps].z;
      d[3 * LDA] = ra[ps].w;
    }
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) *reinterpret_cast<f32x4*>(bs + b_lds[ps]) = rb[ps];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // wave-uniform activity of the column tiles (N may end inside the block tile)
  const int ncol_act = min(WN, max(0, (p.N - (n0 + wn * 32 * WN) + 31) / 32));

  auto compute = [&](int buf) {
    const float* as = As + buf * ATILE + (lane >> 5) * LDA + wm * (32 * WM) + (lane & 31);
    const float* bs = Bs + buf * BTILE + (lane >> 5) * LDB + wn * (32 * WN) + (lane & 31);
    if (ncol_act == WN) {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[WM], b[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = as[(2 * kk) * LDA + 32 * i];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = bs[(2 * kk) * LDB + 32 * j];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
      }
    } else if (ncol_act > 0) {  // only the first column tile is live (WN == 2)
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const float b0 = bs[(2 * kk) * LDB];
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[i][0] = mfma32(as[(2 * kk) * LDA + 32 * i], b0, acc[i][0]);
      }
    }
  };

  if (NBUF == 2) {
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) gload(kt + 1);
      compute(cur);
      if (kt + 1 < nk) sstore(cur ^ 1);
      __syncthreads();
    }
  } else {
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
      sstore(0);
      __syncthreads();
      if (kt + 1 < nk) gload(kt + 1);
      compute(0);
      __syncthreads();
    }
  }

  if (ncol_act == 0) return;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      if (j >= ncol_act) continue;
      const int col = n0 + wn * (32 * WN) + j * 32 + (lane & 31);
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (32 * WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M) epilogue_store<EPI>(p, row, col, acc[i][j][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
template <int EPI, int WM, int WN, int NBUF, int OCC>
static int launch_cfg(const NudfGemmNN& p, hipStream_t st) {
  const int tiles = ((p.M + 64 * WM - 1) / (64 * WM)) * ((p.N + 64 * WN - 1) / (64 * WN));
  hipLaunchKernelGGL((gemm_nn_kernel<EPI, WM, WN, NBUF, OCC>), dim3(tiles), dim3(256), 0, st, p);
  NUDF_CHECK_LAUNCH("nudf_gemm_nn");
  return 0;
}

static int g_variant = -1;  // NUDF_GEMM_VARIANT: 0 auto, 1 = 128x128 double-buffered, 2 = 128x128 single,
                            // 3 = 64x128 single, 4 = 64x64 single   (tuning / A-B measurements only)
extern "C" int nudf_set_gemm_variant(int v) {
  const int old = g_variant;
  g_variant = v;
  return old;
}

template <int EPI>
static int launch_nn(const NudfGemmNN& p, hipStream_t st) {
  if (g_variant < 0) {
    const char* e = getenv("NUDF_GEMM_VARIANT");
    g_variant = e ? atoi(e) : 0;
  }
  int v = g_variant;
  if (v == 0) {
    // measured on MI355X (scripts/gemm_bench.py, profiles/r01_gemm_variants.txt): the 64x64 tile at 8 waves/SIMD
    // wins on every shape of this workload -- these layer GEMMs sit at the fp32 ridge (K = 128..256, several
    // [M,N] epilogue operands), so overlapping many small blocks' epilogues with other blocks' MFMA phases
    // matters more than LDS reuse; 64x128 is within 5 % on M = 32768
    v = 4;
  }
  switch (v) {
    case 1: return launch_cfg<EPI, 2, 2, 2, 2>(p, st);
    case 2: return launch_cfg<EPI, 2, 2, 1, 4>(p, st);
    case 3: return launch_cfg<EPI, 1, 2, 1, 4>(p, st);
    default: return launch_cfg<EPI, 1, 1, 1, 6>(p, st);
  }
}

extern "C" int nudf_gemm_nn(const NudfGemmNN* args, void* stream) {
  const NudfGemmNN& p = *args;
  hipStream_t st = (hipStream_t)stream;
  if (p.M <= 0 || p.N <= 0) return 0;
  if (p.K <= 0 || (p.K % BK) != 0 || (p.lda % 4) != 0 || (p.ldb % 4) != 0 ||
      (((uintptr_t)p.A) & 15) || (((uintptr_t)p.B) & 15)) {
    nudf_set_error("nudf_gemm_nn: K%32, lda%4, ldb%4 and 16-byte alignment required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  switch (p.epi) {
    case NUDF_EPI_NONE: return launch_nn<NUDF_EPI_NONE>(p, st);
    case NUDF_EPI_SOFTPLUS: return launch_nn<NUDF_EPI_SOFTPLUS>(p, st);
    case NUDF_EPI_RELU: return launch_nn<NUDF_EPI_RELU>(p, st);
    case NUDF_EPI_MUL: return launch_nn<NUDF_EPI_MUL>(p, st);
    case NUDF_EPI_MULMASK: return launch_nn<NUDF_EPI_MULMASK>(p, st);
    case NUDF_EPI_TANGENT: return launch_nn<NUDF_EPI_TANGENT>(p, st);
    case NUDF_EPI_BWD: return launch_nn<NUDF_EPI_BWD>(p, st);
    case NUDF_EPI_SIGMOID: return launch_nn<NUDF_EPI_SIGMOID>(p, st);
    case NUDF_EPI_UDFHEAD: return launch_nn<NUDF_EPI_UDFHEAD>(p, st);
    case NUDF_EPI_SKIPSPLIT: return launch_nn<NUDF_EPI_SKIPSPLIT>(p, st);
    case NUDF_EPI_RELU_DUAL: return launch_nn<NUDF_EPI_RELU_DUAL>(p, st);
    case NUDF_EPI_ADDMASK: return launch_nn<NUDF_EPI_ADDMASK>(p, st);
    case NUDF_EPI_MULSP: return launch_nn<NUDF_EPI_MULSP>(p, st);
    default:
      nudf_set_error("nudf_gemm_nn: unknown epilogue", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
  }
}
