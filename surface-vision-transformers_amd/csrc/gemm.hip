// sitk GEMM kernels for gfx950: the nn.Linear family of the SiT hot path.
//
//   gemm_nt   C[m][n] = sum_k A[m][k] W[n][k]      forward Linears and input gradients
//   wgrad     dW[n][k] += sum_m dY[m][n] X[m][k]   weight gradients (+ optional bias gradient)
//
// Both use 16x16 MFMA tiles with the WEIGHT/feature index on the accumulator's register axis
// (acc[jj] <-> feature 4*(lane>>4)+jj) and the token index on lane&15, so every lane owns 4
// consecutive output features: epilogues read bias/residual and write results as 8/16-byte vectors.
// Operand tiles are staged global -> registers -> LDS ([rows][128 B] swizzled image, common.h) with
// the next tile's loads issued before the current tile's MFMAs.
#include "common.h"

namespace sitk {

struct RowMap {
  int group, stride, offset;
};
SITK_DEV int map_row(const RowMap& r, int m) {
  return r.group ? (m / r.group) * r.stride + r.offset + (m % r.group) : m;
}
static RowMap to_rowmap(const sitk_rowmap& r) { return RowMap{r.group, r.stride, r.offset}; }

struct GemmParams {
  int M, N, K;
  const void* A;
  int lda;
  RowMap amap;
  const void* W;
  int ldw;
  void* out;
  int ldo;
  RowMap omap;
  void* out2;
  const float* bias;
  const void* aux;
  int ldaux;
  RowMap auxmap;
};

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), sharing exp(-x^2/2) between GELU and GELU'.
struct GeluParts {
  float cdf;  // 0.5 (1 + erf(x / sqrt2))
  float pdf;  // exp(-x^2/2) / sqrt(2 pi)
};
SITK_DEV GeluParts gelu_parts(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  const float e = __expf(-ax * ax);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);
  const float erfv = copysignf(erf_abs, x);
  return GeluParts{0.5f * (1.0f + erfv), e * 0.39894228040143267794f};
}

template <typename T, typename TO, int EPI>
SITK_DEV void gemm_epilogue(const GemmParams& p, int m, int n, f32x4 v) {
  if (p.bias) v += load4(p.bias + n);
  const size_t orow = (size_t)map_row(p.omap, m) * p.ldo + n;
  if constexpr (EPI == SITK_EPI_STORE) {
    store4(reinterpret_cast<TO*>(p.out) + orow, v);
  } else if constexpr (EPI == SITK_EPI_BIAS_RES) {
    const f32x4 r = load4(reinterpret_cast<const float*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux + n);
    store4(reinterpret_cast<float*>(p.out) + orow, v + r);
  } else if constexpr (EPI == SITK_EPI_BIAS_GELU) {
    store4(reinterpret_cast<T*>(p.out) + orow, v);
    f32x4 g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = v[i] * gelu_parts(v[i]).cdf;
    store4(reinterpret_cast<T*>(p.out2) + orow, g);
  } else if constexpr (EPI == SITK_EPI_DGELU) {
    const f32x4 u = load4(reinterpret_cast<const T*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const GeluParts gp = gelu_parts(u[i]);
      v[i] *= fmaf(u[i], gp.pdf, gp.cdf);
    }
    store4(reinterpret_cast<T*>(p.out) + orow, v);
  }
}

// tail chunk of a row whose length is not a multiple of the 16-byte vector: element-wise, zero filled
template <typename T, typename TS>
SITK_DEV u32x4 vec_load_partial(const TS* p, int nvalid) {
  constexpr int EPV = 16 / (int)sizeof(T);
  T tmp[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) tmp[e] = from_f32<T>(e < nvalid ? to_f32(p[e]) : 0.f);
  u32x4 r;
  __builtin_memcpy(&r, tmp, 16);
  return r;
}

// ------------------------------------------------------------------------------------------
// NT GEMM.  256 threads = 4 waves arranged WM (token) x WN (feature).
// ------------------------------------------------------------------------------------------
template <typename T, typename TA, typename TO, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int EPV = Mma<T>::EPV;
  constexpr int BKE = 128 / (int)sizeof(T);  // contraction elements per 128-byte tile row
  constexpr int ACH = BM * 8 / 256, WCH = BN * 8 / 256;
  __shared__ __attribute__((aligned(256))) char smem[(BM + BN) * 128];
  char* sA = smem;
  char* sW = smem + BM * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const TA* __restrict__ A = reinterpret_cast<const TA*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

  u32x4 ra[ACH], rw[WCH];
  const u32x4 zero = {0u, 0u, 0u, 0u};
  size_t a_base[ACH], w_base[WCH];
  bool a_ok[ACH], w_ok[WCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int c = tid + 256 * i, row = c >> 3, m = m0 + row;
    a_ok[i] = m < p.M;
    a_base[i] = a_ok[i] ? (size_t)map_row(p.amap, m) * p.lda + (c & 7) * EPV : 0;
  }
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int c = tid + 256 * i, row = c >> 3, n = n0 + row;
    w_ok[i] = n < p.N;
    w_base[i] = w_ok[i] ? (size_t)n * p.ldw + (c & 7) * EPV : 0;
  }
  auto gload = [&](int kt) {
    const int kb = kt * BKE;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int k = kb + ((tid + 256 * i) & 7) * EPV;
      ra[i] = zero;
      if (a_ok[i] && k < p.K)
        ra[i] = (k + EPV <= p.K) ? VecLoad<T, TA>::load(A + a_base[i] + kb) : vec_load_partial<T, TA>(A + a_base[i] + kb, p.K - k);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int k = kb + ((tid + 256 * i) & 7) * EPV;
      rw[i] = zero;
      if (w_ok[i] && k < p.K)
        rw[i] = (k + EPV <= p.K) ? VecLoad<T, T>::load(W + w_base[i] + kb) : vec_load_partial<T, T>(W + w_base[i] + kb, p.K - k);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<u32x4*>(sA + lds_off(c >> 3, (c & 7) * 16)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<u32x4*>(sW + lds_off(c >> 3, (c & 7) * 16)) = rw[i];
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = (p.K + BKE - 1) / BKE;
  gload(0);
  lstore();
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fw[NT], fa[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
        fw[i] = *reinterpret_cast<const u32x4*>(sW + lds_off(wn * (BN / WN) + 16 * i + fr, ks * 64 + fq * 16));
#pragma unroll
      for (int j = 0; j < MT; ++j)
        fa[j] = *reinterpret_cast<const u32x4*>(sA + lds_off(wm * (BM / WM) + 16 * j + fr, ks * 64 + fq * 16));
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = Mma<T>::mma(fw[i], fa[j], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < KT) {
      lstore();
      __syncthreads();
    }
  }

#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + wm * (BM / WM) + 16 * j + fr;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + wn * (BN / WN) + 16 * i + 4 * fq;
      if (m < p.M && n < p.N) gemm_epilogue<T, TO, EPI>(p, m, n, acc[i][j]);
    }
  }
}

template <typename T, typename TA, typename TO, int EPI>
static int launch_gemm_nt(const GemmParams& p, hipStream_t s) {
  const bool wide = (p.N % 128 == 0) || p.N > 1024;
  if (wide) {
    constexpr int BM = 128, BN = 128;
    const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_nt_kernel<T, TA, TO, EPI, BM, BN, 2, 2>), dim3(grid), dim3(256), 0, s, p);
  } else {
    constexpr int BM = 128, BN = 64;
    const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_nt_kernel<T, TA, TO, EPI, BM, BN, 4, 1>), dim3(grid), dim3(256), 0, s, p);
  }
  return check_launch("gemm_nt");
}

template <typename T>
static int dispatch_gemm_nt(const sitk_gemm_desc* d, hipStream_t s) {
  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.amap = to_rowmap(d->amap);
  p.W = d->W; p.ldw = d->ldw;
  p.out = d->out; p.ldo = d->ldo; p.omap = to_rowmap(d->omap);
  p.out2 = d->out2; p.bias = d->bias;
  p.aux = d->aux; p.ldaux = d->ldaux; p.auxmap = to_rowmap(d->auxmap);
  constexpr bool is_f32 = sizeof(T) == 4;
  const bool af32 = d->a_is_f32 && !is_f32;  // in f32 mode A is always "T"
  const bool of32 = d->out_is_f32 && !is_f32;
  switch (d->epilogue) {
    case SITK_EPI_STORE:
      if (af32 && of32) return launch_gemm_nt<T, float, float, SITK_EPI_STORE>(p, s);
      if (af32) return launch_gemm_nt<T, float, T, SITK_EPI_STORE>(p, s);
      if (of32) return launch_gemm_nt<T, T, float, SITK_EPI_STORE>(p, s);
      return launch_gemm_nt<T, T, T, SITK_EPI_STORE>(p, s);
    case SITK_EPI_BIAS_RES:
      SITK_REQUIRE(d->out_is_f32 || is_f32, "gemm_nt: BIAS_RES writes fp32");
      SITK_REQUIRE(d->aux != nullptr, "gemm_nt: BIAS_RES needs aux");
      if (af32) return launch_gemm_nt<T, float, float, SITK_EPI_BIAS_RES>(p, s);
      return launch_gemm_nt<T, T, float, SITK_EPI_BIAS_RES>(p, s);
    case SITK_EPI_BIAS_GELU:
      SITK_REQUIRE(!d->out_is_f32 || is_f32, "gemm_nt: BIAS_GELU writes the compute dtype");
      SITK_REQUIRE(d->out2 != nullptr, "gemm_nt: BIAS_GELU needs out2");
      SITK_REQUIRE(!af32, "gemm_nt: BIAS_GELU takes A in the compute dtype");
      return launch_gemm_nt<T, T, T, SITK_EPI_BIAS_GELU>(p, s);
    case SITK_EPI_DGELU:
      SITK_REQUIRE(!d->out_is_f32 || is_f32, "gemm_nt: DGELU writes the compute dtype");
      SITK_REQUIRE(d->aux != nullptr, "gemm_nt: DGELU needs aux");
      if (af32) return launch_gemm_nt<T, float, T, SITK_EPI_DGELU>(p, s);
      return launch_gemm_nt<T, T, T, SITK_EPI_DGELU>(p, s);
  }
  set_error("gemm_nt: unknown epilogue %d", d->epilogue);
  return SITK_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------
// Weight gradient.  Output tile 64 (n) x 64 (k) per workgroup, tokens split over gridDim chunks,
// fp32 atomics into dW.  Stage = 64 tokens of dY[:, n0:n0+64] and X[:, k0:k0+64] in LDS in their
// memory orientation ([token][feature]); MFMA operands need [feature][8 tokens] so fragments are
// read TRANSPOSED: ds_read_b64_tr_b16 for bf16, plain ds_read_b32 for f32.
//   A-operand = dY^T (rows i = n),  B-operand = X (cols j = k)  ->  acc[jj] <-> dW[n0+4g+jj][k0+(l&15)]
// ------------------------------------------------------------------------------------------
struct WgradParams {
  int M, N, K;
  const void* dY;
  int lddy;
  RowMap dymap;
  const void* X;
  int ldx;
  RowMap xmap;
  float* dW;
  int lddw;
  float* db;
  int chunk;  // tokens per workgroup (multiple of 64)
};

// transposed fragment: 8 "token" slots for feature column `col` (16-col block at byte cb of the
// panel row), tokens t0 + slot(j): bf16 j<4: 8g? -- see callers; returns the operand vector.
template <typename T>
struct TrFrag;

template <>
struct TrFrag<bf16> {
  // tile: [rows][128 B] image of bf16; block columns c0..c0+15 (c0 multiple of 16, < 64);
  // rows r0 + 8*(lane>>4) + {0..3} -> elements 0..3 and + {4..7} -> elements 4..7   (natural k order)
  static SITK_DEV u32x4 load_k8(const char* tile, int r0, int c0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = r0 + 8 * g + q;
    const int cb = (c0 + 4 * pp) * 2;
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) i16x4*)(tile + lds_off(row, cb)));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) i16x4*)(tile + lds_off(row + 4, cb)));
    u32x4 r;
    r[0] = __builtin_bit_cast(u32x2, lo)[0];
    r[1] = __builtin_bit_cast(u32x2, lo)[1];
    r[2] = __builtin_bit_cast(u32x2, hi)[0];
    r[3] = __builtin_bit_cast(u32x2, hi)[1];
    return r;
  }
  // rows r0 + 4*(lane>>4) + {0..3} -> elements 0..3 and r0 + 16 + 4*(lane>>4) + {0..3} -> 4..7:
  // the slot order of an accumulator pair reused as the other operand (attention kernels).
  static SITK_DEV u32x4 load_acc_order(const char* tile, int r0, int c0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = r0 + 4 * g + q;
    const int cb = (c0 + 4 * pp) * 2;
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) i16x4*)(tile + lds_off(row, cb)));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) i16x4*)(tile + lds_off(row + 16, cb)));
    u32x4 r;
    r[0] = __builtin_bit_cast(u32x2, lo)[0];
    r[1] = __builtin_bit_cast(u32x2, lo)[1];
    r[2] = __builtin_bit_cast(u32x2, hi)[0];
    r[3] = __builtin_bit_cast(u32x2, hi)[1];
    return r;
  }
};

template <>
struct TrFrag<float> {
  // f32 panel rows hold 32 floats (128 B); 16-col block c0 in {0,16}; step = 16 tokens:
  // element j <-> token r0 + 4*(lane>>4) + j
  static SITK_DEV u32x4 load_k8(const char* tile, int r0, int c0, int lane) {
    const int g = lane >> 4, col = c0 + (lane & 15);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(tile + lds_off(r0 + 4 * g + j, col * 4));
    return __builtin_bit_cast(u32x4, v);
  }
};

template <typename T, typename TDY>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  constexpr int EPV = Mma<T>::EPV;
  constexpr int PC = 128 / (int)sizeof(T);   // columns per 128-byte panel: 64 (bf16) / 32 (f32)
  constexpr int NP = 64 / PC;                // panels per 64-column operand tile: 1 / 2
  constexpr int TS = Mma<T>::KSTEP;          // tokens per mma step: 32 / 16
  constexpr int BT = 64;                     // tokens per stage
  constexpr int CH = BT * 64 * (int)sizeof(T) / 16 / 256;  // 16-byte chunks per thread per operand
  __shared__ __attribute__((aligned(256))) char smem[2 * NP * BT * 128 + 4 * 64 * 4];
  char* sY = smem;
  char* sX = smem + NP * BT * 128;
  float* sB = reinterpret_cast<float*>(smem + 2 * NP * BT * 128);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;  // 2 x 2 waves over (n, k); each 32 x 32
  const int tiles_k = (p.K + 63) / 64;
  const int tile = blockIdx.x, n0 = (tile / tiles_k) * 64, k0 = (tile % tiles_k) * 64;
  const int mbeg = blockIdx.y * p.chunk, mend = min(p.M, mbeg + p.chunk);
  const TDY* __restrict__ dY = reinterpret_cast<const TDY*>(p.dY);
  const T* __restrict__ X = reinterpret_cast<const T*>(p.X);
  const bool do_bias = p.db != nullptr && k0 == 0;

  u32x4 ry[CH], rx[CH];
  const u32x4 zero = {0u, 0u, 0u, 0u};
  constexpr int CPR = 64 / EPV;  // 16-byte chunks per 64-column row: 8 / 16
  auto gload = [&](int mt) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, cc = c % CPR, m = mt + row;
      const int n = n0 + cc * EPV, k = k0 + cc * EPV;
      const bool ok = m < mend;
      ry[i] = zero;
      if (ok && n < p.N) {
        const TDY* src = dY + (size_t)map_row(p.dymap, m) * p.lddy + n;
        ry[i] = (n + EPV <= p.N) ? VecLoad<T, TDY>::load(src) : vec_load_partial<T, TDY>(src, p.N - n);
      }
      rx[i] = (ok && k < p.K) ? VecLoad<T, T>::load(X + (size_t)map_row(p.xmap, m) * p.ldx + k) : zero;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, cc = c % CPR;
      const int panel = (cc * 16) / 128, pb = (cc * 16) % 128;
      *reinterpret_cast<u32x4*>(sY + panel * BT * 128 + lds_off(row, pb)) = ry[i];
      *reinterpret_cast<u32x4*>(sX + panel * BT * 128 + lds_off(row, pb)) = rx[i];
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  gload(mbeg);
  for (int mt = mbeg; mt < mend; mt += BT) {
    __syncthreads();  // previous stage fully consumed
    lstore();
    __syncthreads();
    if (mt + BT < mend) gload(mt + BT);
#pragma unroll
    for (int st = 0; st < BT / TS; ++st) {
      u32x4 fy[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cn = wn * 32 + 16 * i, ck = wk * 32 + 16 * i;
        fy[i] = TrFrag<T>::load_k8(sY + (cn / PC) * BT * 128, st * TS, cn % PC, lane);
        fx[i] = TrFrag<T>::load_k8(sX + (ck / PC) * BT * 128, st * TS, ck % PC, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma<T>::mma(fy[i], fx[j], acc[i][j]);
    }
    if (do_bias) {  // column sums of the dY stage: thread -> column tid&63, 16 tokens each
      const int col = tid & 63, r0 = (tid >> 6) * 16;
      const char* base = sY + (col / PC) * BT * 128;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        bsum += to_f32(*reinterpret_cast<const T*>(base + lds_off(r0 + r, (col % PC) * (int)sizeof(T))));
    }
  }

  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 32 + 16 * j + fr;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int n = n0 + wn * 32 + 16 * i + 4 * fq + jj;
        if (n < p.N && k < p.K) unsafeAtomicAdd(p.dW + (size_t)n * p.lddw + k, acc[i][j][jj]);
      }
    }
  if (do_bias) {
    __syncthreads();
    sB[tid] = bsum;
    __syncthreads();
    if (tid < 64 && n0 + tid < p.N)
      unsafeAtomicAdd(p.db + n0 + tid, sB[tid] + sB[tid + 64] + sB[tid + 128] + sB[tid + 192]);
  }
}

template <typename T>
static int dispatch_wgrad(const sitk_wgrad_desc* d, hipStream_t s) {
  WgradParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.dY = d->dY; p.lddy = d->lddy; p.dymap = to_rowmap(d->dymap);
  p.X = d->X; p.ldx = d->ldx; p.xmap = to_rowmap(d->xmap);
  p.dW = d->dW; p.lddw = d->lddw; p.db = d->db;
  const int tiles = cdiv(d->N, 64) * cdiv(d->K, 64);
  int splits = cdiv(1024, tiles);                      // ~4 workgroups per CU in flight
  splits = std::max(1, std::min(splits, cdiv(d->M, 256)));
  p.chunk = cdiv(cdiv(d->M, splits), 64) * 64;
  splits = cdiv(d->M, p.chunk);
  dim3 grid(tiles, splits);
  constexpr bool is_f32 = sizeof(T) == 4;
  if (d->dy_is_f32 && !is_f32)
    hipLaunchKernelGGL((wgrad_kernel<T, float>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, T>), grid, dim3(256), 0, s, p);
  return check_launch("gemm_wgrad");
}

}  // namespace sitk

extern "C" int sitk_gemm_nt(const sitk_gemm_desc* d, int dtype, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(d != nullptr, "gemm_nt: null descriptor");
  SITK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: empty problem %d %d %d", d->M, d->N, d->K);
  SITK_REQUIRE(d->N % 4 == 0 && d->K % 4 == 0, "gemm_nt: N %% 4 and K %% 4 required (N=%d K=%d)", d->N, d->K);
  {
    const int vec_a = (d->a_is_f32 || dtype == SITK_F32) ? 4 : 8, vec_w = dtype == SITK_F32 ? 4 : 8;
    SITK_REQUIRE(d->lda % vec_a == 0 && d->ldw % vec_w == 0 && d->ldo % 4 == 0,
                 "gemm_nt: leading dims must keep rows 16-byte aligned (lda=%d ldw=%d ldo=%d)", d->lda, d->ldw, d->ldo);
  }
  SITK_REQUIRE(d->A && d->W && d->out, "gemm_nt: null operand");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_BF16) return dispatch_gemm_nt<bf16>(d, s);
  if (dtype == SITK_F32) return dispatch_gemm_nt<float>(d, s);
  set_error("gemm_nt: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

extern "C" int sitk_gemm_wgrad(const sitk_wgrad_desc* d, int dtype, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(d != nullptr, "gemm_wgrad: null descriptor");
  SITK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_wgrad: empty problem");
  SITK_REQUIRE(d->N % 4 == 0 && d->K % 8 == 0, "gemm_wgrad: N %% 4 and K %% 8 required (N=%d K=%d)", d->N, d->K);
  SITK_REQUIRE(d->lddy % 4 == 0 && d->ldx % 8 == 0, "gemm_wgrad: leading dims must keep 16-byte alignment");
  SITK_REQUIRE(d->dY && d->X && d->dW, "gemm_wgrad: null operand");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_BF16) return dispatch_wgrad<bf16>(d, s);
  if (dtype == SITK_F32) return dispatch_wgrad<float>(d, s);
  set_error("gemm_wgrad: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}
