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
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sitk {

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
// Phi(x) = x >= 0 ? 1 - h : h with h = 0.5 * erfc(|x|/sqrt2) = 0.5 * poly(t) * exp(-x^2/2), t = 1/(1 + p|x|/sqrt2).
//   ACCURATE (f32 mode): A&S 7.1.26, 5 terms, |erf err| <= 1.5e-7
//   else     (bf16 mode): A&S 7.1.25, 3 terms, |erf err| <= 2.5e-5 (far below one bf16 ulp of the
//                         output; the VALU epilogue of the fc1 / dfc2 GEMMs is their bottleneck)
template <bool ACCURATE>
SITK_DEV GeluParts gelu_parts_t(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-x^2 / 2)
  float h;
  if constexpr (ACCURATE) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    h = poly * t * e;
  } else {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f * 0.70710678118654752440f, ax, 1.0f));
    float poly = fmaf(0.5f * 0.7478556f, t, 0.5f * -0.0958798f);
    poly = fmaf(poly, t, 0.5f * 0.3480242f);
    h = poly * t * e;
  }
  return GeluParts{x >= 0.f ? 1.0f - h : h, e * 0.39894228040143267794f};
}
template <typename T>
SITK_DEV GeluParts gelu_parts(float x) { return gelu_parts_t<sizeof(T) == 4>(x); }

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
    f32x4 g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const GeluParts gp = gelu_parts<T>(v[i]);
      g[i] = v[i] * gp.cdf;
      v[i] = fmaf(v[i], gp.pdf, gp.cdf) - 0.5f;              // gelu'(u) - 1/2 = Phi(u) - 1/2 + u phi(u): saved instead of u
    }
    store4(reinterpret_cast<T*>(p.out) + orow, v);
    store4(reinterpret_cast<T*>(p.out2) + orow, g);
  } else if constexpr (EPI == SITK_EPI_DGELU) {
    v *= load4(reinterpret_cast<const T*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux + n) + 0.5f;   // aux = the saved gelu'(u) - 1/2
    store4(reinterpret_cast<T*>(p.out) + orow, v);
  }
}

// Epilogue math only (no stores): v <- acc (+bias) (+residual | * aux); BIAS_GELU: v2 <- gelu(v), v <- gelu'(v).
template <typename T, int EPI, bool RES_LATER = false>
SITK_DEV void epilogue_math(const GemmParams& p, int m, int n, f32x4& v, f32x4& v2) {
  if (p.bias) v += load4(p.bias + n);
  if constexpr (EPI == SITK_EPI_BIAS_RES) {
    if constexpr (!RES_LATER) v += load4(reinterpret_cast<const float*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux + n);
  } else if constexpr (EPI == SITK_EPI_BIAS_GELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const GeluParts gp = gelu_parts<T>(v[i]);
      v2[i] = v[i] * gp.cdf;
      v[i] = fmaf(v[i], gp.pdf, gp.cdf) - 0.5f;              // gelu'(u) - 1/2, saved instead of u
    }
  } else if constexpr (EPI == SITK_EPI_DGELU) {
    v *= load4(reinterpret_cast<const T*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux + n) + 0.5f;   // aux = the saved gelu'(u) - 1/2
  }
}

// A 16-row x BN-column half tile held as v[i] = 4 features (16 i + 4*(lane>>4) ..) of row lane&15 goes
// through a wave-private LDS region and out to global memory as whole rows, 16 bytes per lane.
// RES (fp32 outputs of the BIAS_RES epilogue): the residual is added HERE, read as whole rows like the store, instead of in
// accumulator layout in front of the staging (16 rows x 64 B per wave instruction there).
template <typename TS, int BN, bool RES = false>
SITK_DEV void staged_rows_store(char* slot, const f32x4 (&v)[BN / 16], TS* out, const GemmParams& p, int mrow, int n0, int lane) {
  constexpr int PB = BN * (int)sizeof(TS) + 16;     // padded row pitch (bytes)
  constexpr int CPRW = BN * (int)sizeof(TS) / 16;   // 16-byte chunks per row
  constexpr int EPC = 16 / (int)sizeof(TS);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < BN / 16; ++i) store4(reinterpret_cast<TS*>(slot + fr * PB) + 16 * i + 4 * fq, v[i]);
#pragma unroll
  for (int c0 = 0; c0 < 16 * CPRW; c0 += 64) {
    const int c = c0 + lane, row = c / CPRW, cc = c % CPRW;
    u32x4 d = *reinterpret_cast<const u32x4*>(slot + row * PB + cc * 16);
    const int m = mrow + row, n = n0 + cc * EPC;
    if (m < p.M && n < p.N) {
      if constexpr (RES) {
        static_assert(!RES || sizeof(TS) == 4, "residual epilogue writes fp32");
        const f32x4 r = load4(reinterpret_cast<const float*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux + n);
        d = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, d) + r);
      }
      *reinterpret_cast<u32x4*>(out + (size_t)map_row(p.omap, m) * p.ldo + n) = d;
    }
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
  constexpr int STAGE = (BM + BN) * 128;    // one LDS stage: A tile then W tile
  __shared__ __attribute__((aligned(256))) char smem[2 * STAGE];
  char* sA = smem;
  char* sW = smem + BM * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const TA* __restrict__ A = reinterpret_cast<const TA*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

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
  // Pipeline: global loads run TWO k-tiles ahead in two register stages; LDS is double buffered, so
  // there is one barrier per k-tile and the write of tile k+1 overlaps other waves' MFMAs of tile k.
  auto gload = [&](int kt, u32x4 (&ra)[ACH], u32x4 (&rw)[WCH]) {
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
  auto lstore = [&](int buf, const u32x4 (&ra)[ACH], const u32x4 (&rw)[WCH]) {
    char* bA = sA + buf * STAGE;
    char* bW = sW + buf * STAGE;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<u32x4*>(bA + lds_off(c >> 3, (c & 7) * 16)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<u32x4*>(bW + lds_off(c >> 3, (c & 7) * 16)) = rw[i];
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const char* bA = sA + buf * STAGE;
    const char* bW = sW + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fw[NT], fa[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
        fw[i] = *reinterpret_cast<const u32x4*>(bW + lds_off(wn * (BN / WN) + 16 * i + fr, ks * 64 + fq * 16));
#pragma unroll
      for (int j = 0; j < MT; ++j)
        fa[j] = *reinterpret_cast<const u32x4*>(bA + lds_off(wm * (BM / WM) + 16 * j + fr, ks * 64 + fq * 16));
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = Mma<T>::mma(fw[i], fa[j], acc[i][j]);
    }
  };

  const int KT = (p.K + BKE - 1) / BKE;
  u32x4 ra0[ACH], rw0[WCH], ra1[ACH], rw1[WCH];
  gload(0, ra0, rw0);
  if (KT > 1) gload(1, ra1, rw1);
  lstore(0, ra0, rw0);
  __syncthreads();
  for (int kt = 0; kt < KT; kt += 2) {
    if (kt + 2 < KT) gload(kt + 2, ra0, rw0);
    compute(0);
    if (kt + 1 < KT) lstore(1, ra1, rw1);
    __syncthreads();
    if (kt + 1 >= KT) break;
    if (kt + 3 < KT) gload(kt + 3, ra1, rw1);
    compute(1);
    if (kt + 2 < KT) lstore(0, ra0, rw0);
    __syncthreads();
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

// ------------------------------------------------------------------------------------------
// Small-K NT GEMM (K <= 192 bf16: to_qkv, to_out.0, net.0 and their input gradients for dim 192):
// weight-resident, token-streaming, barrier-free main loop.
//   * a workgroup (8 waves, 2 per SIMD) owns one BN-column tile of W for the whole kernel: its
//     (BN x K) panel is brought into LDS once by LDS-DMA;
//   * every WAVE streams its own 32-token strips of A through a private LDS slot
//     (global_load_lds_dwordx4, 12 KiB per strip): wait own DMA -> 2 x (BN/16) x K/32 MFMAs ->
//     issue the next strip's DMA -> epilogue (bias / GELU / residual, 8-16 byte stores) while that
//     DMA is in flight.  No workgroup barrier after the prologue; the second wave of each SIMD fills
//     the MFMA, VALU and memory gaps of the first.
//   * strips of the same tokens for the different column tiles run on one XCD (xcd_remap), so A is
//     fetched from HBM once and re-served by that XCD's L2.
// ------------------------------------------------------------------------------------------
__device__ u32x4 g_zero_page_nt[4];

template <typename TO, int BN>
constexpr int wres_stage_bytes() { return (16 * (BN * (int)sizeof(TO) + 16) + 255) / 256 * 256; }
template <typename TO, int BN, int KT, int MT>
constexpr int wres_slot_bytes() {
  return KT * MT * 16 * 128 > wres_stage_bytes<TO, BN>() ? KT * MT * 16 * 128 : wres_stage_bytes<TO, BN>();
}

// WAVES waves per workgroup, each streaming strips of 16*MT tokens (16 waves x 16 rows when the LDS
// budget allows: 4 waves per SIMD hide the DMA / epilogue latencies of one another).
template <typename TO, int EPI, int BN, int KT, int WAVES, int MT>
__global__ __launch_bounds__(WAVES * 64) void gemm_nt_wres_kernel(GemmParams p, int groups) {
  using T = h16;
  constexpr int NT = BN / 16;                 // MFMA column tiles per strip
  constexpr int ROWS = 16 * MT;               // tokens per strip
  constexpr int PPP = ROWS / 8;               // 8-row DMA pieces per 128-byte panel of a strip
  constexpr int WB = KT * BN * 128;           // weight panel bytes
  constexpr int SB = wres_slot_bytes<TO, BN, KT, MT>();  // strip slot, also the epilogue's staging area
  __shared__ __attribute__((aligned(256))) char smem[WB + WAVES * SB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int g = L / tiles_n, n0 = (L % tiles_n) * BN;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_page_nt);
  const int r8 = lane >> 3;                                          // row within an 8-row DMA piece
  // ---- weight panel: KT panels x BN rows, pieces of 8 rows shared by the waves ----
  for (int q = wave; q < KT * BN / 8; q += WAVES) {
    const int kt = q / (BN / 8), row = (q % (BN / 8)) * 8 + r8;
    const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int k = kt * 64 + ((lane & 7) ^ (key << 1)) * 8;
    const int n = n0 + row;
    const T* src = (n < p.N && k < p.K) ? W + (size_t)n * p.ldw + k : reinterpret_cast<const T*>(zero);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + kt * BN * 128 + (q % (BN / 8)) * 1024), 16, 0, 0);
  }
  char* slot = smem + WB + wave * SB;
  const int nstrips = (p.M + ROWS - 1) / ROWS;
  const int unit = g * WAVES + wave, nunits = groups * WAVES;
  auto issue = [&](int strip) {
#pragma unroll
    for (int q = 0; q < KT * PPP; ++q) {
      const int kt = q / PPP, row = (q % PPP) * 8 + r8;
      const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      const int k = kt * 64 + ((lane & 7) ^ (key << 1)) * 8;
      const int m = strip * ROWS + row;
      const bool ok = m < p.M && k < p.K;
      const T* src = ok ? A + (size_t)map_row(p.amap, ok ? m : 0) * p.lda + k : reinterpret_cast<const T*>(zero);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + kt * ROWS * 128 + (q % PPP) * 1024), 16, 0, 0);
    }
  };
  int strip = unit;
  if (strip < nstrips) issue(strip);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                    // weight panel visible to all waves

  const int fr = lane & 15, fq = lane >> 4;
  for (; strip < nstrips; strip += nunits) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's strip has landed
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fa[MT], fw[NT];
#pragma unroll
        for (int j = 0; j < MT; ++j)
          fa[j] = *reinterpret_cast<const u32x4*>(slot + kt * ROWS * 128 + lds_off(16 * j + fr, ks * 64 + fq * 16));
#pragma unroll
        for (int i = 0; i < NT; ++i)
          fw[i] = *reinterpret_cast<const u32x4*>(smem + kt * BN * 128 + lds_off(16 * i + fr, ks * 64 + fq * 16));
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] = Mma<T>::mma(fw[i], fa[j], acc[i][j]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // slot fully read
    // ---- epilogue: math in registers, then the 16-row tiles go through this wave's own LDS slot so
    // that every store instruction writes whole rows (16 lanes x 16 B = 256 contiguous bytes)
    // instead of sixteen 32-byte fragments ----
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int mrow = strip * ROWS + 16 * j;
      f32x4 v1[NT], v2[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + 16 * i + 4 * fq, m = mrow + fr;
        v1[i] = acc[i][j];
        v2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M && n < p.N) epilogue_math<T, EPI, true>(p, m, n, v1[i], v2[i]);   // (residual: row layout, below)
      }
      staged_rows_store<TO, BN, EPI == SITK_EPI_BIAS_RES>(slot, v1, reinterpret_cast<TO*>(p.out), p, mrow, n0, lane);
      if constexpr (EPI == SITK_EPI_BIAS_GELU) staged_rows_store<T, BN>(slot, v2, reinterpret_cast<T*>(p.out2), p, mrow, n0, lane);
    }
    if (strip + nunits < nstrips) issue(strip + nunits);            // slot free again: refill it
  }
}

// ------------------------------------------------------------------------------------------
// Large-K NT GEMM for feature counts that are multiples of 192 (net.3, and the input gradients
// through net.0 / to_qkv; the patch embedding): tile 128 tokens x 192 features x 64 (k), 8 waves as
// 4 (tokens) x 2 (features), each 32 x 96.  A and W k-tiles travel global -> LDS by LDS-DMA into two
// stages (counted vmcnt + raw s_barrier, the next k-tile's DMA stays in flight under the MFMAs);
// fragments are read by ds_read_b128 inside asm blocks so that the compiler does not drain that DMA.
// 77 flop per staged byte (the 128 x 64 generic tile: 43), A is read once per 192 features.
// ------------------------------------------------------------------------------------------
#define SITK_N192_READS(KSBASE_A, KSBASE_W)                                                        \
  asm volatile(                                                                                    \
      "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\t"                                 \
      "ds_read_b128 %2, %9 offset:16384\n\tds_read_b128 %3, %9 offset:18432\n\t"                   \
      "ds_read_b128 %4, %9 offset:20480\n\tds_read_b128 %5, %9 offset:22528\n\t"                   \
      "ds_read_b128 %6, %9 offset:24576\n\tds_read_b128 %7, %9 offset:26624\n\t"                   \
      "s_waitcnt lgkmcnt(0)"                                                                       \
      : "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fw[0]), "=&v"(fw[1]), "=&v"(fw[2]), "=&v"(fw[3]), "=&v"(fw[4]), \
        "=&v"(fw[5])                                                                               \
      : "v"(KSBASE_A), "v"(KSBASE_W));                                                             \
  __builtin_amdgcn_sched_barrier(0)

// WM = waves along the token dimension (4: 128-token tiles, 3: 96-token tiles for grids that would
// otherwise leave a third of the CUs idle); 2 waves along the 192 features.
#ifdef SITK_N192_STAMPS
__device__ unsigned long long g_n192_stamps[16 * 8];      // diagnostic build: [wave][phase] cycle sums of one workgroup
#define SITK_NST(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); nst[i] += tn - ntp; ntp = tn; }
#else
#define SITK_NST(i)
#endif
template <typename TO, int EPI, int WM, int NSTG = 4, int MINW = 1>
__global__ __launch_bounds__(WM * 128, MINW) void gemm_nt_n192_kernel(GemmParams p) {
  using T = h16;
#ifdef SITK_N192_STAMPS
  unsigned long long nst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ntp = __builtin_amdgcn_s_memtime();
  const unsigned long long nt0 = ntp;
#endif
  constexpr int BM = WM * 32;
  constexpr int APC = BM / 8;              // 8-row DMA pieces of the A tile
  constexpr int PIECES = (APC + 24) / (2 * WM);  // per wave and stage: 5 (WM 4) / 6 (WM 3)
  static_assert(PIECES * 2 * WM == APC + 24, "pieces must divide evenly");
  constexpr int STG = (BM + 192) * 128;    // A rows then W rows 0..191
  // NSTG = 4: 160 KB, the whole LDS of a CU; NSTG = 2: 80 KB, two workgroups per CU (each other's fill and epilogue cover)
  __shared__ __attribute__((aligned(256))) char smem[NSTG * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  const int tiles_n = p.N / 192;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / tiles_n) * BM, n0 = (L % tiles_n) * 192;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_page_nt);

  // LDS-DMA roles: APC + 24 pieces of 8 rows per stage (A then W), PIECES per wave
  const int r8 = lane >> 3;
  const T* src_row[PIECES];
  int src_chunk[PIECES];
  bool src_ok[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int q = wave * PIECES + i, row = (q < APC ? q : q - APC) * 8 + r8;
    const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    src_chunk[i] = ((lane & 7) ^ (key << 1)) * 8;
    if (q < APC) {
      const int m = m0 + row;
      src_ok[i] = m < p.M;
      src_row[i] = A + (size_t)map_row(p.amap, src_ok[i] ? m : 0) * p.lda;
    } else {
      const int n = n0 + row;
      src_ok[i] = n < p.N;
      src_row[i] = W + (size_t)(src_ok[i] ? n : 0) * p.ldw;
    }
  }
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int k = kt * 64 + src_chunk[i];
      const T* src = (src_ok[i] && k < p.K) ? src_row[i] + k : reinterpret_cast<const T*>(zero);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + buf * STG + (wave * PIECES + i) * 1024), 16, 0, 0);
    }
  };
  // per-lane fragment addresses (LDS byte addresses): row*128 + ((ks*64 + fq*16) ^ key<<5); key from lane&15 only
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int keyl = ((fr >> 1) & 1) | (((fr >> 3) & 1) << 1);
  uint32_t aoff[2], woff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int byte = (ks * 64 + fq * 16) ^ (keyl << 5);
    aoff[ks] = lbase + (wm * 32 + fr) * 128 + byte;
    woff[ks] = lbase + (BM - 128) * 128 + (wn * 96 + fr) * 128 + byte;   // the read macro adds 16384 = 128 rows
  }

  f32x4 acc[6][2];
#pragma unroll
  for (int i = 0; i < 6; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int KT = (p.K + 63) / 64;
  // NSTG-deep ring: k-tiles kt+1 .. kt+NSTG-2 are in flight while kt is multiplied; ONE barrier per
  // k-tile (a wave passes barrier kt+1 only after its reads of stage kt, so refilling that stage
  // right after the barrier is safe).
#pragma unroll
  for (int i = 0; i < NSTG - 1; ++i)
    if (i < KT) issue(i, i);
  SITK_NST(0)                                                  // prologue (addresses + first DMA issue)
  for (int kt = 0; kt < KT; ++kt) {
    const int rem = min(NSTG - 2, KT - 1 - kt);
    if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SITK_NST(1)                                                // DMA wait
    __builtin_amdgcn_s_barrier();
    SITK_NST(2)                                                // barrier
    if (kt + NSTG - 1 < KT) issue(kt + NSTG - 1, (kt + NSTG - 1) % NSTG);
    SITK_NST(3)                                                // DMA issue
    const uint32_t bo = (kt % NSTG) * STG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[2], fw[6];
      const uint32_t ka = aoff[ks] + bo, kw = woff[ks] + bo;
      SITK_N192_READS(ka, kw);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma<T>::mma(fw[i], fa[j], acc[i][j]);
    }
    SITK_NST(4)                                                // fragment reads + MFMAs
  }
  // GELU' epilogue: the saved gelu'(u) of this wave's two 16 x 96 blocks is requested HERE, as whole 192-byte row segments (3 x 16
  // bytes per lane and block), and turned into accumulator layout through the wave's staging slot below.  Read in accumulator
  // layout (8 bytes per lane: 16 rows x 32 bytes per wave instruction, six of them per block) the same bytes cost 23 us of
  // the 112 us of config 3's d net.3 product (profiles/r03_gemm_experiments.txt).
  u32x4 urow[2][3];
  if constexpr (EPI == SITK_EPI_DGELU) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int c = it * 64 + lane, row = c / 12, cc = c % 12, m = m0 + wm * 32 + 16 * j + row;
        urow[j][it] = u32x4{0u, 0u, 0u, 0u};
        if (m < p.M)
          urow[j][it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.aux) + (size_t)map_row(p.auxmap, m) * p.ldaux +
                                                        n0 + wn * 96 + cc * 8);
      }
  }
  __builtin_amdgcn_s_barrier();
  SITK_NST(5)
  // epilogue through a wave-private staging area (the k-tile stages are free now)
  char* slot = smem + wave * 6656;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int mrow = m0 + wm * 32 + 16 * j;
    f32x4 v1[6], v2[6];
    if constexpr (EPI == SITK_EPI_DGELU) {
      constexpr int PU = 96 * 2 + 16;                              // padded row pitch of the u block in the slot
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int c = it * 64 + lane;
        *reinterpret_cast<u32x4*>(slot + (c / 12) * PU + (c % 12) * 16) = urow[j][it];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const f32x4 u = load4(reinterpret_cast<const T*>(slot + fr * PU) + 16 * i + 4 * fq);
        v1[i] = acc[i][j];
        if (p.bias) v1[i] += load4(p.bias + n0 + wn * 96 + 16 * i + 4 * fq);
        v1[i] *= u + 0.5f;                                       // u = the gelu'(.) - 1/2 values the forward pass saved
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int n = n0 + wn * 96 + 16 * i + 4 * fq, m = mrow + fr;
        v1[i] = acc[i][j];
        v2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M && n < p.N) epilogue_math<T, EPI, true>(p, m, n, v1[i], v2[i]);     // (the residual joins in row layout below)
      }
    }
    staged_rows_store<TO, 96, EPI == SITK_EPI_BIAS_RES>(slot, v1, reinterpret_cast<TO*>(p.out), p, mrow, n0 + wn * 96, lane);
    if constexpr (EPI == SITK_EPI_BIAS_GELU) staged_rows_store<T, 96>(slot, v2, reinterpret_cast<T*>(p.out2), p, mrow, n0 + wn * 96, lane);
  }
#ifdef SITK_N192_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SITK_NST(6)                                                  // epilogue incl. store drain
  nst[7] = ntp - nt0;
  if (blockIdx.x == gridDim.x / 2 + 3 && lane == 0)
    for (int i = 0; i < 8; ++i) g_n192_stamps[wave * 8 + i] = nst[i];
#endif
}
#ifdef SITK_N192_STAMPS
extern "C" int sitk_n192_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_n192_stamps), sizeof(g_n192_stamps)) == hipSuccess ? 0 : -1;
}
#endif

template <typename TO, int EPI>
static int launch_gemm_nt_n192(const GemmParams& p, hipStream_t s) {
  const int tiles_n = p.N / 192;
  const int g128 = cdiv(p.M, 128) * tiles_n, g96 = cdiv(p.M, 96) * tiles_n;
  // More than one round of tiles (dim 384 / 768: 1 900 - 7 700 tiles): a 2-stage ring of 80 KB and <= 128 registers, so
  // that TWO workgroups share a CU and each runs its MFMAs under the other's ring fill and epilogue (with K = 384 a
  // tile has six k-steps: a workgroup alone on its CU spends as long filling and storing as multiplying).  Measured on
  // the eight encoder GEMMs (tools/gemm_bench.py, 40 992 tokens): dim 384 717 -> 622 us, dim 768 2019 -> 1628 us; a
  // persistent one-workgroup-per-CU form whose k-tile stream ran across tile boundaries gained 3 % / 0 % (its epilogue
  // still stalls the only workgroup of the CU).  One round: the 4-stage ring, tile height by CU coverage.
  if (g128 > 256) {
    hipLaunchKernelGGL((gemm_nt_n192_kernel<TO, EPI, 4, 2, 4>), dim3(g128), dim3(512), 0, s, p);
  } else if (g128 < 205 && g96 <= 256) {
    hipLaunchKernelGGL((gemm_nt_n192_kernel<TO, EPI, 3>), dim3(g96), dim3(384), 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_nt_n192_kernel<TO, EPI, 4>), dim3(g128), dim3(512), 0, s, p);
  }
  return check_launch("gemm_nt_n192");
}

template <typename TO, int EPI, int BN, int KT>
static int launch_gemm_nt_wres_cfg(const GemmParams& p, hipStream_t s) {
  constexpr int lds16 = KT * BN * 128 + 16 * wres_slot_bytes<TO, BN, KT, 1>();
  const int tiles_n = cdiv(p.N, BN);
  if constexpr (lds16 <= 160 * 1024) {
    const int groups = std::max(1, std::min(256 / tiles_n, cdiv(cdiv(p.M, 16), 16)));
    hipLaunchKernelGGL((gemm_nt_wres_kernel<TO, EPI, BN, KT, 16, 1>), dim3(groups * tiles_n), dim3(1024), 0, s, p, groups);
  } else {
    const int groups = std::max(1, std::min(256 / tiles_n, cdiv(cdiv(p.M, 32), 8)));
    hipLaunchKernelGGL((gemm_nt_wres_kernel<TO, EPI, BN, KT, 8, 2>), dim3(groups * tiles_n), dim3(512), 0, s, p, groups);
  }
  return check_launch("gemm_nt_wres");
}

template <typename TO, int EPI>
static int launch_gemm_nt_wres(const GemmParams& p, hipStream_t s) {
  const int KT = cdiv(p.K, 64);
  if (p.N % 128 == 0) {
    if (KT == 1) return launch_gemm_nt_wres_cfg<TO, EPI, 128, 1>(p, s);
    if (KT == 2) return launch_gemm_nt_wres_cfg<TO, EPI, 128, 2>(p, s);
    return launch_gemm_nt_wres_cfg<TO, EPI, 128, 3>(p, s);
  }
  if (KT == 1) return launch_gemm_nt_wres_cfg<TO, EPI, 64, 1>(p, s);
  if (KT == 2) return launch_gemm_nt_wres_cfg<TO, EPI, 64, 2>(p, s);
  return launch_gemm_nt_wres_cfg<TO, EPI, 64, 3>(p, s);
}

template <typename T, typename TA, typename TO, int EPI>
static int launch_gemm_nt(const GemmParams& p, hipStream_t s) {
  if constexpr (std::is_same<T, h16>::value && std::is_same<TA, h16>::value) {
    // weight-resident streaming kernel: K up to 192, 16-byte aligned rows, enough tokens to stream
    if (p.K <= 192 && p.K % 8 == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 && p.M >= 1024 && p.N % 8 == 0 &&
        (sizeof(TO) == 4 || p.ldo % 8 == 0))
      return launch_gemm_nt_wres<TO, EPI>(p, s);
    if (p.K > 192 && p.N % 192 == 0 && p.K % 8 == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 && p.M >= 1024 &&
        (sizeof(TO) == 4 || p.ldo % 8 == 0) && (EPI != SITK_EPI_DGELU || p.ldaux % 8 == 0))
      return launch_gemm_nt_n192<TO, EPI>(p, s);
  }
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
struct TrFrag<h16> {
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
  const int ntiles = tiles_k * ((p.N + 63) / 64);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);   // token-chunk major: a chunk's tiles share an XCD's L2
  const int tile = lid % ntiles, n0 = (tile / tiles_k) * 64, k0 = (tile % tiles_k) * 64;
  const int mbeg = (lid / ntiles) * p.chunk, mend = min(p.M, mbeg + p.chunk);
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
  dim3 grid(tiles * splits);
  constexpr bool is_f32 = sizeof(T) == 4;
  if (d->dy_is_f32 && !is_f32)
    hipLaunchKernelGGL((wgrad_kernel<T, float>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, T>), grid, dim3(256), 0, s, p);
  return check_launch("gemm_wgrad");
}

// ------------------------------------------------------------------------------------------
// Grouped weight gradient, LDS-DMA pipeline (the fast path; operands in the compute dtype).
//   * up to WG_MAX_PROBLEMS independent problems (the four Linears of an encoder layer) in ONE launch;
//   * output tile 64 (n) x 64 (k) per workgroup.  The 4 waves split the TOKENS of a stage (wave w
//     owns mma step w), each accumulating the whole 64x64 tile (4x4 MFMA tiles): one transposed
//     fragment read per MFMA, half the LDS traffic of splitting the tile over the waves;
//   * stages (4 steps of tokens x 64 columns of dY and of X) go global -> LDS by
//     global_load_lds_dwordx4 (per-lane source address = swizzle + row map + zero page for
//     out-of-range rows), double buffered, counted vmcnt + raw s_barrier;
//   * the four partial tiles are summed through LDS and added to dW with full-row float atomics;
//     db comes from one extra MFMA per n-block against a ones fragment (k-tile 0 only).
// ------------------------------------------------------------------------------------------
constexpr int WG_MAX_PROBLEMS = 4;
struct WgProblem {
  const void* dY;
  const void* X;
  float* dW;
  float* db;
  int M, N, K, lddy, ldx, lddw;
  RowMap dymap, xmap;
  int tiles_k, block_begin, splits, chunk;
};
struct WgGroup {
  WgProblem p[WG_MAX_PROBLEMS];
  int count;
};

__device__ u32x4 g_zero_page[4];  // 64 zero bytes: source of out-of-range LDS-DMA lanes

template <typename T>
__global__ __launch_bounds__(256) void wgrad_group_kernel(WgGroup grp) {
  constexpr int EPV = Mma<T>::EPV;
  constexpr int TS = Mma<T>::KSTEP;            // tokens per mma step: 32 (bf16) / 16 (f32)
  constexpr int BT = 4 * TS;                   // tokens per stage
  constexpr int PC = 128 / (int)sizeof(T);     // columns per 128-byte panel
  constexpr int NP = 64 / PC;                  // panels per operand: 1 / 2
  constexpr int OPB = NP * BT * 128;           // bytes per operand stage = 16 KB
  constexpr int GL = 2 * OPB / 1024 / 4;       // LDS-DMA instructions per wave per stage = 8
  __shared__ __attribute__((aligned(256))) char smem[4 * OPB];   // 2 buffers x (dY, X) = 64 KB

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int pi = 0;
  // XCD-aware order: logical ids that are consecutive run on the same XCD, and the logical order
  // is token-chunk major, so all output tiles of one token chunk share that XCD's L2 (the chunk of
  // dY and X is then fetched from HBM/MALL once instead of once per tile).
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
#pragma unroll
  for (int i = 1; i < WG_MAX_PROBLEMS; ++i)
    if (i < grp.count && bid >= grp.p[i].block_begin) pi = i;
  const WgProblem P = grp.p[pi];   // one scalar load of the selected problem into SGPRs
  const int local = bid - P.block_begin;
  const int tiles = P.tiles_k * ((P.N + 63) / 64);
  const int split = local / tiles, tile = local % tiles;
  const int n0 = (tile / P.tiles_k) * 64, k0 = (tile % P.tiles_k) * 64;
  const int mbeg = split * P.chunk, mend = min(P.M, mbeg + P.chunk);
  const T* __restrict__ dY = reinterpret_cast<const T*>(P.dY);
  const T* __restrict__ X = reinterpret_cast<const T*>(P.X);
  const bool do_bias = P.db != nullptr && k0 == 0;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  // LDS byte addresses of this lane's transposed-read blocks (bf16 fast path): rows wave*TS + 8g + q
  // (+4 for the second read), column block i at 32 * (i ^ key(row)) + 8 * (lane & 3)  [= lds_off()]
  uint32_t tr_base = 0, tr_off[4] = {0u, 0u, 0u, 0u};
  if constexpr (sizeof(T) == 2) {
    static_assert(sizeof(T) != 2 || OPB == 16384, "asm offsets assume 16 KB operand stages");
    tr_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int rowl = wave * TS + 8 * (lane >> 4) + ((lane >> 2) & 3);
    const int key = ((rowl >> 1) & 1) | (((rowl >> 3) & 1) << 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) tr_off[i] = rowl * 128 + 32 * (i ^ key) + 8 * (lane & 3);
  }

  // LDS-DMA roles: one instruction moves 1 KiB = 8 rows x 128 B of one panel.  An operand stage is
  // PPO = 16 pieces; waves 0,1 fetch dY, waves 2,3 fetch X (operand choice is wave-uniform and hoisted).
  constexpr int PPO = OPB / 1024;
  static_assert(2 * GL == PPO, "two waves per operand");
  const int op = wave >> 1;
  const T* __restrict__ obase = op ? X : dY;
  const int old_ = op ? P.ldx : P.lddy, oc0 = op ? k0 : n0, oclim = op ? P.K : P.N;
  const RowMap omap = op ? P.xmap : P.dymap;
  auto issue = [&](int mt, int buf) {
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int w = (wave & 1) * GL + i;
      const int panel = w / (BT / 8), rb = w % (BT / 8);
      const int row = rb * 8 + (lane >> 3);                          // token within the stage
      const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      const int chunk = (lane & 7) ^ (key << 1);                     // logical 16-byte chunk this lane fetches
      const int col = oc0 + panel * PC + chunk * EPV;
      const int m = mt + row;
      const bool ok = m < mend && col < oclim;
      const T* src = ok ? obase + (size_t)map_row(omap, ok ? m : 0) * old_ + col : reinterpret_cast<const T*>(zero);
      char* dst = smem + buf * 2 * OPB + op * OPB + panel * BT * 128 + rb * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 ones;
  {
    T one[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) one[e] = from_f32<T>(1.0f);
    __builtin_memcpy(&ones, one, 16);
  }

  const int nstage = (mend - mbeg + BT - 1) / BT;
  if (nstage > 0) issue(mbeg, 0);
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstage) {
      issue(mbeg + (s + 1) * BT, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");     // stage s landed, s+1 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* sY = smem + buf * 2 * OPB;
    const char* sX = sY + OPB;
    u32x4 fy[4], fx[4];
    if constexpr (sizeof(T) == 2) {
      // The 16 transposed reads go through ONE asm statement (with their lgkmcnt wait): a read
      // the compiler can see would make it drain the in-flight LDS-DMA of the next stage
      // (s_waitcnt vmcnt(0)) before every stage.
      const uint32_t b = tr_base + buf * 2 * OPB;
      const uint32_t a0 = b + tr_off[0], a1 = b + tr_off[1], a2 = b + tr_off[2], a3 = b + tr_off[3];
      u32x2 y0l, y0h, y1l, y1h, y2l, y2h, y3l, y3h, x0l, x0h, x1l, x1h, x2l, x2h, x3l, x3h;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:512\n\t"
          "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:512\n\t"
          "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:512\n\t"
          "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:512\n\t"
          "ds_read_b64_tr_b16 %8, %16 offset:16384\n\tds_read_b64_tr_b16 %9, %16 offset:16896\n\t"
          "ds_read_b64_tr_b16 %10, %17 offset:16384\n\tds_read_b64_tr_b16 %11, %17 offset:16896\n\t"
          "ds_read_b64_tr_b16 %12, %18 offset:16384\n\tds_read_b64_tr_b16 %13, %18 offset:16896\n\t"
          "ds_read_b64_tr_b16 %14, %19 offset:16384\n\tds_read_b64_tr_b16 %15, %19 offset:16896\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(y0l), "=&v"(y0h), "=&v"(y1l), "=&v"(y1h), "=&v"(y2l), "=&v"(y2h), "=&v"(y3l), "=&v"(y3h),
            "=&v"(x0l), "=&v"(x0h), "=&v"(x1l), "=&v"(x1h), "=&v"(x2l), "=&v"(x2h), "=&v"(x3l), "=&v"(x3h)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      __builtin_amdgcn_sched_barrier(0);
      fy[0] = u32x4{y0l[0], y0l[1], y0h[0], y0h[1]};
      fy[1] = u32x4{y1l[0], y1l[1], y1h[0], y1h[1]};
      fy[2] = u32x4{y2l[0], y2l[1], y2h[0], y2h[1]};
      fy[3] = u32x4{y3l[0], y3l[1], y3h[0], y3h[1]};
      fx[0] = u32x4{x0l[0], x0l[1], x0h[0], x0h[1]};
      fx[1] = u32x4{x1l[0], x1l[1], x1h[0], x1h[1]};
      fx[2] = u32x4{x2l[0], x2l[1], x2h[0], x2h[1]};
      fx[3] = u32x4{x3l[0], x3l[1], x3h[0], x3h[1]};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fy[i] = TrFrag<T>::load_k8(sY + ((16 * i) / PC) * BT * 128, wave * TS, (16 * i) % PC, lane);
        fx[i] = TrFrag<T>::load_k8(sX + ((16 * i) / PC) * BT * 128, wave * TS, (16 * i) % PC, lane);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(fy[i], fx[j], acc[i][j]);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] = Mma<T>::mma(fy[i], ones, accb[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // buffer may be refilled
  }

  // ---- sum the 4 waves' partial tiles through LDS (fp32 64x64 = 16 KB each), then atomics ----
  float* red = reinterpret_cast<float*>(smem);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) red[wave * 4096 + (16 * i + 4 * fq + jj) * 64 + 16 * j + fr] = acc[i][j][jj];
  __syncthreads();
  // one wave-instruction adds 64 consecutive floats (256 contiguous bytes) of one dW row: the
  // full-rate float-atomic shape (4 x 64-byte requests, no partial lines)
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + wave, c = lane;
    const float v = red[row * 64 + c] + red[4096 + row * 64 + c] + red[8192 + row * 64 + c] + red[12288 + row * 64 + c];
    const int n = n0 + row, k = k0 + c;
    if (n < P.N && k < P.K) unsafeAtomicAdd(P.dW + (size_t)n * P.lddw + k, v);
  }
  if (do_bias && fr == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int n = n0 + 16 * i + 4 * fq + jj;
        if (n < P.N) unsafeAtomicAdd(P.db + n, accb[i][jj]);
      }
  }
}

static bool wgrad_fast_ok(const sitk_wgrad_desc* d, int dtype) {
  const int epv = dtype == SITK_H16 ? 8 : 4;
  return !(d->dy_is_f32 && dtype != SITK_F32) && d->N % epv == 0 && d->K % epv == 0 && d->lddy % epv == 0 &&
         d->ldx % epv == 0;
}

template <typename T>
static int launch_wgrad_group(const sitk_wgrad_desc* d, int count, hipStream_t s) {
  constexpr int BT = 4 * Mma<T>::KSTEP;
  WgGroup g;
  g.count = count;
  int tiles_total = 0;
  for (int i = 0; i < count; ++i) tiles_total += cdiv(d[i].N, 64) * cdiv(d[i].K, 64);
  // ~2 workgroups per CU in flight over the whole group; every problem gets the same token chunking
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    WgProblem& p = g.p[i];
    p.dY = d[i].dY; p.X = d[i].X; p.dW = d[i].dW; p.db = d[i].db;
    p.M = d[i].M; p.N = d[i].N; p.K = d[i].K; p.lddy = d[i].lddy; p.ldx = d[i].ldx; p.lddw = d[i].lddw;
    p.dymap = to_rowmap(d[i].dymap); p.xmap = to_rowmap(d[i].xmap);
    p.tiles_k = cdiv(p.K, 64);
    int splits = std::max(1, cdiv(512, tiles_total));
    splits = std::min(splits, std::max(1, p.M / (2 * BT)));
    p.chunk = cdiv(cdiv(p.M, splits), BT) * BT;
    p.splits = cdiv(p.M, p.chunk);
    p.block_begin = blocks;
    blocks += cdiv(p.N, 64) * p.tiles_k * p.splits;
  }
  hipLaunchKernelGGL((wgrad_group_kernel<T>), dim3(blocks), dim3(256), 0, s, g);
  return check_launch("gemm_wgrad_group");
}

}  // namespace sitk

SITK_F16_TWIN(sitk_gemm_wgrad_group)
extern "C" int sitk_gemm_wgrad_group(const sitk_wgrad_desc* d, int count, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gemm_wgrad_group, d, count, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(d != nullptr && count >= 1 && count <= WG_MAX_PROBLEMS, "gemm_wgrad_group: 1..%d problems", WG_MAX_PROBLEMS);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  bool fast = true;
  for (int i = 0; i < count; ++i) {
    SITK_REQUIRE(d[i].M > 0 && d[i].N > 0 && d[i].K > 0 && d[i].dY && d[i].X && d[i].dW, "gemm_wgrad_group: bad problem %d", i);
    fast = fast && wgrad_fast_ok(&d[i], dtype);
  }
  if (!fast) {  // generic path, one launch per problem
    for (int i = 0; i < count; ++i) SITK_TRY(sitk_gemm_wgrad(&d[i], dtype, stream));
    return SITK_OK;
  }
  if (dtype == SITK_H16) return launch_wgrad_group<h16>(d, count, s);
  if (dtype == SITK_F32) return launch_wgrad_group<float>(d, count, s);
  set_error("gemm_wgrad_group: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_gemm_nt)
extern "C" int sitk_gemm_nt(const sitk_gemm_desc* d, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gemm_nt, d, dtype, stream);
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
  if (dtype == SITK_H16) return dispatch_gemm_nt<h16>(d, s);
  if (dtype == SITK_F32) return dispatch_gemm_nt<float>(d, s);
  set_error("gemm_nt: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_gemm_wgrad)
extern "C" int sitk_gemm_wgrad(const sitk_wgrad_desc* d, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gemm_wgrad, d, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(d != nullptr, "gemm_wgrad: null descriptor");
  SITK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_wgrad: empty problem");
  SITK_REQUIRE(d->N % 4 == 0 && d->K % 8 == 0, "gemm_wgrad: N %% 4 and K %% 8 required (N=%d K=%d)", d->N, d->K);
  SITK_REQUIRE(d->lddy % 4 == 0 && d->ldx % 8 == 0, "gemm_wgrad: leading dims must keep 16-byte alignment");
  SITK_REQUIRE(d->dY && d->X && d->dW, "gemm_wgrad: null operand");
  if (wgrad_fast_ok(d, dtype)) return sitk_gemm_wgrad_group(d, 1, dtype, stream);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return dispatch_wgrad<h16>(d, s);
  if (dtype == SITK_F32) return dispatch_wgrad<float>(d, s);
  set_error("gemm_wgrad: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}
