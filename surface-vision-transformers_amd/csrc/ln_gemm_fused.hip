// sitk fused LayerNorm + projection kernels of the attention half of an encoder block (bf16, dim = 192):
//
//   forward   h = LayerNorm(x) ;  y = h W^T                      layers.i.0.norm + layers.i.0.fn.to_qkv (no bias)
//   backward  dh = dy W ;  dx = dres + LayerNorm'(dh)             + per-workgroup dgamma / dbeta partials
//
// They are the two halves of the fused MLP kernel (mlp_fused.hip) taken apart: forward is its LayerNorm
// prologue + FIRST product with the accumulators stored instead of chained; backward is its SECOND product
// (B operand streamed from HBM instead of produced in registers) + LayerNorm-backward epilogue.  Same
// geometry: a workgroup owns 128 tokens, wave (tg, hh) = 32 tokens x one half of every 64-wide chunk of the
// streamed dimension; weight chunks (24 KB) travel global -> LDS by LDS-DMA through a 2-slot ring with one
// raw barrier per chunk; fragment reads sit in asm blocks and are software pipelined (see mlp_fused.hip for
// the layouts, the slot permutation that makes a lane's 8 accumulator values consecutive features, the
// buffer-descriptor row I/O and the store keep-alive).  One launch replaces LayerNorm + GEMM (forward) or
// GEMM + LayerNorm backward (backward) and the (tokens x 192) round trip through HBM between them.
#include <cstdlib>

#include "common.h"
#include "fused_epilogue.h"
#include "ln_gemm_bwd_body.h"

namespace sitk {

__device__ u32x4 g_zero_page_lg[4];
__device__ unsigned long long g_lg_stamps[8 * 16];   // diagnostic build only (SITK_LG_STAMPS)

// ------------------------------------------------------------------------------------------------------
// forward: y = LayerNorm(x) W^T
// ------------------------------------------------------------------------------------------------------
template <int TG>
__global__ __launch_bounds__(128 * TG) void ln_gemm_fwd_kernel(LnGemmParams p) {
  constexpr int D = LG_D, BLK = 32 * TG, NW = 2 * TG, PPW = 24 / NW;   // rows, waves, DMA pieces per wave and chunk
  __shared__ __attribute__((aligned(256))) char smem[LG_SMEM_FWD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int tg = wave >> 1, hh = wave & 1;
  const int blk0 = blockIdx.x * BLK;
  const int N = p.N, nchunks = N / 64;

  // ---- W chunk DMA: 24 pieces of 8 slot rows x 128 B (3 k-panels x 64 rows), PPW per wave.  Slot row
  //      32 hs + 16 it + r holds output feature 32 hs + 8 (r >> 2) + 4 it + (r & 3) of the chunk ----
  const int r8 = lane >> 3;
  int soff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int qq = wave * PPW + i;
    const int kt = qq >> 3, s = (qq & 7) * 8 + r8;
    const int r = s & 15, it = (s >> 4) & 1, hs = s >> 5;
    const int feat = 32 * hs + 8 * (r >> 2) + 4 * it + (r & 3);
    const int key = ((s >> 1) & 1) | (((s >> 3) & 1) << 1);
    soff[i] = feat * D + kt * 64 + (((lane & 7) ^ (key << 1)) * 8);
  }
  auto issue = [&](int c, int buf) {
    char* base = smem + buf * LG_WB + wave * PPW * 1024;
    const h16* src = p.w + (size_t)c * 64 * D;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + soff[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
  };
  issue(0, 0);
#ifdef SITK_LG_STAMPS
  unsigned long long st[16];
  int sn = 0;
  st[sn++] = __builtin_amdgcn_s_memtime();
#define LGSTAMP() st[sn++] = __builtin_amdgcn_s_memtime();
#else
#define LGSTAMP()
#endif

  // per-workgroup buffer descriptors (see mlp_fused.hip): rows past R read 0 / are not written
  const size_t nrows = (size_t)(p.R - blk0 < BLK ? p.R - blk0 : BLK);
  const size_t RD = nrows * D, oD = (size_t)blk0 * D;
  const __amdgpu_buffer_rsrc_t r_x = lg_rsrc(p.x + oD, RD * 4);
  const __amdgpu_buffer_rsrc_t r_h = lg_rsrc(p.h + oD, p.h ? RD * 2 : 0);
  const __amdgpu_buffer_rsrc_t r_y = lg_rsrc(p.y + (size_t)blk0 * N, nrows * N * 2);

  // ---- LayerNorm of the block's rows: wave = 16 rows, 16 lanes per row, 4 rows per pass, all loads in flight ----
  char* sH = smem + LG_OFF_H;
  {
    const int j = lane & 15, sub = lane >> 4;
    f32x4 gm[3], bt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gm[i] = load4(p.gamma + 4 * (j + 16 * i)); bt[i] = load4(p.beta + 4 * (j + 16 * i)); }
    f32x4 v[4][3];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        v[pass][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            r_x, ((wave * 16 + pass * 4 + sub) * D + 4 * (j + 16 * i)) * 4, 0, 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LGSTAMP()   // 1: x loads landed
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = wave * 16 + pass * 4 + sub, row = blk0 + r;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) s += v[pass][i][0] + v[pass][i][1] + v[pass][i][2] + v[pass][i][3];
      s = row16_sum(s);
      const float mu = s * (1.0f / D);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[pass][i][e] - mu; ss += d * d; }
      ss = row16_sum(ss);
      const float rs = rsqrtf(ss * (1.0f / D) + 1e-5f);
      const bool ok = row < p.R;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c4 = j + 16 * i;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ok ? (v[pass][i][e] - mu) * rs * gm[i][e] + bt[i][e] : 0.f;
        const int byte = c4 * 8;
        const u32x2 ob = {lg_pack_h16(o[0], o[1]), lg_pack_h16(o[2], o[3])};
        *reinterpret_cast<u32x2*>(sH + (byte >> 7) * (BLK * 128) + lds_off(r, byte & 127)) = ob;
        if (p.h) __builtin_amdgcn_raw_buffer_store_b64(ob, r_h, (r * D + 4 * c4) * 2, 0, 0);
      }
      if (ok && j == 0 && p.mean) { p.mean[row] = mu; p.rstd[row] = rs; }
    }
    LGSTAMP()   // 2: LN computed, stores issued
    __syncthreads();
    LGSTAMP()   // 3: barrier
  }
  u32x4 hf[2][6];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < 6; ++k)
      hf[t][k] = *reinterpret_cast<const u32x4*>(sH + (k >> 1) * (BLK * 128) +
                                                 lds_off(32 * tg + 16 * t + fr, (k & 1) * 64 + fq * 16));

  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int keyl = ((fr >> 1) & 1) | (((fr >> 3) & 1) << 1);
  uint32_t aw[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) aw[ks] = lbase + (32 * hh + fr) * 128 + ((ks * 64 + fq * 16) ^ (keyl << 5));
  const int vo[2] = {((32 * tg + fr) * N + 32 * hh + 8 * fq) * 2, ((32 * tg + 16 + fr) * N + 32 * hh + 8 * fq) * 2};

  LGSTAMP()     // 4: frags loaded
  for (int c = 0; c < nchunks; ++c) {
    if (c == 1 || c == 5) { LGSTAMP() }   // 5: end of chunk 0 ; 6: end of chunk 4
    // chunk c's DMA precedes the previous iteration's 2 stores (pinned by the "memory" clobbers)
    if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint32_t bo = (c & 1) * LG_WB;
    const uint32_t a0 = aw[0] + bo, a1 = aw[1] + bo;
    u32x4 x0, x1, x2, x3, y0, y1, y2, y3;
    SITK_LG_ISSUE4(x0, x1, x2, x3, a0, a1, 0, 2048, 0, 2048);
    if (c + 1 < nchunks) issue(c + 1, (c + 1) & 1);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define SITK_LG_MMAS(KT, f0, f1, f2, f3)                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    acc[0][0] = Mma<h16>::mma(f0, hf[0][2 * KT], acc[0][0]);                                               \
    acc[0][1] = Mma<h16>::mma(f0, hf[1][2 * KT], acc[0][1]);                                               \
    acc[1][0] = Mma<h16>::mma(f1, hf[0][2 * KT], acc[1][0]);                                               \
    acc[1][1] = Mma<h16>::mma(f1, hf[1][2 * KT], acc[1][1]);                                               \
    acc[0][0] = Mma<h16>::mma(f2, hf[0][2 * KT + 1], acc[0][0]);                                           \
    acc[0][1] = Mma<h16>::mma(f2, hf[1][2 * KT + 1], acc[0][1]);                                           \
    acc[1][0] = Mma<h16>::mma(f3, hf[0][2 * KT + 1], acc[1][0]);                                           \
    acc[1][1] = Mma<h16>::mma(f3, hf[1][2 * KT + 1], acc[1][1]);                                           \
    __builtin_amdgcn_sched_barrier(0);
    SITK_LG_WAIT_ISSUE4(x0, x1, x2, x3, y0, y1, y2, y3, a0, a1, 8192, 10240, 8192, 10240);
    SITK_LG_MMAS(0, x0, x1, x2, x3)
    SITK_LG_WAIT_ISSUE4(y0, y1, y2, y3, x0, x1, x2, x3, a0, a1, 16384, 18432, 16384, 18432);
    SITK_LG_MMAS(1, y0, y1, y2, y3)
    SITK_LG_WAIT4(x0, x1, x2, x3);
    SITK_LG_MMAS(2, x0, x1, x2, x3)
#undef SITK_LG_MMAS
    // lane holds features c*64 + 32 hh + 8 fq + 4 i + e of token 32 tg + 16 t + fr: one 16-byte store per tile
    u32x4 sd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 v0 = acc[0][t], v1 = acc[1][t];
      sd[t] = u32x4{lg_pack_h16(v0[0], v0[1]), lg_pack_h16(v0[2], v0[3]), lg_pack_h16(v1[0], v1[1]), lg_pack_h16(v1[2], v1[3])};
      __builtin_amdgcn_raw_buffer_store_b128(sd[t], r_y, vo[t], c * 128, 0);
    }
    asm volatile("" : : "v"(sd[0]), "v"(sd[1]));             // store keep-alive (mlp_fused.hip)
  }
  LGSTAMP()     // 7: loop done
#ifdef SITK_LG_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st[sn++] = __builtin_amdgcn_s_memtime();   // 8: stores drained
  if (blockIdx.x == 80 && lane == 0)
    for (int i = 0; i < sn; ++i) g_lg_stamps[wave * 16 + i] = st[i] - st[0];
#endif
}

// ------------------------------------------------------------------------------------------------------
// backward: dx = dres + LayerNorm'(dy W) -- the body lives in ln_gemm_bwd_body.h
// ------------------------------------------------------------------------------------------------------
template <int TG, int TT = 2>
__global__ __launch_bounds__(128 * TG) void ln_gemm_bwd_kernel(LnGemmParams p) {
  __shared__ __attribute__((aligned(256))) char smem[lg_bwd_smem<TG, TT>()];
  ln_gemm_bwd_body<TG, TT>(p, smem, reinterpret_cast<const h16*>(g_zero_page_lg));
}

static int lg_check(const char* what, int64_t rows, int D, int N, int dtype) {
  SITK_REQUIRE(dtype == SITK_H16 && D == LG_D && N % 64 == 0 && N >= 64 && rows > 0 && rows * (int64_t)N < (1ll << 30),
               "%s: the fused path is specialised for h16, dim 192, N %% 64 == 0 (got dtype %d dim %d N %d)", what, dtype, D, N);
  return SITK_OK;
}

}  // namespace sitk

using namespace sitk;

#ifdef SITK_LG_STAMPS
extern "C" int sitk_lg_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lg_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

SITK_F16_TWIN(sitk_ln_gemm_fused_supported)
extern "C" int sitk_ln_gemm_fused_supported(int D, int N, int dtype) {
  SITK_FORWARD_F16(dtype, sitk_ln_gemm_fused_supported, D, N, dtype);
  return dtype == SITK_H16 && D == LG_D && N % 64 == 0 && N >= 64;
}

SITK_F16_TWIN(sitk_ln_gemm_fwd)
extern "C" int sitk_ln_gemm_fwd(const float* x, const float* ln_w, const float* ln_b, const void* w_c, void* h, float* mean,
                                float* rstd, void* y, int64_t rows, int D, int N, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_ln_gemm_fwd, x, ln_w, ln_b, w_c, h, mean, rstd, y, rows, D, N, dtype, stream);
  SITK_REQUIRE(x && ln_w && ln_b && w_c && y, "ln_gemm_fwd: null pointer");
  SITK_REQUIRE((mean == nullptr) == (rstd == nullptr), "ln_gemm_fwd: mean and rstd go together");
  SITK_TRY(lg_check("ln_gemm_fwd", rows, D, N, dtype));
  LnGemmParams p = {};
  p.x = x; p.gamma = ln_w; p.beta = ln_b; p.w = reinterpret_cast<const h16*>(w_c);
  p.h = reinterpret_cast<h16*>(h); p.mean = mean; p.rstd = rstd; p.y = reinterpret_cast<h16*>(y);
  p.R = (int)rows; p.N = N;
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  if (fused_block_rows(rows) == 96) hipLaunchKernelGGL(ln_gemm_fwd_kernel<3>, dim3(cdiv((int)rows, 96)), dim3(384), 0, hs, p);
  else hipLaunchKernelGGL(ln_gemm_fwd_kernel<4>, dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  return check_launch("ln_gemm_fwd");
}

extern "C" size_t sitk_ln_gemm_bwd_partial_floats(int64_t rows) {
  return rows > 0 ? (size_t)cdiv64(rows, fused_bwd_block_rows(rows)) * 2 * LG_D : 0;
}

SITK_F16_TWIN(sitk_ln_gemm_bwd)
extern "C" int sitk_ln_gemm_bwd(const void* dy, const void* wt_c, const float* x, const float* mean, const float* rstd,
                                const float* ln_w, const float* dres, float* dx, void* dx_c, float* partials, int64_t rows,
                                int D, int N, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_ln_gemm_bwd, dy, wt_c, x, mean, rstd, ln_w, dres, dx, dx_c, partials, rows, D, N, dtype, stream);
  SITK_REQUIRE(dy && wt_c && x && mean && rstd && ln_w && dx && partials, "ln_gemm_bwd: null pointer");
  SITK_TRY(lg_check("ln_gemm_bwd", rows, D, N, dtype));
  LnGemmParams p = {};
  p.x = x; p.gamma = ln_w; p.w = reinterpret_cast<const h16*>(wt_c);
  p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
  p.y = const_cast<h16*>(reinterpret_cast<const h16*>(dy));
  p.dres = dres; p.dx = dx; p.dxc = reinterpret_cast<h16*>(dx_c); p.partials = partials;
  p.R = (int)rows; p.N = N;
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  static const int tt1 = sitk_ab_switch("SITK_LG_TT1", 1);   // 12 waves x 16 tokens; 0: the 6 x 32 variant (A/B)
  if (fused_bwd_block_rows(rows) == 96 && tt1) hipLaunchKernelGGL((ln_gemm_bwd_kernel<6, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  else if (fused_bwd_block_rows(rows) == 96) hipLaunchKernelGGL(ln_gemm_bwd_kernel<3>, dim3(cdiv((int)rows, 96)), dim3(384), 0, hs, p);
  else hipLaunchKernelGGL(ln_gemm_bwd_kernel<4>, dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  return check_launch("ln_gemm_bwd");
}
