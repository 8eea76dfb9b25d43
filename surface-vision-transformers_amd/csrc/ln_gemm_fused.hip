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

namespace sitk {

__device__ u32x4 g_zero_page_lg[4];
__device__ unsigned long long g_lg_stamps[8 * 16];   // diagnostic build only (SITK_LG_STAMPS)

struct LnGemmParams {
  // forward                                   backward
  const float* x;      // (R,192) layer input            | same (saved)
  const float* gamma;  // LayerNorm weight
  const float* beta;   // LayerNorm bias                  | unused
  const h16* w;       // W (N,192), N = 3 heads 64       | W^T (192,N)
  h16* h;             // (R,192) LN output, saved        | unused
  float* mean;         // (R) written                     | read
  float* rstd;
  h16* y;             // (R,N) written                   | dy (R,N) read
  const float* dres;   // -                               | (R,192) fp32 residual gradient added to LN'(dh)
  float* dx;           // -                               | (R,192) fp32
  h16* dxc;           // -                               | (R,192) compute-dtype copy of dx
  float* partials;     // -                               | (gridDim.x, 2, 192)
  int R, N;
};

constexpr int LG_D = 192;
constexpr int LG_WB = 24576;                   // one weight chunk: 24 pieces of 8 rows x 128 B
constexpr int LG_OFF_H = 2 * LG_WB;            // forward: operand strip 3 k-panels x 128 rows x 128 B = 48 KB
constexpr int LG_SMEM_FWD = LG_OFF_H + 3 * 128 * 128;   // (TG = 4; TG = 3 uses the first 3 x 96 rows of every panel)
constexpr int LG_SMEM_BWD = FE_SMEM_BYTES > 8 * 12288 ? FE_SMEM_BYTES : 8 * 12288;   // exchange area / row-layout epilogue

SITK_DEV uint32_t lg_pack_h16(float a, float b) {
  h16x2 v;
  v[0] = (h16)a; v[1] = (h16)b;
  return __builtin_bit_cast(uint32_t, v);
}
SITK_DEV __amdgpu_buffer_rsrc_t lg_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

#define SITK_LG_WAIT_ISSUE4(c0, c1, c2, c3, n0, n1, n2, n3, aA, aB, oA0, oA1, oB0, oB1)                     \
  asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                  \
               "ds_read_b128 %4, %8 offset:" #oA0 "\n\tds_read_b128 %5, %8 offset:" #oA1 "\n\t"            \
               "ds_read_b128 %6, %9 offset:" #oB0 "\n\tds_read_b128 %7, %9 offset:" #oB1                   \
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)        \
               : "v"(aA), "v"(aB)                                                                          \
               : "memory")
#define SITK_LG_ISSUE4(n0, n1, n2, n3, aA, aB, oA0, oA1, oB0, oB1)                                          \
  asm volatile("ds_read_b128 %0, %4 offset:" #oA0 "\n\tds_read_b128 %1, %4 offset:" #oA1 "\n\t"            \
               "ds_read_b128 %2, %5 offset:" #oB0 "\n\tds_read_b128 %3, %5 offset:" #oB1                   \
               : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)                                                \
               : "v"(aA), "v"(aB)                                                                          \
               : "memory")
#define SITK_LG_WAIT4(c0, c1, c2, c3)                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory")

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
// backward: dx = dres + LayerNorm'(dy W)
// ------------------------------------------------------------------------------------------------------
// The loop is nine chunks of 24 MFMAs per wave -- 0.2 us of matrix work per chunk against ~1.2 us for an LDS-DMA to
// land -- so it runs at the speed of its prefetch: a 4-slot ring keeps THREE chunks in flight (2 slots: 24.1 us per
// launch at the BASELINE shape; 4 slots: see profiles/README.md).  The B operand (the block's rows of dy, 64 columns per chunk) travels
// through the same ring: as register loads it would sit in the in-order vmcnt queue between the DMA pieces and force
// every older piece home with it.
constexpr int LG_BWD_SLOTS = 4;
template <int TG, int TT = 2>
__global__ __launch_bounds__(128 * TG) void ln_gemm_bwd_kernel(LnGemmParams p) {
  constexpr int D = LG_D, BLK = 16 * TT * TG, NW = 2 * TG;
  constexpr int DYB = BLK * 128;                       // dy chunk image: BLK rows x 128 B
  constexpr int SLOT = LG_WB + DYB;                    // W^T chunk (24 KB) + dy chunk
  constexpr int NP = 24 + BLK / 8, PPW = NP / NW;      // DMA pieces per chunk (36 / 40) and per wave (6 / 5; 3 with 12 waves)
  static_assert(PPW * NW == NP, "pieces must divide evenly");
  constexpr int RING = LG_BWD_SLOTS * SLOT;
  constexpr int SMEM = RING > LG_SMEM_BWD ? RING : LG_SMEM_BWD;
  static_assert(SMEM <= 163840, "LDS budget");
  __shared__ __attribute__((aligned(256))) char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int tg = wave >> 1, hh = wave & 1;
  const int blk0 = blockIdx.x * BLK;
  const int N = p.N, nchunks = N / 64;

  // ---- DMA pieces of one chunk: 24 of W^T (192 rows x 128 B) + BLK / 8 of dy (BLK rows x 128 B); PPW per wave ----
  const int r8 = lane >> 3;
  const h16* psrc[PPW];
  int pdst[PPW];
  const h16* zerop = reinterpret_cast<const h16*>(g_zero_page_lg);
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wave * PPW + i;
    const int row = (q < 24 ? q : q - 24) * 8 + r8;
    const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int col = ((lane & 7) ^ (key << 1)) * 8;
    if (q < 24) {
      psrc[i] = p.w + (size_t)row * N + col;
      pdst[i] = q * 1024;
    } else {
      psrc[i] = blk0 + row < p.R ? p.y + (size_t)(blk0 + row) * N + col : nullptr;     // rows past R: zero page
      pdst[i] = LG_WB + (q - 24) * 1024;
    }
  }
  auto issue = [&](int c) {
    char* base = smem + (c % LG_BWD_SLOTS) * SLOT;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const h16* src = psrc[i] ? psrc[i] + (size_t)c * 64 : zerop;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + pdst[i]), 16, 0, 0);
    }
  };
#pragma unroll
  for (int c = 0; c < LG_BWD_SLOTS - 1; ++c)
    if (c < nchunks) issue(c);

  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int keyl = ((fr >> 1) & 1) | (((fr >> 3) & 1) << 1);
  const uint32_t aw2 = lbase + fr * 128 + ((hh * 64 + fq * 16) ^ (keyl << 5));        // + dt*2048 (+ slot)
  const uint32_t ab0 = lbase + LG_WB + (16 * TT * tg + fr) * 128 + ((hh * 64 + fq * 16) ^ (keyl << 5));   // token tile 0; tile 1: + 2048

  f32x4 yacc[12][TT];
#pragma unroll
  for (int dt = 0; dt < 12; ++dt)
#pragma unroll
    for (int t = 0; t < TT; ++t) yacc[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nchunks; ++c) {
    // chunk c has landed; chunks c + 1 .. c + SLOTS - 2 (PPW instructions each) may stay in flight
    const int ahead = min(LG_BWD_SLOTS - 2, nchunks - 1 - c);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // ... for every wave; slot (c - 1) % SLOTS is no longer read
    if (c + LG_BWD_SLOTS - 1 < nchunks) issue(c + LG_BWD_SLOTS - 1);
    const uint32_t bo = (c % LG_BWD_SLOTS) * SLOT;
    const uint32_t a2 = aw2 + bo, b2 = ab0 + bo;
    u32x4 x0, x1, x2, x3, y0, y1, y2, y3, pf[TT];
    if constexpr (TT == 2)
      asm volatile("ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:2048\n\t"
                   "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:2048\n\t"
                   "ds_read_b128 %2, %6 offset:4096\n\tds_read_b128 %3, %6 offset:6144"
                   : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(pf[0]), "=&v"(pf[TT - 1])
                   : "v"(a2), "v"(b2)
                   : "memory");
    else
      asm volatile("ds_read_b128 %4, %6\n\t"
                   "ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:2048\n\t"
                   "ds_read_b128 %2, %5 offset:4096\n\tds_read_b128 %3, %5 offset:6144"
                   : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(pf[0])
                   : "v"(a2), "v"(b2)
                   : "memory");
#define SITK_LG_MMAS2(J, f0, f1, f2, f3)                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 0][t] = Mma<h16>::mma(f0, pf[t], yacc[4 * J + 0][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 1][t] = Mma<h16>::mma(f1, pf[t], yacc[4 * J + 1][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 2][t] = Mma<h16>::mma(f2, pf[t], yacc[4 * J + 2][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 3][t] = Mma<h16>::mma(f3, pf[t], yacc[4 * J + 3][t]); \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TT == 2)
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   "ds_read_b128 %6, %10 offset:8192\n\tds_read_b128 %7, %10 offset:10240\n\t"
                   "ds_read_b128 %8, %10 offset:12288\n\tds_read_b128 %9, %10 offset:14336"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(pf[0]), "+v"(pf[TT - 1]), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                   : "v"(a2)
                   : "memory");
    else   // (one operand per register: listing pf[0] twice would make hipcc copy it while its read is in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   "ds_read_b128 %5, %9 offset:8192\n\tds_read_b128 %6, %9 offset:10240\n\t"
                   "ds_read_b128 %7, %9 offset:12288\n\tds_read_b128 %8, %9 offset:14336"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(pf[0]), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                   : "v"(a2)
                   : "memory");
    SITK_LG_MMAS2(0, y0, y1, y2, y3)
    SITK_LG_WAIT_ISSUE4(x0, x1, x2, x3, y0, y1, y2, y3, a2, a2, 16384, 18432, 20480, 22528);
    SITK_LG_MMAS2(1, x0, x1, x2, x3)
    SITK_LG_WAIT4(y0, y1, y2, y3);
    SITK_LG_MMAS2(2, y0, y1, y2, y3)
#undef SITK_LG_MMAS2
  }

  // ---- pair exchange: wave hh finishes features [96 hh, 96 hh + 96) (12 tiles x 1 KB per wave) ----
  __syncthreads();
  {
    char* mine = smem + wave * (6144 * TT);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int t = 0; t < TT; ++t)
        *reinterpret_cast<f32x4*>(mine + ((i * TT + t) * 64 + lane) * 16) = hh ? yacc[i][t] : yacc[6 + i][t];
  }
  __syncthreads();
  f32x4 v[6][TT];
  {
    const char* theirs = smem + (wave ^ 1) * (6144 * TT);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(theirs + ((i * TT + t) * 64 + lane) * 16);
        v[i][t] = (hh ? yacc[6 + i][t] : yacc[i][t]) + o;
      }
  }
  // ---- LayerNorm backward on dh = v, in row layout (fused_epilogue.h) ----
  ln_bwd_rows_epilogue<TG, TT>(smem, v, tid, blk0, p.R, p.x, p.mean, p.rstd, p.gamma, p.dres, p.dx, p.dxc,
                       p.partials + (size_t)blockIdx.x * 2 * D);
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
  return rows > 0 ? (size_t)cdiv64(rows, fused_block_rows(rows)) * 2 * LG_D : 0;
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
  if (fused_block_rows(rows) == 96 && tt1) hipLaunchKernelGGL((ln_gemm_bwd_kernel<6, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  else if (fused_block_rows(rows) == 96) hipLaunchKernelGGL(ln_gemm_bwd_kernel<3>, dim3(cdiv((int)rows, 96)), dim3(384), 0, hs, p);
  else hipLaunchKernelGGL(ln_gemm_bwd_kernel<4>, dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  return check_launch("ln_gemm_bwd");
}
