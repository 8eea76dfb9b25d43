// sitk: what ln_gemm_fused.hip and mlp_fused.hip share -- the parameter block, LDS constants and fragment-read macros of the
// fused LayerNorm + to_qkv kernels, and the BODY of their backward kernel (d to_qkv + LayerNorm backward) as a device function,
// so that mlp_fused.hip can chain it with the MLP backward of the next layer in one launch (round 4).
#pragma once
#include "common.h"
#include "fused_epilogue.h"

namespace sitk {

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
// backward: dx = dres + LayerNorm'(dy W)
// ------------------------------------------------------------------------------------------------------
// The loop is nine chunks of 24 MFMAs per wave -- 0.2 us of matrix work per chunk against ~1.2 us for an LDS-DMA to
// land -- so it runs at the speed of its prefetch: a 4-slot ring keeps THREE chunks in flight (2 slots: 24.1 us per
// launch at the BASELINE shape; 4 slots: see profiles/README.md).  The B operand (the block's rows of dy, 64 columns per chunk) travels
// through the same ring: as register loads it would sit in the in-order vmcnt queue between the DMA pieces and force
// every older piece home with it.
constexpr int LG_BWD_SLOTS = 4;
template <int TG, int TT>
constexpr int lg_bwd_smem() {
  constexpr int RING = LG_BWD_SLOTS * (LG_WB + 16 * TT * TG * 128);
  return RING > LG_SMEM_BWD ? RING : LG_SMEM_BWD;
}
// the kernel's body as a device function (smem: lg_bwd_smem<TG, TT>() bytes, 256-byte aligned; zerop: 64 zero bytes in global
// memory): ln_gemm_bwd_kernel below is this and nothing else; mlp_fused.hip chains it with the MLP backward of the next layer
template <int TG, int TT>
SITK_DEV void ln_gemm_bwd_body(const LnGemmParams& p, char* smem, const h16* zerop) {
  constexpr int D = LG_D, BLK = 16 * TT * TG, NW = 2 * TG;
  constexpr int DYB = BLK * 128;                       // dy chunk image: BLK rows x 128 B
  constexpr int SLOT = LG_WB + DYB;                    // W^T chunk (24 KB) + dy chunk
  constexpr int NP = 24 + BLK / 8, PPW = NP / NW;      // DMA pieces per chunk (36 / 40) and per wave (6 / 5; 3 with 12 waves)
  static_assert(PPW * NW == NP, "pieces must divide evenly");
  static_assert(lg_bwd_smem<TG, TT>() <= 163840, "LDS budget");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int tg = wave >> 1, hh = wave & 1;
  const int blk0 = blockIdx.x * BLK;
  const int N = p.N, nchunks = N / 64;

  // ---- DMA pieces of one chunk: 24 of W^T (192 rows x 128 B) + BLK / 8 of dy (BLK rows x 128 B); PPW per wave ----
  const int r8 = lane >> 3;
  const h16* psrc[PPW];
  int pdst[PPW];
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


}  // namespace sitk
