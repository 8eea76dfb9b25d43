// Shared epilogue of the fused backward kernels (mlp_fused.hip, ln_gemm_fused.hip): LayerNorm backward
// of a 128-token block whose input gradient dh sits in MFMA accumulator layout.
//
//   dx = dres + rstd (dh gamma - mean(dh gamma) - xhat mean(dh gamma xhat)),  xhat = (x - mean) rstd
//   per-workgroup partial sums of dgamma = sum_rows dh xhat and dbeta = sum_rows dh
//
// In accumulator layout a wave instruction touches 16 rows x 64 bytes: half cache lines, the other half
// belonging to another wave.  Measured on MI355X the row-indexed traffic of this epilogue (x, dres in, dx and
// its compute-dtype copy out: 55 MB for the BASELINE shape, all workgroups at once) then runs at ~2.3 TB/s.
// So dh takes one trip through LDS into row-major fp32 (pitch 196 floats), and the LayerNorm backward runs
// like the stand-alone kernel of norm.hip: 16 lanes per row, every load and store a run of whole rows
// (256 contiguous bytes per row and instruction), row sums by DPP.
#pragma once
#include "common.h"

namespace sitk {

constexpr int FE_D = 192;
constexpr int FE_PITCH = 196;                              // floats; rows shift by 16 B per row in the 256-B bank window
// A workgroup owns 16 TT TG rows: TG token groups of TT 16-row MFMA tiles, 2 waves per group (the two halves of
// the feature dimension).  (TG, TT) = (4, 2), (3, 2) or (6, 1); see fused_block_rows().
constexpr int fe_smem_bytes(int tg) { return 32 * tg * FE_PITCH * 4; }   // row buffer (the column partials alias it)
constexpr int FE_SMEM_BYTES = fe_smem_bytes(4);            // 100352

// Rows per workgroup of the fused kernels for a problem of `rows` tokens.  Every such kernel is a single wave of
// workgroups (one per CU), so its duration is that of ONE workgroup: 96-row workgroups are 25 % shorter than
// 128-row ones as long as all of them still fit on the 256 CUs at once (BASELINE config 2: 20 544 rows = 214
// workgroups of 96 instead of 161 of 128).
// Larger problems run in rounds of 256 workgroups: the choice is the one with fewer rounds x rows -- 96 rows again wherever
// its last round costs less than what 128-row workgroups add to every round (B = 128: two rounds either way, 2 x 96 against
// 2 x 128).  Round 5: until then every problem of more than 256 x 96 rows took 128-row workgroups, which also sent it down the
// less fused path (the block-tail and chained backward kernels exist for 96 rows only): 4.78 ms per step at B = 128.
static inline int fused_block_rows(int64_t rows) {
  const int64_t r96 = ((rows + 95) / 96 + 255) / 256 * 96, r128 = ((rows + 127) / 128 + 255) / 256 * 128;
  return r96 <= r128 ? 96 : 128;
}
// the BACKWARD kernels' choice (d to_qkv + norm backward, MLP backward and their pair launch share one partition).  Diagnostic
// build: SITK_BWD_ROWS128=1 forces 128-row workgroups -- 161 instead of 214 at BASELINE config 2 -- to measure what room for the
// all-reduce channels of a data-parallel step costs (tools/dp_cu_budget.py); the shipped library takes fused_block_rows().
static inline int fused_bwd_block_rows(int64_t rows) {
  static const int force128 = sitk_ab_switch("SITK_BWD_ROWS128", 0);
  return force128 ? 128 : fused_block_rows(rows);
}

// v[i][t]: this wave's finished half of dh -- features 96 hh + 16 i + 4 fq + e of token 16 TT tg + 16 t + fr
// (pair exchange already done).  smem: >= FE_SMEM_BYTES, free for use by every wave (callers sync before).
// All threads of the workgroup must call it.
template <int TG, int TT = 2>
SITK_DEV void ln_bwd_rows_epilogue(char* smem, const f32x4 (&v)[6][TT], int tid, int blk0, int R, const float* __restrict__ x,
                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ dres, float* __restrict__ dx,
                                   h16* __restrict__ dxc, float* __restrict__ partials_block) {
  constexpr int D = FE_D;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, tg = wave >> 1, hh = wave & 1;
  const int j = lane & 15, sub = lane >> 4;                   // row pass: 16 lanes per row, 4 rows per pass
  float* rowbuf = reinterpret_cast<float*>(smem);

  // ---- request this wave's 8 TT rows of x and dres (2 TT passes x 3 x 16 B per lane each), statistics, gamma ----
  constexpr int BLK = 16 * TT * TG;
  const size_t nrows = (size_t)(R - blk0 < BLK ? R - blk0 : BLK);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)blk0 * D, 0,
                                                                       (int)(nrows * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dres ? dres : x) + (size_t)blk0 * D, 0,
                                                                       dres ? (int)(nrows * D * 4) : 0, 0x00020000);
  f32x4 xv[2 * TT][3], dv[2 * TT][3], gm[3];
  float mu[2 * TT], rs[2 * TT];
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass) {
    const int r = wave * (8 * TT) + pass * 4 + sub;
    const bool ok = blk0 + r < R;
    mu[pass] = ok ? mean[blk0 + r] : 0.f;
    rs[pass] = ok ? rstd[blk0 + r] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int off = (r * D + 4 * (j + 16 * i)) * 4;
      xv[pass][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, off, 0, 0));
      dv[pass][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_d, off, 0, 0));   // 0 when dres == NULL
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) gm[i] = load4(gamma + 4 * (j + 16 * i));

  // ---- dh: accumulator layout -> row-major fp32 (the callers' exchange area is being overwritten) ----
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int i = 0; i < 6; ++i)
      *reinterpret_cast<f32x4*>(rowbuf + (16 * TT * tg + 16 * t + fr) * FE_PITCH + 96 * hh + 16 * i + 4 * fq) = v[i][t];
  __syncthreads();

  // ---- LayerNorm backward, 4 rows per pass ----
  f32x4 dgs[3], dbs[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { dgs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dbs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 outv[2 * TT][3];
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass) {
    const int r = wave * (8 * TT) + pass * 4 + sub;
    f32x4 dh[3], xh[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dh[i] = *reinterpret_cast<const f32x4*>(rowbuf + r * FE_PITCH + 4 * (j + 16 * i));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[i][e] = (xv[pass][i][e] - mu[pass]) * rs[pass];    // rows past R: x = 0, mu = rs = 0, dh = 0
        const float gy = dh[i][e] * gm[i][e];
        s1 += gy;
        s2 += gy * xh[i][e];
        dgs[i][e] += dh[i][e] * xh[i][e];
        dbs[i][e] += dh[i][e];
      }
    }
    const float m1 = row16_sum(s1) * (1.0f / D), m2 = row16_sum(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        outv[pass][i][e] = rs[pass] * (dh[i][e] * gm[i][e] - m1 - xh[i][e] * m2) + dv[pass][i][e];
  }
  // ---- stores: whole rows ----
  const __amdgpu_buffer_rsrc_t r_o = __builtin_amdgcn_make_buffer_rsrc(dx + (size_t)blk0 * D, 0, (int)(nrows * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc(dxc ? dxc + (size_t)blk0 * D : (h16*)dx, 0,
                                                                       dxc ? (int)(nrows * D * 2) : 0, 0x00020000);
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass) {
    const int r = wave * (8 * TT) + pass * 4 + sub;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int col = 4 * (j + 16 * i);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[pass][i]), r_o, (r * D + col) * 4, 0, 0);
      h16x4 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) ob[e] = (h16)outv[pass][i][e];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), r_c, (r * D + col) * 2, 0, 0);   // dropped when dxc == NULL
    }
  }
  // ---- dgamma / dbeta partials of the block: [wave][sub] slots in LDS, then one thread per column ----
  __syncthreads();                                            // every row of dh has been read
  float* colbuf = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    *reinterpret_cast<f32x4*>(colbuf + ((wave * 4 + sub) * 2 + 0) * D + 4 * (j + 16 * i)) = dgs[i];
    *reinterpret_cast<f32x4*>(colbuf + ((wave * 4 + sub) * 2 + 1) * D + 4 * (j + 16 * i)) = dbs[i];
  }
  __syncthreads();
  if (tid < 2 * D) {
    float s = 0.f;
#pragma unroll 8
    for (int q = 0; q < 8 * TG; ++q) s += colbuf[q * 2 * D + tid];
    partials_block[tid] = s;
  }
}

// Forward counterpart: out = v + bias + x, rows of x re-read and rows of out written whole (same layouts as above).
template <int TG, int TT = 2>
SITK_DEV void residual_rows_epilogue(char* smem, const f32x4 (&v)[6][TT], int tid, int blk0, int R, const float* __restrict__ x,
                                     const float* __restrict__ bias, float* __restrict__ out) {
  constexpr int D = FE_D;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, tg = wave >> 1, hh = wave & 1;
  const int j = lane & 15, sub = lane >> 4;
  float* rowbuf = reinterpret_cast<float*>(smem);
  constexpr int BLK = 16 * TT * TG;
  const size_t nrows = (size_t)(R - blk0 < BLK ? R - blk0 : BLK);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)blk0 * D, 0,
                                                                       (int)(nrows * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_o = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blk0 * D, 0, (int)(nrows * D * 4), 0x00020000);
  f32x4 xv[2 * TT][3], bb[3];
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      xv[pass][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          r_x, ((wave * (8 * TT) + pass * 4 + sub) * D + 4 * (j + 16 * i)) * 4, 0, 0));
#pragma unroll
  for (int i = 0; i < 3; ++i) bb[i] = load4(bias + 4 * (j + 16 * i));
  __syncthreads();                                            // the callers' exchange area is being overwritten
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int i = 0; i < 6; ++i)
      *reinterpret_cast<f32x4*>(rowbuf + (16 * TT * tg + 16 * t + fr) * FE_PITCH + 96 * hh + 16 * i + 4 * fq) = v[i][t];
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass) {
    const int r = wave * (8 * TT) + pass * 4 + sub;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int col = 4 * (j + 16 * i);
      const f32x4 o = *reinterpret_cast<const f32x4*>(rowbuf + r * FE_PITCH + col) + bb[i] + xv[pass][i];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_o, (r * D + col) * 4, 0, 0);
    }
  }
}

// Prologue of the fused attention-output + MLP forward: v = o Wo^T sits in accumulator layout (same layout as
// above).  Rows take the trip through LDS, then per row: x_mid = v + bias + x (stored whole), LayerNorm
// statistics, h = LN(x_mid) into the operand strip `strip` ([k-panel][32 TG rows][128 B], swizzled with lds_off)
// and, when asked for, to global memory.  Ends with a workgroup barrier (the strip is complete).
template <int TG, int TT = 2>
SITK_DEV void proj_residual_ln_rows(char* rowbuf_bytes, char* strip, const f32x4 (&v)[6][TT], int tid, int blk0, int R,
                                    const float* __restrict__ x, const float* __restrict__ bias,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float* __restrict__ xmid, h16* __restrict__ h, float* __restrict__ mean,
                                    float* __restrict__ rstd) {
  constexpr int D = FE_D, BLK = 16 * TT * TG;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, tg = wave >> 1, hh = wave & 1;
  const int j = lane & 15, sub = lane >> 4;
  float* rowbuf = reinterpret_cast<float*>(rowbuf_bytes);
  const size_t nrows = (size_t)(R - blk0 < BLK ? R - blk0 : BLK);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)blk0 * D, 0,
                                                                       (int)(nrows * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_m = __builtin_amdgcn_make_buffer_rsrc(xmid + (size_t)blk0 * D, 0, (int)(nrows * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc(h ? h + (size_t)blk0 * D : (h16*)xmid, 0,
                                                                       h ? (int)(nrows * D * 2) : 0, 0x00020000);
  f32x4 xv[2 * TT][3], bb[3], gm[3], bt[3];
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      xv[pass][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          r_x, ((wave * (8 * TT) + pass * 4 + sub) * D + 4 * (j + 16 * i)) * 4, 0, 0));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    bb[i] = load4(bias + 4 * (j + 16 * i));
    gm[i] = load4(gamma + 4 * (j + 16 * i));
    bt[i] = load4(beta + 4 * (j + 16 * i));
  }
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int i = 0; i < 6; ++i)
      *reinterpret_cast<f32x4*>(rowbuf + (16 * TT * tg + 16 * t + fr) * FE_PITCH + 96 * hh + 16 * i + 4 * fq) = v[i][t];
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2 * TT; ++pass) {
    const int r = wave * (8 * TT) + pass * 4 + sub, row = blk0 + r;
    const bool ok = row < R;
    f32x4 xm[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int col = 4 * (j + 16 * i);
      xm[i] = *reinterpret_cast<const f32x4*>(rowbuf + r * FE_PITCH + col) + bb[i] + xv[pass][i];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xm[i]), r_m, (r * D + col) * 4, 0, 0);
      s += xm[i][0] + xm[i][1] + xm[i][2] + xm[i][3];
    }
    const float mu = row16_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = xm[i][e] - mu; ss += d * d; }
    const float rs = rsqrtf(row16_sum(ss) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c4 = j + 16 * i;
      h16x4 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) ob[e] = (h16)(ok ? (xm[i][e] - mu) * rs * gm[i][e] + bt[i][e] : 0.f);
      const int byte = c4 * 8;
      *reinterpret_cast<h16x4*>(strip + (byte >> 7) * (BLK * 128) + lds_off(r, byte & 127)) = ob;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), r_h, (r * D + 4 * c4) * 2, 0, 0);   // dropped when h == NULL
    }
    if (ok && j == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
  }
  __syncthreads();
}

}  // namespace sitk
