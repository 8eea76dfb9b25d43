// sitk LayerNorm forward/backward and column reductions (HBM-bound kernels, one wave per row,
// wave64 shuffle reductions, 16-byte accesses).
#include <algorithm>

#include "common.h"

namespace sitk {

// Row layout: LPR lanes (16 / 32 / 64) share one row, so a wave normalises 64 / LPR rows at once;
// lane j of a row group owns float4 groups c = j, j + LPR, ... (NV of them, in registers).
// Row statistics are reduced with xor-shuffles inside the LPR-lane group.
template <int LPR>
SITK_DEV float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int D) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % LPR, sub = lane / LPR;
  const int nvec = D >> 2;
  f32x4 g[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = j + LPR * i;
    if (c < nvec) { g[i] = load4(gamma + 4 * c); b[i] = load4(beta + 4 * c); }
  }
  const float invD = 1.0f / (float)D;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; r0 < rows; r0 += (int64_t)gridDim.x * 4 * RPW) {
    const int64_t row = r0 + sub;
    const bool ok = row < rows;
    const float* xr = x + (ok ? row : 0) * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = j + LPR * i;
      v[i] = (ok && c < nvec) ? load4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = group_sum<LPR>(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = j + LPR * i;
      if (c < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; ss += d * d; }
      }
    }
    const float rs = rsqrtf(group_sum<LPR>(ss) * invD + 1e-5f);
    if (ok) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = j + LPR * i;
        if (c < nvec) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[i][e] + b[i][e];
          store4(y + row * D + 4 * c, o);
        }
      }
      if (j == 0) { mean[row] = mu; rstd[row] = rs; }
    }
  }
}

// dx = dres + rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat))
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy   (per-lane partials -> LDS -> one atomic per
// column per workgroup)
template <typename T, int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* dres,
                                                            float* dx, T* __restrict__ dx_c, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ partials,
                                                            int64_t rows, int D, int rows_per_block) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][4 * RPW][NV * LPR * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % LPR, sub = lane / LPR;
  const int nvec = D >> 2;
  f32x4 g[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = j + LPR * i;
    g[i] = c < nvec ? load4(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.0f / (float)D;
  const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
  const int64_t rend = rbeg + rows_per_block < rows ? rbeg + rows_per_block : rows;
  for (int64_t r0 = rbeg + wave * RPW; r0 < rend; r0 += 4 * RPW) {
    const int64_t row = r0 + sub;
    const bool ok = row < rend;
    const int64_t rr = ok ? row : rbeg;
    const float mu = mean[rr], rs = rstd[rr];
    f32x4 xh[NV], gy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = j + LPR * i;
      xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      gy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok && c < nvec) {
        const f32x4 xv = load4(x + row * D + 4 * c);
        const f32x4 dyv = load4(dy + row * D + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          gy[i][e] = dyv[e] * g[i][e];
          s1 += gy[i][e];
          s2 += gy[i][e] * xh[i][e];
          dg[i][e] += dyv[e] * xh[i][e];
          db[i][e] += dyv[e];
        }
      }
    }
    s1 = group_sum<LPR>(s1) * invD;
    s2 = group_sum<LPR>(s2) * invD;
    if (ok) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = j + LPR * i;
        if (c < nvec) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2);
          if (dres) o += load4(dres + row * D + 4 * c);
          store4(dx + row * D + 4 * c, o);
          if (dx_c) store4(dx_c + row * D + 4 * c, o);
        }
      }
    }
  }
  // slot (i, j, e) of row-group (wave, sub) holds column 4 * (j + LPR i) + e == (i * LPR + j) * 4 + e
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave * RPW + sub][(i * LPR + j) * 4 + e] = dg[i][e];
      red[1][wave * RPW + sub][(i * LPR + j) * 4 + e] = db[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 4 * RPW; ++r) { a += red[0][r][c]; b += red[1][r][c]; }
    if (partials) {  // plain stores; summed by ln_finalize_kernel (no contended atomics on 2 D addresses)
      partials[(size_t)blockIdx.x * 2 * D + c] = a;
      partials[(size_t)blockIdx.x * 2 * D + D + c] = b;
    } else {
      unsafeAtomicAdd(dgamma + c, a);
      unsafeAtomicAdd(dbeta + c, b);
    }
  }
}

// dgamma[c] += sum_b partials[b][c], dbeta[c] += sum_b partials[b][D + c], bitwise reproducible: a workgroup owns CG
// columns, its 256 / CG row lanes sum interleaved row sets (b = lane, lane + RL, ...), the lanes are folded through LDS
// in lane order and ONE thread per column adds the total to the gradient -- no atomics, a fixed order of additions.
template <int CG>
SITK_DEV void ln_finalize_body(const float* __restrict__ partials, int nblocks, int D, float* __restrict__ dgamma,
                               float* __restrict__ dbeta, f32x4* red) {
  // CG threads across, FOUR columns each (16-byte loads: a workgroup row is CG x 16 contiguous bytes; with one column per
  // thread config 5's 24 reductions read their 189 MB at 1.1 TB/s), 256 / CG row lanes; every column is summed in the
  // same order as before: lane tr takes blocks tr, tr + RL, ..., then the lanes are folded in lane order.
  constexpr int RL = 256 / CG;
  const int tc = threadIdx.x % CG, tr = threadIdx.x / CG;
  const int c = (blockIdx.x * CG + tc) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < 2 * D) {
    int b = tr;
    for (; b + 3 * RL < nblocks; b += 4 * RL) {                   // four independent loads in flight
      const f32x4 v0 = load4(partials + (size_t)b * 2 * D + c), v1 = load4(partials + (size_t)(b + RL) * 2 * D + c);
      const f32x4 v2 = load4(partials + (size_t)(b + 2 * RL) * 2 * D + c), v3 = load4(partials + (size_t)(b + 3 * RL) * 2 * D + c);
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < nblocks; b += RL) s += load4(partials + (size_t)b * 2 * D + c);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tr == 0 && c < 2 * D) {
    f32x4 t = red[tc];
    for (int k = 1; k < RL; ++k) t += red[k * CG + tc];
    float* dst = c < D ? dgamma + c : dbeta + (c - D);
    store4(dst, load4(dst) + t);
  }
}
template <int CG>
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ partials, int nblocks, int D,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ f32x4 red[256];
  ln_finalize_body<CG>(partials, nblocks, D, dgamma, dbeta, red);
}

// out[c] += sum_r in[r][c], optional row flags (row counted iff fa[r] && (fb == null || fb[r])).
// A workgroup covers a group of 4 * CG columns (CG threads across) with 256 / CG row lanes, sums its row range, folds
// the row lanes through LDS and issues ONE atomic per column: few workgroups per column group, because float atomics
// of many workgroups on the same few addresses serialise (2 048 workgroups on 192 addresses took 184 us for 15.8 MB).
template <typename TI, int NT = 256>
__global__ __launch_bounds__(NT) void colsum_kernel(const TI* __restrict__ in, int ld, const uint8_t* __restrict__ fa,
                                                     const uint8_t* __restrict__ fb, int64_t rows, int cols,
                                                     int64_t rows_per_block, float* __restrict__ out,
                                                     float* __restrict__ out2, int cols2, int cg) {
  __shared__ f32x4 red[NT];
  const int tc = threadIdx.x % cg, tr = threadIdx.x / cg, nr = NT / cg;
  const int c4 = (blockIdx.x * cg + tc) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = std::min<int64_t>(rows, r0 + rows_per_block);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < cols && tr < nr)
    for (int64_t r = r0 + tr; r < r1; r += 4 * nr) {          // four rows in flight per thread (flag-dependent loads)
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t rr = r + (int64_t)j * nr;
        const bool ok = rr < r1 && (!fa || (fa[rr] && (!fb || fb[rr])));
        v[j] = ok ? load4(in + rr * ld + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tr == 0 && c4 < cols) {
    for (int k = 1; k < nr; ++k) s += red[k * cg + tc];
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(out + c4 + e, s[e]);
    if (out2 && c4 < cols2) {      // the first cols2 sums once more (d cls_token = d pos_embedding[0])
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(out2 + c4 + e, s[e]);
    }
  }
}

// Short inputs (rows <= COLSUM_DET_ROWS: the per-sample sums of d pos_embedding / d cls_token over the batch): every column
// group is summed by ONE workgroup over ALL rows -- 64 column groups x 4 row lanes, the lanes folded through LDS in lane
// order, plain read-modify-write of the output by one thread per column.  No atomics: bitwise reproducible.
constexpr int COLSUM_DET_ROWS = 4096;
template <typename TI>
__global__ __launch_bounds__(256) void colsum_det_kernel(const TI* __restrict__ in, int ld, const uint8_t* __restrict__ fa,
                                                         const uint8_t* __restrict__ fb, int rows, int cols,
                                                         float* __restrict__ out, float* __restrict__ out2, int cols2) {
  __shared__ f32x4 red[256];
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const int c4 = (blockIdx.x * 64 + tc) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < cols)
    for (int r = tr; r < rows; r += 16) {                        // four rows in flight per thread
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rr = r + 4 * j;
        const bool ok = rr < rows && (!fa || (fa[rr] && (!fb || fb[rr])));
        v[j] = ok ? load4(in + (size_t)rr * ld + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      s += v[0]; s += v[1]; s += v[2]; s += v[3];
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tr == 0 && c4 < cols) {
    s += red[64 + tc]; s += red[128 + tc]; s += red[192 + tc];
    store4(out + c4, load4(out + c4) + s);
    if (out2 && c4 < cols2) store4(out2 + c4, load4(out2 + c4) + s);
  }
}

template <typename TI>
static int launch_colsum(const TI* in, int ld, const uint8_t* fa, const uint8_t* fb, int64_t rows, int cols,
                         float* out, hipStream_t s, float* out2 = nullptr, int cols2 = 0) {
  const int c4n = cols / 4;
  if (rows <= 512 || (rows <= COLSUM_DET_ROWS && c4n >= 1024)) {   // short: one workgroup per column group, no atomics
    hipLaunchKernelGGL((colsum_det_kernel<TI>), dim3(cdiv(c4n, 64)), dim3(256), 0, s, in, ld, fa, fb, (int)rows, cols, out, out2, cols2);
    return check_launch("colsum_det");
  }
  // threads across columns: 16 (a 256-byte row segment per workgroup row; a power of two covering cols / 4 when that is
  // less).  Narrow column groups mean MORE workgroups for the same number of atomics per output address: with one group
  // of 256 threads across the 768 columns of config 5's d mask_token sum only 64 workgroups read its 126 MB (266 us);
  // 12 groups x 42 row ranges read them in ~35 us
  int cg = 16;
  while (cg / 2 >= c4n && cg > 1) cg /= 2;
  const int gx = cdiv(c4n, cg);
  // row ranges: enough workgroups to fill the chip once, at most ~64 atomics per output address
  int64_t gy = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(cdiv64(rows, 256 / cg), 64), std::max(1, 512 / gx)));
  const int64_t rpb = cdiv64(rows, gy);
  gy = cdiv64(rows, rpb);
  // narrow and tall (the encoder-width sums over all tokens: 64 workgroups for 15.8 MB): 1 024 threads per workgroup, so
  // that the few workgroups the atomics allow still keep 64 rows each in flight
  if (cg <= 64 && rpb >= 256)
    hipLaunchKernelGGL((colsum_kernel<TI, 1024>), dim3(gx, (unsigned)gy), dim3(1024), 0, s, in, ld, fa, fb, rows, cols, rpb, out, out2, cols2, cg);
  else
    hipLaunchKernelGGL((colsum_kernel<TI>), dim3(gx, (unsigned)gy), dim3(256), 0, s, in, ld, fa, fb, rows, cols, rpb, out, out2, cols2, cg);
  return check_launch("colsum");
}

// lanes per row: smallest of 16 / 32 / 64 that keeps <= 4 float4 groups per lane
static bool ln_shape(int D, int& lpr, int& nv) {
  const int nvec = D / 4;
  for (int l = 16; l <= 64; l *= 2) {
    const int n = cdiv(nvec, l);
    if (n <= 4) { lpr = l; nv = n; return true; }
  }
  return false;
}

#define SITK_LN_CASES(KERNEL, ...)                                                            \
  switch (lpr * 8 + nv) {                                                                     \
    case 16 * 8 + 1: KERNEL(16, 1, __VA_ARGS__); break;                                       \
    case 16 * 8 + 2: KERNEL(16, 2, __VA_ARGS__); break;                                       \
    case 16 * 8 + 3: KERNEL(16, 3, __VA_ARGS__); break;                                       \
    case 16 * 8 + 4: KERNEL(16, 4, __VA_ARGS__); break;                                       \
    case 32 * 8 + 3: KERNEL(32, 3, __VA_ARGS__); break;                                       \
    case 32 * 8 + 4: KERNEL(32, 4, __VA_ARGS__); break;                                       \
    case 64 * 8 + 3: KERNEL(64, 3, __VA_ARGS__); break;                                       \
    case 64 * 8 + 4: KERNEL(64, 4, __VA_ARGS__); break;                                       \
    default: set_error("layernorm: D=%d unsupported (D <= 1024)", D); return SITK_ERR_INVALID; \
  }

template <typename T>
static int dispatch_ln_fwd(const float* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                           int64_t rows, int D, hipStream_t s) {
  int lpr = 0, nv = 0;
  if (!ln_shape(D, lpr, nv)) { set_error("layernorm: D=%d > 1024 unsupported", D); return SITK_ERR_INVALID; }
  const int rows_per_block = 4 * (64 / lpr);
  const int grid = (int)std::min<int64_t>(cdiv64(rows, rows_per_block), 8192);
  T* yt = reinterpret_cast<T*>(y);
#define SITK_LN_FWD(L, N, ...) \
  hipLaunchKernelGGL((layernorm_fwd_kernel<T, L, N>), dim3(grid), dim3(256), 0, s, x, g, b, yt, mean, rstd, rows, D)
  SITK_LN_CASES(SITK_LN_FWD, 0)
#undef SITK_LN_FWD
  return check_launch("layernorm_fwd");
}

// rows per workgroup / number of workgroups of the backward kernel (also sizes the partials scratch)
static void ln_bwd_grid(int64_t rows, int D, int& rows_per_block, int& grid) {
  int lpr = 16, nv = 1;
  ln_shape(D, lpr, nv);
  rows_per_block = 2 * 4 * (64 / lpr);  // 2 passes of the workgroup's 4 * RPW row slots
  if (cdiv64(rows, rows_per_block) > 4096) rows_per_block = (int)cdiv64(rows, 4096);
  grid = (int)cdiv64(rows, rows_per_block);
}

template <typename T>
static int dispatch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* g,
                           const float* dres, float* dx, void* dx_c, float* dg, float* db, float* partials,
                           int64_t rows, int D, hipStream_t s, bool finalize = true) {
  int lpr = 0, nv = 0;
  if (!ln_shape(D, lpr, nv)) { set_error("layernorm: D=%d > 1024 unsupported", D); return SITK_ERR_INVALID; }
  int rows_per_block, grid;
  ln_bwd_grid(rows, D, rows_per_block, grid);
  const T* dyt = reinterpret_cast<const T*>(dy);
  T* dxc = reinterpret_cast<T*>(dx_c);
#define SITK_LN_BWD(L, N, ...)                                                                                     \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, L, N>), dim3(grid), dim3(256), 0, s, dyt, x, mean, rstd, g, dres, dx, \
                     dxc, dg, db, partials, rows, D, rows_per_block)
  SITK_LN_CASES(SITK_LN_BWD, 0)
#undef SITK_LN_BWD
  SITK_LAUNCH_CHECK("layernorm_bwd");
  if (partials && finalize) {
    if (grid > 64) hipLaunchKernelGGL(ln_finalize_kernel<16>, dim3(cdiv(2 * D, 64)), dim3(256), 0, s, partials, grid, D, dg, db);
    else hipLaunchKernelGGL(ln_finalize_kernel<64>, dim3(cdiv(2 * D, 256)), dim3(256), 0, s, partials, grid, D, dg, db);
    SITK_LAUNCH_CHECK("layernorm_bwd_finalize");
  }
  return SITK_OK;
}

// All deferred LayerNorm parameter-gradient reductions of a backward slice in one launch (blockIdx.z = entry).
template <int CG>
__global__ __launch_bounds__(256) void ln_finalize_multi_kernel(LnFinalizeBatch batch, int nblocks, int D) {
  __shared__ f32x4 red[256];
  const LnFinalizeEntry e = batch.e[blockIdx.z];
  ln_finalize_body<CG>(e.partials, e.nblocks > 0 ? e.nblocks : nblocks, D, e.dgamma, e.dbeta, red);
}

int layernorm_finalize_multi(const LnFinalizeEntry* entries, int count, int64_t rows, int D, hipStream_t s) {
  int rpb, grid;
  ln_bwd_grid(rows, D, rpb, grid);
  for (int i0 = 0; i0 < count; i0 += LN_FINALIZE_MAX) {
    LnFinalizeBatch b;
    const int n = std::min(LN_FINALIZE_MAX, count - i0);
    for (int i = 0; i < n; ++i) b.e[i] = entries[i0 + i];
    int most = 0;
    for (int i = 0; i < n; ++i) most = std::max(most, b.e[i].nblocks > 0 ? b.e[i].nblocks : grid);
    if (most > 64) hipLaunchKernelGGL(ln_finalize_multi_kernel<16>, dim3(cdiv(2 * D, 64), 1, n), dim3(256), 0, s, b, grid, D);
    else hipLaunchKernelGGL(ln_finalize_multi_kernel<64>, dim3(cdiv(2 * D, 256), 1, n), dim3(256), 0, s, b, grid, D);
    SITK_LAUNCH_CHECK("layernorm_finalize_multi");
  }
  return SITK_OK;
}

int layernorm_bwd_deferred(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* dres, float* dx_out, void* dx_out_c, float* partials, int64_t rows, int D, int dtype,
                           hipStream_t s) {
  if (dtype == SITK_H16)
    return dispatch_ln_bwd<h16>(dy, x, mean, rstd, gamma, dres, dx_out, dx_out_c, nullptr, nullptr, partials, rows, D, s, false);
  return dispatch_ln_bwd<float>(dy, x, mean, rstd, gamma, dres, dx_out, dx_out_c, nullptr, nullptr, partials, rows, D, s, false);
}

}  // namespace sitk

SITK_F16_TWIN(sitk_layernorm_fwd)
extern "C" int sitk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int64_t rows, int D, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_layernorm_fwd, x, gamma, beta, y, mean, rstd, rows, D, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  SITK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0, "layernorm_fwd: rows=%lld D=%d (D %% 4 == 0 required)", (long long)rows, D);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return dispatch_ln_fwd<h16>(x, gamma, beta, y, mean, rstd, rows, D, s);
  if (dtype == SITK_F32) return dispatch_ln_fwd<float>(x, gamma, beta, y, mean, rstd, rows, D, s);
  set_error("layernorm_fwd: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

extern "C" size_t sitk_layernorm_bwd_partial_floats(int64_t rows, int D) {
  if (rows <= 0 || D <= 0 || D > 1024) return 0;
  int rpb, grid;
  sitk::ln_bwd_grid(rows, D, rpb, grid);
  return (size_t)grid * 2 * D;
}

SITK_F16_TWIN(sitk_layernorm_bwd)
extern "C" int sitk_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* dres, float* dx_out, void* dx_out_c, float* dgamma,
                                  float* dbeta, float* partials, int64_t rows, int D, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_layernorm_bwd, dy, x, mean, rstd, gamma, dres, dx_out, dx_out_c, dgamma, dbeta, partials, rows, D, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(dy && x && mean && rstd && gamma && dx_out && dgamma && dbeta, "layernorm_bwd: null pointer");
  SITK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0, "layernorm_bwd: rows=%lld D=%d", (long long)rows, D);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16)
    return dispatch_ln_bwd<h16>(dy, x, mean, rstd, gamma, dres, dx_out, dx_out_c, dgamma, dbeta, partials, rows, D, s);
  if (dtype == SITK_F32)
    return dispatch_ln_bwd<float>(dy, x, mean, rstd, gamma, dres, dx_out, dx_out_c, dgamma, dbeta, partials, rows, D, s);
  set_error("layernorm_bwd: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

extern "C" int sitk_colsum_f32(const float* in, int64_t rows, int cols, int ld, float* out, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(in && out && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, "colsum: bad arguments");
  return launch_colsum<float>(in, ld, nullptr, nullptr, rows, cols, out, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sitk_colsum_f32_dup(const float* in, int64_t rows, int cols, int ld, float* out, float* out2, int cols2,
                                   sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(in && out && out2 && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && cols2 > 0 && cols2 % 4 == 0 &&
               cols2 <= cols, "colsum_dup: bad arguments");
  return launch_colsum<float>(in, ld, nullptr, nullptr, rows, cols, out, reinterpret_cast<hipStream_t>(stream), out2, cols2);
}

SITK_F16_TWIN(sitk_masked_colsum)
extern "C" int sitk_masked_colsum(const void* in, int ld, int in_is_f32, int dtype, const uint8_t* flag_a,
                                  const uint8_t* flag_b, int64_t rows, int cols, float* out, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_masked_colsum, in, ld, in_is_f32, dtype, flag_a, flag_b, rows, cols, out, stream);
  using namespace sitk;
  SITK_REQUIRE(in && out && flag_a && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, "masked_colsum: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (in_is_f32 || dtype == SITK_F32) return launch_colsum<float>(reinterpret_cast<const float*>(in), ld, flag_a, flag_b, rows, cols, out, s);
  return launch_colsum<h16>(reinterpret_cast<const h16*>(in), ld, flag_a, flag_b, rows, cols, out, s);
}
