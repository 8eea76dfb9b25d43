// sitk LayerNorm forward/backward and column reductions (HBM-bound kernels, one wave per row,
// wave64 shuffle reductions, 16-byte accesses).
#include <algorithm>

#include "common.h"

namespace sitk {

// One wave normalises one row at a time; a lane owns float4 groups c = lane, lane+64, ...
// NV = ceil(D / 256) groups per lane live in registers.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = D >> 2;
  f32x4 g[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) { g[i] = load4(gamma + 4 * c); b[i] = load4(beta + 4 * c); }
  }
  const float invD = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float* xr = x + row * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < nvec ? load4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; ss += d * d; }
      }
    }
    const float var = wave_sum(ss) * invD;
    const float rs = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[i][e] + b[i][e];
        store4(y + row * D + 4 * c, o);
      }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// dx = dres + rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat))
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy   (per-lane partials -> LDS -> one atomic per
// column per workgroup)
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* dres,
                                                            float* dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int64_t rows, int D) {
  __shared__ float red[2][4][NV * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = D >> 2;
  f32x4 g[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    g[i] = c < nvec ? load4(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NV], gy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
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
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2);
        if (dres) o += load4(dres + row * D + 4 * c);
        store4(dx + row * D + 4 * c, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave][(i * 64 + lane) * 4 + e] = dg[i][e];
      red[1][wave][(i * 64 + lane) * 4 + e] = db[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < NV * 256; c += 256) {
    // column index of slot (i, lane, e): 4 * (lane + 64 i) + e  == c when laid out as above
    if (c < D) {
      const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
      const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
      unsafeAtomicAdd(dgamma + c, a);
      unsafeAtomicAdd(dbeta + c, b);
    }
  }
}

// out[c] += sum_r in[r][c], optional row flags (row counted iff fa[r] && (fb == null || fb[r])).
template <typename TI>
__global__ __launch_bounds__(256) void colsum_kernel(const TI* __restrict__ in, int ld, const uint8_t* __restrict__ fa,
                                                     const uint8_t* __restrict__ fb, int64_t rows, int cols,
                                                     int64_t rows_per_block, float* __restrict__ out) {
  // thread -> 4 consecutive columns; grid.x covers column groups of 1024, grid.y covers row blocks
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c4 >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = std::min<int64_t>(rows, r0 + rows_per_block);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0; r < r1; ++r) {
    if (fa && !(fa[r] && (!fb || fb[r]))) continue;
    s += load4(in + r * ld + c4);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) unsafeAtomicAdd(out + c4 + e, s[e]);
}

template <typename TI>
static int launch_colsum(const TI* in, int ld, const uint8_t* fa, const uint8_t* fb, int64_t rows, int cols,
                         float* out, hipStream_t s) {
  const int gx = cdiv(cols, 1024);
  int64_t gy = std::max<int64_t>(1, std::min<int64_t>(cdiv64(rows, 8), 2048 / gx));
  const int64_t rpb = cdiv64(rows, gy);
  gy = cdiv64(rows, rpb);
  hipLaunchKernelGGL((colsum_kernel<TI>), dim3(gx, (unsigned)gy), dim3(256), 0, s, in, ld, fa, fb, rows, cols, rpb, out);
  return check_launch("colsum");
}

template <typename T>
static int dispatch_ln_fwd(const float* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                           int64_t rows, int D, hipStream_t s) {
  const int grid = (int)std::min<int64_t>(cdiv64(rows, 4), 4096);
  const int nv = cdiv(D, 256);
  T* yt = reinterpret_cast<T*>(y);
  switch (nv) {
    case 1: hipLaunchKernelGGL((layernorm_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, x, g, b, yt, mean, rstd, rows, D); break;
    case 2: hipLaunchKernelGGL((layernorm_fwd_kernel<T, 2>), dim3(grid), dim3(256), 0, s, x, g, b, yt, mean, rstd, rows, D); break;
    case 3: hipLaunchKernelGGL((layernorm_fwd_kernel<T, 3>), dim3(grid), dim3(256), 0, s, x, g, b, yt, mean, rstd, rows, D); break;
    case 4: hipLaunchKernelGGL((layernorm_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, x, g, b, yt, mean, rstd, rows, D); break;
    default: set_error("layernorm: D=%d > 1024 unsupported", D); return SITK_ERR_INVALID;
  }
  return check_launch("layernorm_fwd");
}

template <typename T>
static int dispatch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* g,
                           const float* dres, float* dx, float* dg, float* db, int64_t rows, int D, hipStream_t s) {
  // ~64 rows per workgroup keeps the atomic traffic at D*8 bytes per 64 rows
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(rows, 64), 2048));
  const int nv = cdiv(D, 256);
  const T* dyt = reinterpret_cast<const T*>(dy);
  switch (nv) {
    case 1: hipLaunchKernelGGL((layernorm_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, dyt, x, mean, rstd, g, dres, dx, dg, db, rows, D); break;
    case 2: hipLaunchKernelGGL((layernorm_bwd_kernel<T, 2>), dim3(grid), dim3(256), 0, s, dyt, x, mean, rstd, g, dres, dx, dg, db, rows, D); break;
    case 3: hipLaunchKernelGGL((layernorm_bwd_kernel<T, 3>), dim3(grid), dim3(256), 0, s, dyt, x, mean, rstd, g, dres, dx, dg, db, rows, D); break;
    case 4: hipLaunchKernelGGL((layernorm_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, dyt, x, mean, rstd, g, dres, dx, dg, db, rows, D); break;
    default: set_error("layernorm: D=%d > 1024 unsupported", D); return SITK_ERR_INVALID;
  }
  return check_launch("layernorm_bwd");
}

}  // namespace sitk

extern "C" int sitk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int64_t rows, int D, int dtype, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  SITK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0, "layernorm_fwd: rows=%lld D=%d (D %% 4 == 0 required)", (long long)rows, D);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_BF16) return dispatch_ln_fwd<bf16>(x, gamma, beta, y, mean, rstd, rows, D, s);
  if (dtype == SITK_F32) return dispatch_ln_fwd<float>(x, gamma, beta, y, mean, rstd, rows, D, s);
  set_error("layernorm_fwd: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

extern "C" int sitk_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* dres, float* dx_out, float* dgamma, float* dbeta,
                                  int64_t rows, int D, int dtype, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(dy && x && mean && rstd && gamma && dx_out && dgamma && dbeta, "layernorm_bwd: null pointer");
  SITK_REQUIRE(rows > 0 && D > 0 && D % 4 == 0, "layernorm_bwd: rows=%lld D=%d", (long long)rows, D);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_BF16) return dispatch_ln_bwd<bf16>(dy, x, mean, rstd, gamma, dres, dx_out, dgamma, dbeta, rows, D, s);
  if (dtype == SITK_F32) return dispatch_ln_bwd<float>(dy, x, mean, rstd, gamma, dres, dx_out, dgamma, dbeta, rows, D, s);
  set_error("layernorm_bwd: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

extern "C" int sitk_colsum_f32(const float* in, int64_t rows, int cols, int ld, float* out, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(in && out && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, "colsum: bad arguments");
  return launch_colsum<float>(in, ld, nullptr, nullptr, rows, cols, out, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sitk_masked_colsum(const void* in, int ld, int in_is_f32, int dtype, const uint8_t* flag_a,
                                  const uint8_t* flag_b, int64_t rows, int cols, float* out, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(in && out && flag_a && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, "masked_colsum: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (in_is_f32 || dtype == SITK_F32) return launch_colsum<float>(reinterpret_cast<const float*>(in), ld, flag_a, flag_b, rows, cols, out, s);
  return launch_colsum<bf16>(reinterpret_cast<const bf16*>(in), ld, flag_a, flag_b, rows, cols, out, s);
}
