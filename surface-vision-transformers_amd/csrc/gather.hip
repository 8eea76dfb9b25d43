// sitk patch gather / layout kernels (HBM-bound integer-indexed copies).
//   gather_tokens : (B, 40962, C=4) channels-last surface + (P, V) vertex table -> tokens (B*P, ld)
//                   tools/preprocessing.py:74-84 fused with Rearrange('b c n v -> b n (v c)')
//   patchify      : reference input layout (B, C, P, V) -> tokens            (models/sit.py:47-49)
//   cast_rows, stage_weight : fp32 -> compute dtype staging (with padding / transpose)
#include <algorithm>

#include "common.h"

namespace sitk {

// One WAVE per token row (4 rows per workgroup in flight): a lane takes vertex slots v = lane, lane + 64, ...: it loads the vertex
// id (2 B, coalesced along v), one channels-last vertex record (C fp32 channels: one 16-byte load for C = 4) and writes C
// consecutive token features; the slots between V C and ld are the row's zero pad.  A surface is 164 C KB, so the 1.2x
// re-reads of shared edge / corner vertices are served by L2 (profiles/r04_pmc_gather.txt: FETCH_SIZE against the batch's
// bytes).  [Rounds 1 - 3 gave a 256-thread workgroup to every row: 160 of 256 threads busy, 20 480 workgroups; C = 4 only.]
// LDS staging of a patch's records (north_star's wording) would add a hop without removing a byte: every record is read
// once per patch it belongs to and written once; what the kernel needs is many independent 16-byte loads in flight.
template <typename T, int C>
__global__ __launch_bounds__(256) void gather_tokens_kernel(const float* __restrict__ x, const uint16_t* __restrict__ table,
                                                            T* __restrict__ tokens, int64_t rows, int n_vertices, int P, int V, int ld,
                                                            const float* __restrict__ mean, const float* __restrict__ stdv,
                                                            const int32_t* __restrict__ sample_idx = nullptr,
                                                            const float* __restrict__ targets_all = nullptr,
                                                            float* __restrict__ target_out = nullptr, int n_targets = 0) {
  // resident data set (tools/train.py:97-113,282): batch row b is sample sample_idx[b] of x; its n_targets labels ride along
  if (targets_all && blockIdx.x == 0) {
    const int64_t B = rows / P;
    for (int64_t i = threadIdx.x; i < B * n_targets; i += 256)
      target_out[i] = targets_all[(int64_t)sample_idx[i / n_targets] * n_targets + i % n_targets];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // optional per-channel normalisation (x - mean[c]) / std[c]  (tools/preprocessing.py:72); a true division
  // so that the result is bit-identical to the reference's numpy expression evaluated in fp32
  const bool norm = mean != nullptr;
  float mu[C], sd[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { mu[c] = norm ? mean[c] : 0.f; sd[c] = norm ? stdv[c] : 1.f; }
  const int K = V * C;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {  // row = b * P + p
    const int p = (int)(row % P);
    const int64_t b = sample_idx ? (int64_t)sample_idx[row / P] : row / P;
    const uint16_t* trow = table + (size_t)p * V;
    const float* xb = x + (size_t)b * n_vertices * C;
    T* orow = tokens + (size_t)row * ld;
    if constexpr (C == 4) {
      const int slots = ld >> 2;                      // 4-element slots per token row, the first V carry data, the rest zero pad
      for (int v0 = lane; v0 < slots; v0 += 256) {    // four slots per lane and pass: 4 ids, then 4 records in flight together
        int vid[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vid[j] = v0 + 64 * j < V ? (int)trow[v0 + 64 * j] : -1;
        f32x4 val[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          val[j] = vid[j] >= 0 ? *reinterpret_cast<const f32x4*>(xb + (size_t)vid[j] * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (norm && vid[j] >= 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[j][e] = (val[j][e] - mu[e]) / sd[e];
          }
          if (v0 + 64 * j < slots) store4(orow + 4 * (v0 + 64 * j), val[j]);
        }
      }
    } else {
      for (int v = lane; v < V; v += 64) {
        const float* rec = xb + (size_t)trow[v] * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float val = rec[c];
          if (norm) val = (val - mu[c]) / sd[c];
          orow[v * C + c] = from_f32<T>(val);
        }
      }
      for (int f = K + lane; f < ld; f += 64) orow[f] = from_f32<T>(0.f);
    }
  }
}

// (B, C, P, V) -> (B*P, ld): thread per (row, v) reads C strided channels (coalesced along v for
// each channel plane) and writes C consecutive features.
template <typename T, int C>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, T* __restrict__ tokens, int64_t rows, int P, int V, int ld) {
  const int K = V * C;
  for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
    const int p = (int)(row % P);
    const int64_t b = row / P;
    for (int v = blockIdx.x * 256 + threadIdx.x; v * C < ld; v += gridDim.x * 256) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int f = v * C + c;
        if (f < ld) {
          const float val = f < K ? x[(((size_t)b * C + c) * P + p) * V + v] : 0.f;
          tokens[(size_t)row * ld + f] = from_f32<T>(val);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                                        int64_t rows, int cols) {
  const int slots = ldd >> 2;
  const int64_t total = rows * slots;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / slots;
    const int c = (int)(i % slots) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c + 3 < cols) {
      v = load4(src + r * lds_ + c);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < cols) v[e] = src[r * lds_ + c + e];
    }
    store4(dst + r * ldd + c, v);
  }
}

// 32x32 LDS tile transpose + cast: w (rows, cols) fp32 -> w_t (cols, ldt) T
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ w, int rows, int cols, T* __restrict__ wt, int ldt) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? w[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;  // output row = c, col = r
    if (c < cols && r < ldt) wt[(size_t)c * ldt + r] = from_f32<T>(r < rows ? tile[tx][ty + 8 * i] : 0.f);
  }
}

template <typename T>
static int run_gather(const float* x, const uint16_t* table, void* tokens, int B, int nv, int C, int P, int V, int ld,
                      const float* mean, const float* stdv, hipStream_t s, const int32_t* sample_idx = nullptr,
                      const float* targets_all = nullptr, float* target_out = nullptr, int n_targets = 0) {
  const int64_t rows = (int64_t)B * P;
  const dim3 grid((unsigned)std::min<int64_t>(cdiv64(rows, 4), 16384));      // 4 rows (waves) per workgroup, grid-stride beyond 65 536 rows
  T* t = reinterpret_cast<T*>(tokens);
  switch (C) {
    case 1: hipLaunchKernelGGL((gather_tokens_kernel<T, 1>), grid, dim3(256), 0, s, x, table, t, rows, nv, P, V, ld, mean, stdv, sample_idx, targets_all, target_out, n_targets); break;
    case 2: hipLaunchKernelGGL((gather_tokens_kernel<T, 2>), grid, dim3(256), 0, s, x, table, t, rows, nv, P, V, ld, mean, stdv, sample_idx, targets_all, target_out, n_targets); break;
    case 3: hipLaunchKernelGGL((gather_tokens_kernel<T, 3>), grid, dim3(256), 0, s, x, table, t, rows, nv, P, V, ld, mean, stdv, sample_idx, targets_all, target_out, n_targets); break;
    case 4: hipLaunchKernelGGL((gather_tokens_kernel<T, 4>), grid, dim3(256), 0, s, x, table, t, rows, nv, P, V, ld, mean, stdv, sample_idx, targets_all, target_out, n_targets); break;
    default: set_error("gather_tokens: num_channels=%d unsupported (1..4)", C); return SITK_ERR_INVALID;
  }
  return check_launch("gather_tokens");
}

template <typename T>
static int run_patchify(const float* x, void* tokens, int B, int C, int P, int V, int ld, hipStream_t s) {
  const int64_t rows = (int64_t)B * P;
  dim3 grid(cdiv(cdiv(ld, C), 256), (unsigned)std::min<int64_t>(rows, 65535));
  T* t = reinterpret_cast<T*>(tokens);
  switch (C) {
    case 1: hipLaunchKernelGGL((patchify_kernel<T, 1>), grid, dim3(256), 0, s, x, t, rows, P, V, ld); break;
    case 2: hipLaunchKernelGGL((patchify_kernel<T, 2>), grid, dim3(256), 0, s, x, t, rows, P, V, ld); break;
    case 3: hipLaunchKernelGGL((patchify_kernel<T, 3>), grid, dim3(256), 0, s, x, t, rows, P, V, ld); break;
    case 4: hipLaunchKernelGGL((patchify_kernel<T, 4>), grid, dim3(256), 0, s, x, t, rows, P, V, ld); break;
    default: set_error("patchify: num_channels=%d unsupported (1..4)", C); return SITK_ERR_INVALID;
  }
  return check_launch("patchify");
}

template <typename T>
static int run_cast_rows(const float* src, int lds_, void* dst, int ldd, int64_t rows, int cols, hipStream_t s) {
  const int64_t total = rows * (ldd / 4);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(total, 256), 8192));
  hipLaunchKernelGGL((cast_rows_kernel<T>), dim3(grid), dim3(256), 0, s, src, lds_, reinterpret_cast<T*>(dst), ldd, rows, cols);
  return check_launch("cast_rows");
}

template <typename T>
static int run_stage_weight(const float* w, int rows, int cols, void* wc, int ldc, void* wt, int ldt, hipStream_t s) {
  if (wc) SITK_TRY(run_cast_rows<T>(w, cols, wc, ldc, rows, cols, s));
  if (wt) {
    dim3 grid(cdiv(cols, 32), cdiv(std::max(rows, ldt), 32));
    hipLaunchKernelGGL((transpose_cast_kernel<T>), grid, dim3(256), 0, s, w, rows, cols, reinterpret_cast<T*>(wt), ldt);
    SITK_LAUNCH_CHECK("stage_weight");
  }
  return SITK_OK;
}

}  // namespace sitk

SITK_F16_TWIN(sitk_gather_tokens)
extern "C" int sitk_gather_tokens(const float* x_bvc, const uint16_t* table_pv, void* tokens, int B, int n_vertices,
                                  int C, int P, int V, int ld, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gather_tokens, x_bvc, table_pv, tokens, B, n_vertices, C, P, V, ld, dtype, stream);
  return sitk_gather_tokens_norm(x_bvc, table_pv, nullptr, nullptr, tokens, B, n_vertices, C, P, V, ld, dtype, stream);
}

SITK_F16_TWIN(sitk_gather_tokens_norm)
extern "C" int sitk_gather_tokens_norm(const float* x_bvc, const uint16_t* table_pv, const float* mean, const float* stdv,
                                       void* tokens, int B, int n_vertices, int C, int P, int V, int ld, int dtype,
                                       sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gather_tokens_norm, x_bvc, table_pv, mean, stdv, tokens, B, n_vertices, C, P, V, ld, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(x_bvc && table_pv && tokens, "gather_tokens: null pointer");
  SITK_REQUIRE((mean == nullptr) == (stdv == nullptr), "gather_tokens_norm: mean and std go together");
  SITK_REQUIRE(C >= 1 && C <= 4, "gather_tokens: num_channels must be 1..4 (got %d)", C);
  SITK_REQUIRE(B > 0 && P > 0 && V > 0 && n_vertices > 0 && n_vertices <= 65536, "gather_tokens: bad shape");
  SITK_REQUIRE(ld >= V * C && ld % 4 == 0, "gather_tokens: ld=%d must be >= V*C=%d and a multiple of 4", ld, V * C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return run_gather<h16>(x_bvc, table_pv, tokens, B, n_vertices, C, P, V, ld, mean, stdv, s);
  if (dtype == SITK_F32) return run_gather<float>(x_bvc, table_pv, tokens, B, n_vertices, C, P, V, ld, mean, stdv, s);
  set_error("gather_tokens: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_gather_tokens_idx)
extern "C" int sitk_gather_tokens_idx(const float* x_all, const int32_t* sample_idx, const uint16_t* table_pv, const float* mean,
                                      const float* stdv, void* tokens, const float* targets_all, float* target_out,
                                      int n_targets, int B, int n_vertices, int C, int P, int V, int ld, int dtype,
                                      sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gather_tokens_idx, x_all, sample_idx, table_pv, mean, stdv, tokens, targets_all, target_out, n_targets, B, n_vertices, C, P, V, ld, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(x_all && sample_idx && table_pv && tokens, "gather_tokens_idx: null pointer");
  SITK_REQUIRE((mean == nullptr) == (stdv == nullptr), "gather_tokens_idx: mean and std go together");
  SITK_REQUIRE((targets_all == nullptr) == (target_out == nullptr) && (!targets_all || n_targets > 0),
               "gather_tokens_idx: targets_all, target_out and n_targets go together");
  SITK_REQUIRE(C >= 1 && C <= 4, "gather_tokens_idx: num_channels must be 1..4 (got %d)", C);
  SITK_REQUIRE(B > 0 && P > 0 && V > 0 && n_vertices > 0 && n_vertices <= 65536, "gather_tokens_idx: bad shape");
  SITK_REQUIRE(ld >= V * C && ld % 4 == 0, "gather_tokens_idx: ld=%d must be >= V*C=%d and a multiple of 4", ld, V * C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16)
    return run_gather<h16>(x_all, table_pv, tokens, B, n_vertices, C, P, V, ld, mean, stdv, s, sample_idx, targets_all, target_out, n_targets);
  if (dtype == SITK_F32)
    return run_gather<float>(x_all, table_pv, tokens, B, n_vertices, C, P, V, ld, mean, stdv, s, sample_idx, targets_all, target_out, n_targets);
  set_error("gather_tokens_idx: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_patchify)
extern "C" int sitk_patchify(const float* x_bcpv, void* tokens, int B, int C, int P, int V, int ld, int dtype,
                             sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_patchify, x_bcpv, tokens, B, C, P, V, ld, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(x_bcpv && tokens, "patchify: null pointer");
  SITK_REQUIRE(B > 0 && P > 0 && V > 0 && C > 0, "patchify: bad shape");
  SITK_REQUIRE(ld >= V * C && ld % 4 == 0, "patchify: ld=%d must be >= V*C=%d and a multiple of 4", ld, V * C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return run_patchify<h16>(x_bcpv, tokens, B, C, P, V, ld, s);
  if (dtype == SITK_F32) return run_patchify<float>(x_bcpv, tokens, B, C, P, V, ld, s);
  set_error("patchify: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_cast_rows)
extern "C" int sitk_cast_rows(const float* src, int lds_, void* dst, int ldd, int64_t rows, int cols, int dtype,
                              sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_cast_rows, src, lds_, dst, ldd, rows, cols, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(src && dst && rows > 0 && cols > 0, "cast_rows: bad arguments");
  SITK_REQUIRE(ldd >= cols && ldd % 4 == 0, "cast_rows: ldd=%d must be >= cols=%d and a multiple of 4", ldd, cols);
  SITK_REQUIRE(lds_ % 4 == 0, "cast_rows: source leading dim must be a multiple of 4");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return run_cast_rows<h16>(src, lds_, dst, ldd, rows, cols, s);
  if (dtype == SITK_F32) return run_cast_rows<float>(src, lds_, dst, ldd, rows, cols, s);
  set_error("cast_rows: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_stage_weight)
extern "C" int sitk_stage_weight(const float* w, int rows, int cols, void* w_c, int ldc, void* w_t, int ldt, int dtype,
                                 sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_stage_weight, w, rows, cols, w_c, ldc, w_t, ldt, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(w && rows > 0 && cols > 0, "stage_weight: bad arguments");
  SITK_REQUIRE(!w_c || (ldc >= cols && ldc % 8 == 0), "stage_weight: ldc=%d", ldc);
  SITK_REQUIRE(!w_t || (ldt >= rows && ldt % 8 == 0), "stage_weight: ldt=%d", ldt);
  SITK_REQUIRE(cols % 4 == 0, "stage_weight: cols %% 4 required");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return run_stage_weight<h16>(w, rows, cols, w_c, ldc, w_t, ldt, s);
  if (dtype == SITK_F32) return run_stage_weight<float>(w, rows, cols, w_c, ldc, w_t, ldt, s);
  set_error("stage_weight: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}
