// sitk small kernels: pool + head (models/sit.py:78-82), regression losses (tools/train.py:245-248),
// masked-patch-pretraining corruption and loss (models/mpp.py:85-112,132), fused optimizers
// (tools/train.py:228-243,291).  All fp32, HBM- or latency-bound.
#include <algorithm>

#include "common.h"

namespace sitk {

constexpr int HEAD_MAXD = 1024;
constexpr int HEAD_NV = HEAD_MAXD / 64;

// pooled row of sample b into registers: lane owns d = lane + 64 i
SITK_DEV void head_pool(const float* __restrict__ x, int b, int N, int D, int pool_mean, int lane, float (&v)[HEAD_NV]) {
  const float* xb = x + (size_t)b * N * D;
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i) v[i] = 0.f;
  if (pool_mean) {
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int i = 0; i < HEAD_NV; ++i) {
        const int d = lane + 64 * i;
        if (d < D) v[i] += xb[(size_t)n * D + d];
      }
    const float inv = 1.0f / (float)N;
#pragma unroll
    for (int i = 0; i < HEAD_NV; ++i) v[i] *= inv;
  } else {
#pragma unroll
    for (int i = 0; i < HEAD_NV; ++i) {
      const int d = lane + 64 * i;
      if (d < D) v[i] = xb[d];
    }
  }
}

SITK_DEV void head_stats(const float (&v)[HEAD_NV], int D, int lane, float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i) s += (lane + 64 * i < D) ? v[i] : 0.f;
  mu = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i)
    if (lane + 64 * i < D) { const float d = v[i] - mu; ss += d * d; }
  rs = rsqrtf(wave_sum(ss) / (float)D + 1e-5f);
}

// one wave per sample
__global__ __launch_bounds__(64) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ln_w,
                                                      const float* __restrict__ ln_b, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ logits, int N,
                                                      int D, int n_classes, int pool_mean) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float v[HEAD_NV];
  head_pool(x, b, N, D, pool_mean, lane, v);
  float mu, rs;
  head_stats(v, D, lane, mu, rs);
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i) {
    const int d = lane + 64 * i;
    v[i] = d < D ? (v[i] - mu) * rs * ln_w[d] + ln_b[d] : 0.f;
  }
  for (int c = 0; c < n_classes; ++c) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_NV; ++i) {
      const int d = lane + 64 * i;
      if (d < D) s += v[i] * w[(size_t)c * D + d];
    }
    s = wave_sum(s);
    if (lane == 0) logits[(size_t)b * n_classes + c] = s + bias[c];
  }
}

// Parameter gradients of the head and the loss are sums over the samples.  With a workspace every sample stores its
// terms in its own row [d_b (C) | d_w (C D) | d_ln_w (D) | d_ln_b (D) | loss (1)] and head_finalize_kernel adds the rows
// in sample order: bitwise reproducible.  Without one the terms go straight to the gradients as float atomics (their
// sum then depends on arrival order in the last bits).
SITK_DEV int head_ws_sums(int D, int C) { return C + C * D + 2 * D + 1; }      // columns the finalize kernel adds up
SITK_DEV int head_ws_width(int D, int C) { return head_ws_sums(D, C) + C; }   // + the sample's raw d loss / d logits
struct HeadWs {
  float *db, *dw, *dlw, *dlb, *loss, *raw;
  bool on;
  SITK_DEV HeadWs(float* ws, int b, int D, int C) {
    on = ws != nullptr;
    float* row = ws + (size_t)b * head_ws_width(D, C);
    db = row; dw = row + C; dlw = dw + (size_t)C * D; dlb = dlw + D; loss = dlb + D; raw = loss + 1;
  }
  SITK_DEV void add(float* slot, float* grad, float v) const {
    if (on) *slot = v;
    else unsafeAtomicAdd(grad, v);
  }
};

__global__ __launch_bounds__(64) void head_finalize_kernel(const float* __restrict__ ws, int B, int D, int C,
                                                            float* __restrict__ d_b, float* __restrict__ d_w,
                                                            float* __restrict__ d_ln_w, float* __restrict__ d_ln_b,
                                                            float* __restrict__ loss) {
  const int W = head_ws_width(D, C), WS = head_ws_sums(D, C), c = blockIdx.x * 64 + threadIdx.x;
  if (c >= WS) return;
  float s = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {                             // 16 independent loads in flight, added in sample order
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = b0 + i < B ? ws[(size_t)(b0 + i) * W + c] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  if (c < C) d_b[c] += s;
  else if (c < C + C * D) d_w[c - C] += s;
  else if (c < C + C * D + D) d_ln_w[c - C - C * D] += s;
  else if (c < WS - 1) d_ln_b[c - C - C * D - D] += s;
  else if (loss) *loss += s;
}

// one wave per sample: parameter grads and the pooled-row gradient written to dx[b, 0, :]
__global__ __launch_bounds__(64) void head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ln_w,
                                                      const float* __restrict__ ln_b, const float* __restrict__ w,
                                                      const float* __restrict__ dlogits, float* __restrict__ dx,
                                                      float* __restrict__ d_ln_w, float* __restrict__ d_ln_b,
                                                      float* __restrict__ d_w, float* __restrict__ d_b, int N, int D,
                                                      int n_classes, int pool_mean, float* __restrict__ ws) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const HeadWs hw(ws, b, D, n_classes);
  float v[HEAD_NV], dh[HEAD_NV];
  head_pool(x, b, N, D, pool_mean, lane, v);
  float mu, rs;
  head_stats(v, D, lane, mu, rs);
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i) { v[i] = (v[i] - mu) * rs; dh[i] = 0.f; }  // xhat
  for (int c = 0; c < n_classes; ++c) {
    const float dl = dlogits[(size_t)b * n_classes + c];
    if (lane == 0) hw.add(hw.db + c, d_b + c, dl);
#pragma unroll
    for (int i = 0; i < HEAD_NV; ++i) {
      const int d = lane + 64 * i;
      if (d < D) {
        dh[i] += dl * w[(size_t)c * D + d];
        hw.add(hw.dw + (size_t)c * D + d, d_w + (size_t)c * D + d, dl * (v[i] * ln_w[d] + ln_b[d]));
      }
    }
  }
  if (lane == 0 && ws) *hw.loss = 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i) {
    const int d = lane + 64 * i;
    if (d < D) {
      hw.add(hw.dlw + d, d_ln_w + d, dh[i] * v[i]);
      hw.add(hw.dlb + d, d_ln_b + d, dh[i]);
      dh[i] *= ln_w[d];
      s1 += dh[i];
      s2 += dh[i] * v[i];
    }
  }
  s1 = wave_sum(s1) / (float)D;
  s2 = wave_sum(s2) / (float)D;
  const float post = pool_mean ? 1.0f / (float)N : 1.0f;
#pragma unroll
  for (int i = 0; i < HEAD_NV; ++i) {
    const int d = lane + 64 * i;
    if (d < D) dx[(size_t)b * N * D + d] = rs * (dh[i] - s1 - v[i] * s2) * post;
  }
}

// rows n >= 1 of every sample: copy of row 0 (mean pooling) or zeros (cls pooling)
__global__ __launch_bounds__(256) void head_spread_kernel(float* __restrict__ dx, int64_t B, int N, int D, int pool_mean) {
  const int nvec = D >> 2;
  const int64_t total = B * (int64_t)(N - 1) * nvec;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % nvec);
    const int64_t r = i / nvec;
    const int64_t b = r / (N - 1);
    const int n = (int)(r % (N - 1)) + 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (pool_mean) v = load4(dx + (size_t)b * N * D + 4 * c);
    store4(dx + ((size_t)b * N + n) * D + 4 * c, v);
  }
}

// Pool + head + loss + their backward in ONE launch (the regression step of tools/train.py:245-248,288-290 between the
// encoder's forward and backward): one workgroup of 4 waves per sample.  Wave 0 does what head_fwd, loss and head_bwd do
// for its sample -- the loss gradient of a sample depends on no other sample, only the scalar loss is a sum (one atomic
// per sample) -- then all four waves write the sample's rows 1..N-1 of dx (zeros, or copies of row 0 for mean pooling).
// Replaces 4 dependent launches (head_fwd, loss, head_bwd, head_spread) of ~5-9 us each.
// MODE 0: everything in one pass, gradients unscaled.  MODE 1 + MODE 2 (two launches) = the same with every gradient the
// step produces multiplied by a power of two S chosen from THIS batch (f16 compute mode: the narrow exponent of the
// 16-bit gradient operands): pass 1 = forward, loss and the raw d loss / d logits of every sample (stored in the workspace
// rows); pass 2 = backward, where every workgroup first takes the batch maximum of |d loss / d logits| from those rows --
// the same B x C values for everybody, no atomics --, sets S = 2^k with max * S in [64, 128) and scales its sample's loss
// gradient by it; block (0, 0) publishes {S, 1 / S} for the optimizer.  The loss itself is never scaled.
template <int MODE>
__global__ __launch_bounds__(256) void head_loss_fused_kernel(const float* __restrict__ x, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ target,
                                                              float* __restrict__ logits, float* __restrict__ loss,
                                                              float* __restrict__ dx, float* __restrict__ d_ln_w,
                                                              float* __restrict__ d_ln_b, float* __restrict__ d_w,
                                                              float* __restrict__ d_b, int B, int N, int D, int n_classes,
                                                              int pool_mean, int l1, float* __restrict__ ws,
                                                              float* __restrict__ gscale) {
  // grid (B, S): slice y of sample b writes its share of the rows 1..N-1; slice 0 also does the head itself.  Mean pooling
  // copies row 0, which slice 0 produces: S = 1 then (chosen by the launcher).
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0 && blockIdx.y == 0) {
    const HeadWs hw(ws, b, D, n_classes);
    float v[HEAD_NV], dh[HEAD_NV];
    head_pool(x, b, N, D, pool_mean, lane, v);
    float mu, rs;
    head_stats(v, D, lane, mu, rs);
#pragma unroll
    for (int i = 0; i < HEAD_NV; ++i) { v[i] = (v[i] - mu) * rs; dh[i] = 0.f; }  // xhat
    const float inv = 1.0f / (float)(B * n_classes);
    float lsum = 0.f, gs = 1.0f;
    if constexpr (MODE == 2) {
      float m = 0.f;
      const int W = head_ws_width(D, n_classes), raw0 = head_ws_sums(D, n_classes);
      for (int i = lane; i < B * n_classes; i += 64) m = fmaxf(m, fabsf(ws[(size_t)(i / n_classes) * W + raw0 + i % n_classes]));
      m = wave_max(m);
      const int ef = (int)((__builtin_bit_cast(uint32_t, m) >> 23) & 0xffu);     // m = f 2^(ef - 126), f in [0.5, 1)
      if (ef > 0 && ef < 255) gs = __builtin_bit_cast(float, (uint32_t)min(max(260 - ef, 1), 254) << 23);   // 2^(7 - (ef - 126))
      if (b == 0 && lane == 0) { gscale[0] = gs; gscale[1] = 1.0f / gs; }
    }
    for (int c = 0; c < n_classes; ++c) {
      float dl;
      if constexpr (MODE == 2) {
        dl = hw.raw[c] * gs;
        if (lane == 0) hw.add(hw.db + c, d_b + c, dl);
      } else {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_NV; ++i) {
          const int d = lane + 64 * i;
          if (d < D) sacc += (v[i] * ln_w[d] + ln_b[d]) * w[(size_t)c * D + d];
        }
        const float logit = wave_sum(sacc) + bias[c];
        const float df = logit - target[(size_t)b * n_classes + c];
        dl = l1 ? (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv : 2.f * df * inv;
        lsum += l1 ? fabsf(df) : df * df;
        if (lane == 0) {
          logits[(size_t)b * n_classes + c] = logit;
          if constexpr (MODE == 1) hw.raw[c] = dl;
          else hw.add(hw.db + c, d_b + c, dl);
        }
      }
      if constexpr (MODE == 1) continue;
#pragma unroll
      for (int i = 0; i < HEAD_NV; ++i) {
        const int d = lane + 64 * i;
        if (d < D) {
          dh[i] += dl * w[(size_t)c * D + d];
          hw.add(hw.dw + (size_t)c * D + d, d_w + (size_t)c * D + d, dl * (v[i] * ln_w[d] + ln_b[d]));
        }
      }
    }
    if constexpr (MODE != 2) { if (lane == 0) hw.add(hw.loss, loss, lsum * inv); }
    if constexpr (MODE != 1) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < HEAD_NV; ++i) {
        const int d = lane + 64 * i;
        if (d < D) {
          hw.add(hw.dlw + d, d_ln_w + d, dh[i] * v[i]);
          hw.add(hw.dlb + d, d_ln_b + d, dh[i]);
          dh[i] *= ln_w[d];
          s1 += dh[i];
          s2 += dh[i] * v[i];
        }
      }
      s1 = wave_sum(s1) / (float)D;
      s2 = wave_sum(s2) / (float)D;
      const float post = pool_mean ? 1.0f / (float)N : 1.0f;
#pragma unroll
      for (int i = 0; i < HEAD_NV; ++i) {
        const int d = lane + 64 * i;
        if (d < D) dx[(size_t)b * N * D + d] = rs * (dh[i] - s1 - v[i] * s2) * post;
      }
    }
  }
  if constexpr (MODE == 1) return;
  if (pool_mean) __syncthreads();                        // row 0 is copied below (same workgroup: visible after the barrier)
  const int nvec = D >> 2;
  float* dxb = dx + (size_t)b * N * D;
  const int per = ((N - 1) * nvec + (int)gridDim.y - 1) / (int)gridDim.y;
  const int i_end = min((N - 1) * nvec, ((int)blockIdx.y + 1) * per);
  for (int i = (int)blockIdx.y * per + threadIdx.x; i < i_end; i += 256) {
    const int c = i % nvec, n = i / nvec + 1;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (pool_mean) val = load4(dxb + 4 * c);
    store4(dxb + (size_t)n * D + 4 * c, val);
  }
}

// x[b, 0, :] = cls + pos[0, :]   (models/sit.py:70-73: cls token row of the residual stream)
__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, int B, int N, int D) {
  const int nvec = D >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * nvec; i += gridDim.x * 256) {
    const int b = i / nvec, c = i % nvec;
    store4(x + (size_t)b * N * D + 4 * c, load4(cls + 4 * c) + load4(pos + 4 * c));
  }
}

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                   float* __restrict__ loss, float* __restrict__ dpred, int n, int l1) {
  __shared__ float red[4];
  float s = 0.f;
  const float inv = 1.0f / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = pred[i] - target[i];
    if (l1) {
      s += fabsf(d);
      dpred[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv;
    } else {
      s += d * d;
      dpred[i] = 2.f * d * inv;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// ---- masked patch pre-training ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mpp_corrupt_kernel(const float* __restrict__ tokens, const uint8_t* __restrict__ masked,
                                                          const uint8_t* __restrict__ swap_draw,
                                                          const int32_t* __restrict__ random_patches,
                                                          const uint8_t* __restrict__ replace_draw,
                                                          const float* __restrict__ mask_token, T* __restrict__ out,
                                                          int64_t rows, int P, int K, int ld) {
  const int slots = ld >> 2;
  for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
    const bool m = masked[row] != 0;
    const bool rep = m && replace_draw[row] != 0;
    const bool swp = m && swap_draw != nullptr && swap_draw[row] != 0;
    const float* src = tokens + (size_t)row * K;
    if (rep) src = mask_token;
    else if (swp) src = tokens + ((size_t)(row / P) * P + random_patches[row]) * K;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < slots; v += gridDim.x * 256) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (4 * v < K) val = load4(src + 4 * v);
      store4(out + (size_t)row * ld + 4 * v, val);
    }
  }
}

// ---- the four random tensors of models/mpp.py:25-43,85-112 drawn ON THE DEVICE (engine path) -------------------------
// Philox4x32-10 (Salmon et al., SC'11) keyed by the 64-bit seed; counter = (draw index, element, stream id, sample).
// state[0] = seed, state[1] = number of draws made so far: read by every block here, advanced by the kernel that
// consumes the flags (sitk_mpp_gather_corrupt), i.e. after this launch has completed -- so a captured hipGraph draws new
// masks at every replay.
SITK_DEV void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
SITK_DEV void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
SITK_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0, 1), 24 bits

// One workgroup per sample.  corrupted_sequence: EXACTLY n_mask patches per sample, the n_mask largest of P uniform
// scores (models/mpp.py:25-33, get_mask_from_prob: rand -> topk -> scatter_; ties by index: the patch with the smaller
// index wins); swap draw U < p_swap, random_patches uniform in [0, P), replace draw U < p_replace (models/mpp.py:36-43,
// 95-110).  The scores are 24-bit integers (u01 keeps 24 bits), so the n_mask-th largest is found by a RADIX SELECT --
// three passes of a 256-bucket histogram in LDS over the keys that still match the prefix, one wave locating the bucket
// that holds the n_mask-th key -- instead of ranking every patch against every other (rounds 1 - 2: O(P^2), 206 us at
// P = 1280 with every one of P / 64 blocks per sample re-drawing all P scores; now O(P), one block per sample).  Same
// flags as the ranking produced, bit for bit.
constexpr int MPP_MAX_P = 2048;
__global__ __launch_bounds__(256) void mpp_draw_kernel(const uint64_t* __restrict__ state, uint8_t* __restrict__ masked,
                                                       uint8_t* __restrict__ swap_draw, int32_t* __restrict__ random_patches,
                                                       uint8_t* __restrict__ replace_draw, uint8_t* __restrict__ replaced_full,
                                                       int P, int n_mask, float p_swap, float p_replace) {
  __shared__ uint32_t key[MPP_MAX_P];
  __shared__ int hist[256];
  __shared__ int sel[2];                                      // chosen bucket, keys still to take from it and below
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint64_t seed = state[0], draw = state[1];
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint8_t rep[MPP_MAX_P / 256];
#pragma unroll
  for (int t = 0; t < MPP_MAX_P / 256; ++t) {
    const int i = tid + 256 * t;
    rep[t] = 0;
    if (i < P) {
      uint32_t c[4] = {(uint32_t)draw, (uint32_t)i, (uint32_t)(draw >> 32), (uint32_t)b};
      philox4x32(c, k0, k1);
      key[i] = c[0] >> 8;                                     // u01(c[0]) = key / 2^24: the same order
      const size_t row = (size_t)b * P + i;
      if (swap_draw) { swap_draw[row] = u01(c[1]) < p_swap; random_patches[row] = (int32_t)(((uint64_t)c[2] * (uint32_t)P) >> 32); }
      rep[t] = u01(c[3]) < p_replace;
      replace_draw[row] = rep[t];
    }
  }
  // the key T of rank n_mask - 1 (0-based, descending) and r = how many keys equal to T are taken (those of lowest index)
  uint32_t prefix = 0, known = 0;
  int need = n_mask;
  if (n_mask > 0 && n_mask < P) {
    for (int pass = 2; pass >= 0; --pass) {
      hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < P; i += 256)
        if (((key[i] ^ prefix) & known) == 0) atomicAdd(&hist[(key[i] >> (8 * pass)) & 255], 1);
      __syncthreads();
      if (tid < 64) {                                         // lane l owns buckets 255 - 4l .. 252 - 4l (descending order)
        int h[4], s = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = hist[255 - 4 * tid - e]; s += h[e]; }
        int incl = s;                                         // inclusive prefix over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (tid >= d) incl += o;
        }
        int above = incl - s;                                 // keys in buckets above this lane's
        if (above < need && need <= incl) {                   // exactly one lane: the bucket with the need-th key is here
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (above < need && need <= above + h[e]) { sel[0] = 255 - 4 * tid - e; sel[1] = need - above; }
            above += h[e];
          }
        }
      }
      __syncthreads();
      prefix |= (uint32_t)sel[0] << (8 * pass);
      known |= 255u << (8 * pass);
      need = sel[1];
      __syncthreads();
    }
  }
  const uint32_t T = prefix;
#pragma unroll
  for (int t = 0; t < MPP_MAX_P / 256; ++t) {
    const int i = tid + 256 * t;
    if (i < P) {
      uint8_t m;
      if (n_mask <= 0) m = 0;
      else if (n_mask >= P) m = 1;
      else if (key[i] != T) m = key[i] > T;
      else {                                                  // a tie on the threshold (rare): the `need` lowest indices win
        int before = 0;
        for (int j = 0; j < i; ++j) before += key[j] == T;
        m = before < need;
      }
      const size_t row = (size_t)b * P + i;
      masked[row] = m;
      replaced_full[(size_t)b * (P + 1) + 1 + i] = m && rep[t];
    }
  }
  if (tid == 0) replaced_full[(size_t)b * (P + 1)] = 0;
}

// Patch gather (tools/preprocessing.py:74-84 + Rearrange, as gather_tokens_kernel) fused with the corruption of
// models/mpp.py:85-112: one pass writes the clean fp32 tokens (the regression target of models/mpp.py:132) AND the corrupted
// compute-dtype tokens.  A swapped patch reads the surface through the table row of its random partner (same sample),
// a replaced one takes mask_token (replacement wins).  Thread (0, 0, 0) advances the draw counter of mpp_draw_kernel.
template <typename T>
__global__ __launch_bounds__(256) void mpp_gather_corrupt_kernel(const float* __restrict__ x, const uint16_t* __restrict__ table,
                                                                 const int32_t* __restrict__ sample_idx, const float* __restrict__ mean,
                                                                 const float* __restrict__ stdv, const uint8_t* __restrict__ masked,
                                                                 const uint8_t* __restrict__ swap_draw,
                                                                 const int32_t* __restrict__ random_patches,
                                                                 const uint8_t* __restrict__ replace_draw,
                                                                 const float* __restrict__ mask_token, float* __restrict__ clean,
                                                                 T* __restrict__ corrupted, uint64_t* __restrict__ state, int64_t rows,
                                                                 int n_vertices, int P, int V, int K, int ld) {
  if (state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) state[1] += 1;
  const int slots = ld >> 2;
  const bool norm = mean != nullptr;
  const f32x4 mu = norm ? load4(mean) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 sd = norm ? load4(stdv) : f32x4{1.f, 1.f, 1.f, 1.f};
  for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
    const int p = (int)(row % P);
    const int64_t b = sample_idx ? (int64_t)sample_idx[row / P] : row / P;
    const bool m = masked[row] != 0;
    const bool rep = m && replace_draw[row] != 0;
    const bool swp = m && !rep && swap_draw != nullptr && swap_draw[row] != 0;
    const int p2 = swp ? random_patches[row] : p;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < slots; v += gridDim.x * 256) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f}, cor = {0.f, 0.f, 0.f, 0.f};
      if (v < V) {
        val = *reinterpret_cast<const f32x4*>(x + ((size_t)b * n_vertices + table[(size_t)p * V + v]) * 4);
        if (norm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = (val[e] - mu[e]) / sd[e];
        }
        cor = val;
        if (rep) {
          cor = load4(mask_token + 4 * v);
        } else if (swp) {
          cor = *reinterpret_cast<const f32x4*>(x + ((size_t)b * n_vertices + table[(size_t)p2 * V + v]) * 4);
          if (norm) {
#pragma unroll
            for (int e = 0; e < 4; ++e) cor[e] = (cor[e] - mu[e]) / sd[e];
          }
        }
        store4(clean + (size_t)row * K + 4 * v, val);
      }
      store4(corrupted + (size_t)row * ld + 4 * v, cor);
    }
  }
}

__global__ __launch_bounds__(256) void mpp_loss_kernel(const float* __restrict__ out, const float* __restrict__ tokens,
                                                       const uint8_t* __restrict__ masked, float* __restrict__ loss,
                                                       float* __restrict__ dout, int64_t rows, int K, float inv_count) {
  __shared__ float red[4];
  const int nvec = K >> 2;
  float s = 0.f;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const bool m = masked[row] != 0;
    for (int c = threadIdx.x; c < nvec; c += 256) {
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (m) {
        const f32x4 d = load4(out + (size_t)row * K + 4 * c) - load4(tokens + (size_t)row * K + 4 * c);
        s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        g = d * (2.f * inv_count);
      }
      store4(dout + (size_t)row * K + 4 * c, g);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_count);
}

// the same with leading dimensions and the gradient in the compute dtype (engine path: batch_out lives in a row-padded buffer
// written by the weight-resident GEMM, the gradient feeds bf16 GEMMs): whole rows, 16-byte accesses
template <typename T>
__global__ __launch_bounds__(256) void mpp_loss_ld_kernel(const float* __restrict__ out, int ldo, const float* __restrict__ tokens,
                                                          int ldt, const uint8_t* __restrict__ masked, float* __restrict__ loss,
                                                          T* __restrict__ dout, int lddo, int64_t rows, int K, float inv_count,
                                                          float grad_scale) {
  __shared__ float red[4];
  const int nvec = K >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  // one row per wave at a time (a row is 2-3 float4 per lane: every load of the row is in flight before the first use);
  // few workgroups, many rows each: the launch ends in ONE atomic per workgroup on the same address
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const bool m = masked[row] != 0;
    const float* po = out + (size_t)row * ldo;
    const float* pt = tokens + (size_t)row * ldt;
    T* pd = dout + (size_t)row * lddo;
    for (int c0 = 0; c0 < nvec; c0 += 256) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + lane + 64 * j;
        a[j] = b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m && c < nvec) { a[j] = load4(po + 4 * c); b[j] = load4(pt + 4 * c); }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + lane + 64 * j;
        if (c < nvec) {
          const f32x4 d = a[j] - b[j];                         // 0 for unmasked rows
          s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
          store4(pd + 4 * c, d * (2.f * inv_count * grad_scale));
        }
      }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_count);
}

// ---- dropout / GELU as stand-alone elementwise kernels: the UNFUSED encoder path taken when the reference's `dropout` ctor
// argument (models/sit.py:36,57: Attention.to_out.1, FeedForward.net.2 / net.4 of vit_pytorch) is > 0 in training.  Every
// reference configuration sets 0.0 (config/SiT/*/hparams.yml:46), so this path is for API completeness, not speed.
// y = res + x * keep / (1 - p), keep ~ Bernoulli(1 - p) from Philox4x32-10 keyed by the seed; counter = (draw index, vector
// index): 4 elements per Philox call.  The mask is stored (1 byte per element) for the backward pass.
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          float* __restrict__ y, uint8_t* __restrict__ mask, int64_t n, float p,
                                                          const uint64_t* __restrict__ state) {
  const uint64_t seed = state[0], draw = state[1];
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const float scale = 1.0f / (1.0f - p);
  const int64_t nvec = (n + 3) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    uint32_t c[4] = {(uint32_t)draw, (uint32_t)i, (uint32_t)(draw >> 32) ^ 0x5bd1e995u, (uint32_t)(i >> 32)};
    philox4x32(c, k0, k1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = 4 * i + e;
      if (j < n) {
        const bool keep = u01(c[e]) >= p;
        mask[j] = keep;
        y[j] = (res ? res[j] : 0.f) + (keep ? x[j] * scale : 0.f);
      }
    }
  }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ dx, int64_t n, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = mask[i] ? dy[i] * scale : 0.f;
}
__global__ __launch_bounds__(256) void advance_draw_kernel(uint64_t* state) { state[1] += 1; }
// exact-erf GELU (nn.GELU(), utils/utils.py:30's net.1) and its derivative, fp32
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ u, float* __restrict__ g, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] = gelu_erf(u[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ u,
                                                       float* __restrict__ du, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) du[i] = dg[i] * gelu_erf_grad(u[i]);
}

// ---- optimizers ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, float lr, float momentum, float wd, int nesterov, float gscale) {
  const int64_t nvec = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    f32x4 pv = load4(p + 4 * i);
    f32x4 gv = load4(g + 4 * i) * gscale + pv * wd;
    if (momentum != 0.f) {
      f32x4 bv = load4(buf + 4 * i) * momentum + gv;
      store4(buf + 4 * i, bv);
      gv = nesterov ? gv + bv * momentum : bv;
    }
    store4(p + 4 * i, pv - gv * lr);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    float gv = g[i] * gscale + p[i] * wd;
    if (momentum != 0.f) {
      const float bv = buf[i] * momentum + gv;
      buf[i] = bv;
      gv = nesterov ? gv + bv * momentum : bv;
    }
    p[i] -= lr * gv;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, int decoupled, float bc1, float bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pv = p[i], gv = g[i] * gscale;
    if (decoupled) pv *= (1.f - lr * wd);
    else gv += wd * pv;
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * (mv / denom);
  }
}

// ---- optimizers with device-resident hyper-parameters (hipGraph-safe) ----------------------------
// state = 4 doubles {lr, beta1^t, beta2^t, t}.  The learning rate and Adam's bias corrections are READ FROM MEMORY, so a
// captured launch sees set_lr() and the advancing step count.  zero = 1: the consumed gradient is overwritten with zeros
// (zero_grad of the next step folded into this pass), and so are `n_extra` accumulator floats behind the gradients;
// the one at extra index `keep_idx` (the step's loss) is first copied to keep_dst.
__global__ __launch_bounds__(256) void sgd_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                      int64_t n, const double* __restrict__ state, float momentum, float wd,
                                                      int nesterov, float gscale, int zero, int64_t n_extra, int64_t keep_idx,
                                                      float* __restrict__ keep_dst, const float* __restrict__ inv_gscale,
                                                      int* __restrict__ nonfinite, float* __restrict__ g2,
                                                      const float* __restrict__ inv_gscale2, float keep_scale) {
  // g2 (ABI 12): a SECOND gradient buffer of the same layout (the other half of a batch that ran as two concurrent half-batch
  // steps, each with its own loss scale in f16 mode): the pass consumes g * s1 + g2 * s2 and clears both
  const float gscale2 = g2 ? gscale * (inv_gscale2 ? inv_gscale2[0] : 1.f) : 0.f;
  if (inv_gscale) gscale *= inv_gscale[0];               // the step's loss scale (f16 compute mode), undone here
  const float lr = (float)state[0];
  const int64_t nvec = n >> 2;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // nonfinite != NULL (the engine passes it in the loss-scaled f16 mode only): a gradient element that is not finite -- an f16
  // intermediate that overflowed behind the loss scale -- must not reach the parameters or the momentum: that ELEMENT is skipped
  // (zeroed like every consumed gradient) and counted.  NULL: the reference's behaviour, tools/train.py:291 -- no guard, a
  // diverged run shows NaN parameters.
  const bool guard = nonfinite != nullptr;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const f32x4 pv = load4(p + 4 * i);
    f32x4 graw = load4(g + 4 * i) * gscale;
    if (zero) store4(g + 4 * i, z);
    if (g2) {
      graw = graw + load4(g2 + 4 * i) * gscale2;
      if (zero) store4(g2 + 4 * i, z);
    }
    f32x4 gv = graw + pv * wd;
    f32x4 bv = z, bold = z;
    if (momentum != 0.f) {
      bold = load4(buf + 4 * i);
      bv = bold * momentum + gv;
      gv = nesterov ? gv + bv * momentum : bv;
    }
    f32x4 pn = pv - gv * lr;
    if (guard) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(fabsf(graw[e]) <= 3.0e38f)) { ++bad; pn[e] = pv[e]; bv[e] = bold[e]; }
    }
    if (momentum != 0.f) store4(buf + 4 * i, bv);
    store4(p + 4 * i, pn);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    float graw = g[i] * gscale;
    if (zero) g[i] = 0.f;
    if (g2) {
      graw += g2[i] * gscale2;
      if (zero) g2[i] = 0.f;
    }
    if (!guard || fabsf(graw) <= 3.0e38f) {
      float gv = graw + p[i] * wd;
      if (momentum != 0.f) {
        const float bv = buf[i] * momentum + gv;
        buf[i] = bv;
        gv = nesterov ? gv + bv * momentum : bv;
      }
      p[i] -= lr * gv;
    } else {
      ++bad;
    }
  }
  if (bad && nonfinite) atomicAdd(nonfinite, bad);
  if (zero)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_extra; i += (int64_t)gridDim.x * 256) {
      if (i == keep_idx && keep_dst) keep_dst[0] = (g[n + i] + (g2 ? g2[n + i] : 0.f)) * keep_scale;
      g[n + i] = 0.f;
      if (g2) g2[n + i] = 0.f;
    }
}

__global__ void adam_advance_kernel(double* state, double b1, double b2) {
  state[1] *= b1;
  state[2] *= b2;
  state[3] += 1.0;
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, const double* __restrict__ state,
                                                       float b1, float b2, float eps, float wd, int decoupled, float gscale,
                                                       int zero, int64_t n_extra, int64_t keep_idx, float* __restrict__ keep_dst,
                                                       const float* __restrict__ inv_gscale, int* __restrict__ nonfinite,
                                                       float* __restrict__ g2, const float* __restrict__ inv_gscale2,
                                                       float keep_scale) {
  const float gscale2 = g2 ? gscale * (inv_gscale2 ? inv_gscale2[0] : 1.f) : 0.f;      // (see sgd_dev_kernel)
  if (inv_gscale) gscale *= inv_gscale[0];               // the step's loss scale (f16 compute mode), undone here
  const float lr = (float)state[0];
  const float bc1 = (float)(1.0 - state[1]), bc2_sqrt = (float)sqrt(1.0 - state[2]);
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pv = p[i], gv = g[i] * gscale;
    if (g2) {
      gv += g2[i] * gscale2;
      if (zero) g2[i] = 0.f;
    }
    if (nonfinite && !(fabsf(gv) <= 3.0e38f)) {          // not finite, guarded mode: skipped and counted (see sgd_dev_kernel)
      ++bad;
      if (zero) g[i] = 0.f;
      continue;
    }
    if (decoupled) pv *= (1.f - lr * wd);
    else gv += wd * pv;
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * (mv / denom);
    if (zero) g[i] = 0.f;
  }
  if (bad && nonfinite) atomicAdd(nonfinite, bad);
  if (zero)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_extra; i += (int64_t)gridDim.x * 256) {
      if (i == keep_idx && keep_dst) keep_dst[0] = (g[n + i] + (g2 ? g2[n + i] : 0.f)) * keep_scale;
      g[n + i] = 0.f;
      if (g2) g2[n + i] = 0.f;
    }
}

static int grid_for(int64_t work, int per_block, int cap) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(work, per_block), cap));
}

}  // namespace sitk

extern "C" size_t sitk_head_ws_floats(int B, int D, int n_classes) {
  return B > 0 && D > 0 && n_classes > 0 ? (size_t)B * (2 * (size_t)n_classes + (size_t)n_classes * D + 2 * (size_t)D + 1) : 0;
}

extern "C" int sitk_head_fwd(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                             float* logits, int B, int N, int D, int n_classes, int pool_mean, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(x && ln_w && ln_b && w && b && logits, "head_fwd: null pointer");
  SITK_REQUIRE(B > 0 && N > 0 && D > 0 && D <= HEAD_MAXD && n_classes > 0, "head_fwd: bad shape (D <= %d)", HEAD_MAXD);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), x, ln_w, ln_b, w, b,
                     logits, N, D, n_classes, pool_mean);
  return check_launch("head_fwd");
}

extern "C" int sitk_head_bwd(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* dlogits,
                             float* dx, float* d_ln_w, float* d_ln_b, float* d_w, float* d_b, int B, int N, int D,
                             int n_classes, int pool_mean, float* ws, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(x && ln_w && ln_b && w && dlogits && dx && d_ln_w && d_ln_b && d_w && d_b, "head_bwd: null pointer");
  SITK_REQUIRE(B > 0 && N > 0 && D > 0 && D <= HEAD_MAXD && D % 4 == 0 && n_classes > 0, "head_bwd: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(64), 0, s, x, ln_w, ln_b, w, dlogits, dx, d_ln_w, d_ln_b, d_w, d_b, N,
                     D, n_classes, pool_mean, ws);
  SITK_LAUNCH_CHECK("head_bwd");
  if (ws) {
    hipLaunchKernelGGL(head_finalize_kernel, dim3(cdiv((int)sitk_head_ws_floats(1, D, n_classes), 64)), dim3(64), 0, s, ws, B, D,
                       n_classes, d_b, d_w, d_ln_w, d_ln_b, (float*)nullptr);
    SITK_LAUNCH_CHECK("head_finalize");
  }
  if (N > 1) {
    const int64_t total = (int64_t)B * (N - 1) * (D / 4);
    hipLaunchKernelGGL(head_spread_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, dx, (int64_t)B, N, D, pool_mean);
    SITK_LAUNCH_CHECK("head_spread");
  }
  return SITK_OK;
}

extern "C" int sitk_embed_cls_rows(float* x, const float* cls_token, const float* pos, int B, int N, int D,
                                   sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(x && cls_token && pos && B > 0 && N > 0 && D > 0 && D % 4 == 0, "embed_cls_rows: bad arguments");
  hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for((int64_t)B * D / 4, 256, 1024)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, cls_token, pos, B, N, D);
  return check_launch("embed_cls_rows");
}

static int head_loss_launch(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                            const float* target, float* logits, float* loss, float* dx, float* d_ln_w, float* d_ln_b,
                            float* d_w, float* d_b, int B, int N, int D, int n_classes, int pool_mean, int l1, float* ws,
                            float* grad_scale, bool finalize, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(x && ln_w && ln_b && w && b && target && logits && loss && dx && d_ln_w && d_ln_b && d_w && d_b,
               "head_loss_fwd_bwd: null pointer");
  SITK_REQUIRE(B > 0 && N > 0 && D > 0 && D <= HEAD_MAXD && D % 4 == 0 && n_classes > 0, "head_loss_fwd_bwd: bad shape");
  const int slices = pool_mean ? 1 : std::max(1, std::min(8, 512 / B));      // fill the chip with the row writes
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SITK_REQUIRE(!grad_scale || ws, "head_loss_fwd_bwd: the scaled form needs the workspace");
  if (grad_scale) {
    hipLaunchKernelGGL(head_loss_fused_kernel<1>, dim3(B, 1), dim3(64), 0, s, x, ln_w, ln_b, w, b, target, logits, loss, dx, d_ln_w,
                       d_ln_b, d_w, d_b, B, N, D, n_classes, pool_mean, l1, ws, grad_scale);
    SITK_LAUNCH_CHECK("head_loss_fwd");
    hipLaunchKernelGGL(head_loss_fused_kernel<2>, dim3(B, slices), dim3(256), 0, s, x, ln_w, ln_b, w, b, target, logits, loss, dx,
                       d_ln_w, d_ln_b, d_w, d_b, B, N, D, n_classes, pool_mean, l1, ws, grad_scale);
  } else {
    hipLaunchKernelGGL(head_loss_fused_kernel<0>, dim3(B, slices), dim3(256), 0, s, x, ln_w, ln_b, w, b, target, logits, loss, dx,
                       d_ln_w, d_ln_b, d_w, d_b, B, N, D, n_classes, pool_mean, l1, ws, grad_scale);
  }
  SITK_LAUNCH_CHECK("head_loss_fwd_bwd");
  if (ws && finalize) {
    hipLaunchKernelGGL(head_finalize_kernel, dim3(cdiv((int)sitk_head_ws_floats(1, D, n_classes), 64)), dim3(64), 0, s, ws, B, D,
                       n_classes, d_b, d_w, d_ln_w, d_ln_b, loss);
    SITK_LAUNCH_CHECK("head_finalize");
  }
  return SITK_OK;
}

extern "C" int sitk_head_loss_fwd_bwd(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                                      const float* target, float* logits, float* loss, float* dx, float* d_ln_w,
                                      float* d_ln_b, float* d_w, float* d_b, int B, int N, int D, int n_classes,
                                      int pool_mean, int l1, float* ws, float* grad_scale, sitk_stream_t stream) {
  return head_loss_launch(x, ln_w, ln_b, w, b, target, logits, loss, dx, d_ln_w, d_ln_b, d_w, d_b, B, N, D, n_classes, pool_mean, l1,
                          ws, grad_scale, true, stream);
}

// The same without the reduction of the workspace rows: dx (all the backward chain waits for) is complete, the head's parameter
// gradients and the loss are not until sitk_head_finalize has run -- on any stream ordered behind this call, e.g. beside the chain.
extern "C" int sitk_head_loss_fwd_bwd_deferred(const float* x, const float* ln_w, const float* ln_b, const float* w, const float* b,
                                               const float* target, float* logits, float* dx, int B, int N, int D, int n_classes,
                                               int pool_mean, int l1, float* ws, float* grad_scale, sitk_stream_t stream) {
  SITK_REQUIRE(ws, "head_loss_fwd_bwd_deferred: needs the workspace");
  float* unused = ws;        // (never written: every gradient term goes to the workspace rows)
  return head_loss_launch(x, ln_w, ln_b, w, b, target, logits, unused, dx, unused, unused, unused, unused, B, N, D, n_classes,
                          pool_mean, l1, ws, grad_scale, false, stream);
}

extern "C" int sitk_head_finalize(const float* ws, int B, int D, int n_classes, float* d_ln_w, float* d_ln_b, float* d_w,
                                  float* d_b, float* loss, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(ws && d_ln_w && d_ln_b && d_w && d_b && B > 0 && D > 0 && n_classes > 0, "head_finalize: bad arguments");
  hipLaunchKernelGGL(head_finalize_kernel, dim3(cdiv((int)sitk_head_ws_floats(1, D, n_classes), 64)), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), ws, B, D, n_classes, d_b, d_w, d_ln_w, d_ln_b, loss);
  return check_launch("head_finalize");
}

extern "C" int sitk_loss_fwd_bwd(const float* pred, const float* target, float* loss, float* dpred, int n, int l1,
                                 sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(pred && target && loss && dpred && n > 0, "loss_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred, target, loss, dpred, n, l1);
  return check_launch("loss_fwd_bwd");
}

SITK_F16_TWIN(sitk_mpp_corrupt)
extern "C" int sitk_mpp_corrupt(const float* tokens, const uint8_t* masked, const uint8_t* swap_draw,
                                const int32_t* random_patches, const uint8_t* replace_draw, const float* mask_token,
                                void* corrupted, int B, int P, int K, int ld, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_mpp_corrupt, tokens, masked, swap_draw, random_patches, replace_draw, mask_token, corrupted, B, P, K, ld, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(tokens && masked && replace_draw && mask_token && corrupted, "mpp_corrupt: null pointer");
  SITK_REQUIRE((swap_draw == nullptr) == (random_patches == nullptr), "mpp_corrupt: swap_draw and random_patches go together");
  SITK_REQUIRE(B > 0 && P > 0 && K > 0 && K % 4 == 0 && ld >= K && ld % 4 == 0, "mpp_corrupt: bad shape K=%d ld=%d", K, ld);
  const int64_t rows = (int64_t)B * P;
  dim3 grid(cdiv(ld / 4, 256), (unsigned)std::min<int64_t>(rows, 65535));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16)
    hipLaunchKernelGGL((mpp_corrupt_kernel<h16>), grid, dim3(256), 0, s, tokens, masked, swap_draw, random_patches,
                       replace_draw, mask_token, reinterpret_cast<h16*>(corrupted), rows, P, K, ld);
  else if (dtype == SITK_F32)
    hipLaunchKernelGGL((mpp_corrupt_kernel<float>), grid, dim3(256), 0, s, tokens, masked, swap_draw, random_patches,
                       replace_draw, mask_token, reinterpret_cast<float*>(corrupted), rows, P, K, ld);
  else { set_error("mpp_corrupt: bad dtype %d", dtype); return SITK_ERR_INVALID; }
  return check_launch("mpp_corrupt");
}

extern "C" int sitk_mpp_loss_fwd_bwd(const float* out, const float* tokens, const uint8_t* masked, float* loss,
                                     float* dout, int64_t rows, int K, int64_t n_masked_total, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(out && tokens && masked && loss && dout, "mpp_loss: null pointer");
  SITK_REQUIRE(rows > 0 && K > 0 && K % 4 == 0 && n_masked_total > 0, "mpp_loss: bad shape");
  const float inv = 1.0f / ((float)n_masked_total * (float)K);
  hipLaunchKernelGGL(mpp_loss_kernel, dim3(grid_for(rows, 1, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out,
                     tokens, masked, loss, dout, rows, K, inv);
  return check_launch("mpp_loss");
}

extern "C" int sitk_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                             float weight_decay, int nesterov, float grad_scale, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(param && grad && n > 0, "sgd_step: bad arguments");
  SITK_REQUIRE(momentum == 0.f || momentum_buf, "sgd_step: momentum needs a buffer");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     param, grad, momentum_buf, n, lr, momentum, weight_decay, nesterov, grad_scale);
  return check_launch("sgd_step");
}

extern "C" int sitk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, int step,
                              float grad_scale, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, bc1, bc2s, grad_scale);
  return check_launch("adam_step");
}

extern "C" int sitk_sgd_step_dev(float* param, float* grad, float* momentum_buf, int64_t n, const double* state,
                                 float momentum, float weight_decay, int nesterov, float grad_scale, int zero_grad,
                                 int64_t n_extra, int64_t keep_idx, float* keep_dst, const float* inv_loss_scale,
                                 int* nonfinite, float* grad2, const float* inv_loss_scale2, float keep_scale,
                                 sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(param && grad && state && n > 0 && n_extra >= 0, "sgd_step_dev: bad arguments");
  SITK_REQUIRE(momentum == 0.f || momentum_buf, "sgd_step_dev: momentum needs a buffer");
  SITK_REQUIRE(n % 4 == 0 || n_extra == 0, "sgd_step_dev: accumulators behind the gradients need n %% 4 == 0");
  hipLaunchKernelGGL(sgd_dev_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     param, grad, momentum_buf, n, state, momentum, weight_decay, nesterov, grad_scale, zero_grad, n_extra,
                     keep_idx, keep_dst, inv_loss_scale, nonfinite, grad2, inv_loss_scale2, keep_scale);
  return check_launch("sgd_step_dev");
}

extern "C" int sitk_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double* state,
                                  float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, float grad_scale,
                                  int zero_grad, int64_t n_extra, int64_t keep_idx, float* keep_dst,
                                  const float* inv_loss_scale, int* nonfinite, float* grad2, const float* inv_loss_scale2,
                                  float keep_scale, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && n > 0 && n_extra >= 0, "adam_step_dev: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, state, (double)beta1, (double)beta2);
  SITK_LAUNCH_CHECK("adam_advance");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, state,
                     beta1, beta2, eps, weight_decay, decoupled_wd, grad_scale, zero_grad, n_extra, keep_idx, keep_dst,
                     inv_loss_scale, nonfinite, grad2, inv_loss_scale2, keep_scale);
  return check_launch("adam_step_dev");
}

extern "C" int sitk_mpp_draw(const uint64_t* state, uint8_t* masked, uint8_t* swap_draw, int32_t* random_patches,
                             uint8_t* replace_draw, uint8_t* replaced_full, int B, int P, int n_mask, float p_swap,
                             float p_replace, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(state && masked && replace_draw && replaced_full, "mpp_draw: null pointer");
  SITK_REQUIRE((swap_draw == nullptr) == (random_patches == nullptr), "mpp_draw: swap_draw and random_patches go together");
  SITK_REQUIRE(B > 0 && P > 0 && P <= MPP_MAX_P && n_mask >= 0 && n_mask <= P, "mpp_draw: bad shape (P <= %d)", MPP_MAX_P);
  hipLaunchKernelGGL(mpp_draw_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), state, masked, swap_draw,
                     random_patches, replace_draw, replaced_full, P, n_mask, p_swap, p_replace);
  return check_launch("mpp_draw");
}

SITK_F16_TWIN(sitk_mpp_gather_corrupt)
extern "C" int sitk_mpp_gather_corrupt(const float* x, const uint16_t* table_pv, const int32_t* sample_idx, const float* mean,
                                       const float* stdv, const uint8_t* masked, const uint8_t* swap_draw,
                                       const int32_t* random_patches, const uint8_t* replace_draw, const float* mask_token,
                                       float* clean, void* corrupted, uint64_t* state, int B, int n_vertices, int C, int P, int V,
                                       int ld, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_mpp_gather_corrupt, x, table_pv, sample_idx, mean, stdv, masked, swap_draw, random_patches, replace_draw, mask_token, clean, corrupted, state, B, n_vertices, C, P, V, ld, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(x && table_pv && masked && replace_draw && mask_token && clean && corrupted, "mpp_gather_corrupt: null pointer");
  SITK_REQUIRE((swap_draw == nullptr) == (random_patches == nullptr), "mpp_gather_corrupt: swap_draw and random_patches go together");
  SITK_REQUIRE((mean == nullptr) == (stdv == nullptr), "mpp_gather_corrupt: mean and std go together");
  SITK_REQUIRE(C == 4, "mpp_gather_corrupt: channels-last gather is specialised for num_channels == 4 (got %d)", C);
  SITK_REQUIRE(B > 0 && P > 0 && V > 0 && n_vertices > 0 && n_vertices <= 65536, "mpp_gather_corrupt: bad shape");
  const int K = V * C;
  SITK_REQUIRE(ld >= K && ld % 4 == 0, "mpp_gather_corrupt: ld=%d must be >= V*C=%d and a multiple of 4", ld, K);
  const int64_t rows = (int64_t)B * P;
  dim3 grid(cdiv(ld / 4, 256), (unsigned)std::min<int64_t>(rows, 65535));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16)
    hipLaunchKernelGGL((mpp_gather_corrupt_kernel<h16>), grid, dim3(256), 0, s, x, table_pv, sample_idx, mean, stdv, masked, swap_draw,
                       random_patches, replace_draw, mask_token, clean, reinterpret_cast<h16*>(corrupted), state, rows,
                       n_vertices, P, V, K, ld);
  else if (dtype == SITK_F32)
    hipLaunchKernelGGL((mpp_gather_corrupt_kernel<float>), grid, dim3(256), 0, s, x, table_pv, sample_idx, mean, stdv, masked, swap_draw,
                       random_patches, replace_draw, mask_token, clean, reinterpret_cast<float*>(corrupted), state, rows,
                       n_vertices, P, V, K, ld);
  else { set_error("mpp_gather_corrupt: bad dtype %d", dtype); return SITK_ERR_INVALID; }
  return check_launch("mpp_gather_corrupt");
}

SITK_F16_TWIN(sitk_mpp_loss_fwd_bwd_ld)
extern "C" int sitk_mpp_loss_fwd_bwd_ld(const float* out, int ldo, const float* tokens, int ldt, const uint8_t* masked, float* loss,
                                        void* dout, int lddo, int dout_dtype, int64_t rows, int K, int64_t n_masked_total,
                                        float grad_scale, sitk_stream_t stream) {
  SITK_FORWARD_F16(dout_dtype, sitk_mpp_loss_fwd_bwd_ld, out, ldo, tokens, ldt, masked, loss, dout, lddo, dout_dtype, rows, K, n_masked_total, grad_scale, stream);
  using namespace sitk;
  SITK_REQUIRE(out && tokens && masked && loss && dout, "mpp_loss_ld: null pointer");
  SITK_REQUIRE(rows > 0 && K > 0 && K % 4 == 0 && n_masked_total > 0 && ldo >= K && ldt >= K && lddo >= K && ldo % 4 == 0 &&
               ldt % 4 == 0 && lddo % 4 == 0, "mpp_loss_ld: bad shape");
  const float inv = 1.0f / ((float)n_masked_total * (float)K);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(rows, 4, 1024);
  if (dout_dtype == SITK_H16)
    hipLaunchKernelGGL((mpp_loss_ld_kernel<h16>), dim3(grid), dim3(256), 0, s, out, ldo, tokens, ldt, masked, loss,
                       reinterpret_cast<h16*>(dout), lddo, rows, K, inv, grad_scale);
  else if (dout_dtype == SITK_F32)
    hipLaunchKernelGGL((mpp_loss_ld_kernel<float>), dim3(grid), dim3(256), 0, s, out, ldo, tokens, ldt, masked, loss,
                       reinterpret_cast<float*>(dout), lddo, rows, K, inv, grad_scale);
  else { set_error("mpp_loss_ld: bad dtype %d", dout_dtype); return SITK_ERR_INVALID; }
  return check_launch("mpp_loss_ld");
}

extern "C" int sitk_dropout_fwd(const float* x, const float* res, float* y, uint8_t* mask, int64_t n, float p, uint64_t* state,
                                sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(x && y && mask && state && n > 0 && p >= 0.f && p < 1.f, "dropout_fwd: bad arguments (0 <= p < 1)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for((n + 3) / 4, 256, 2048)), dim3(256), 0, s, x, res, y, mask, n, p, state);
  SITK_LAUNCH_CHECK("dropout_fwd");
  hipLaunchKernelGGL(advance_draw_kernel, dim3(1), dim3(1), 0, s, state);      // the next call draws a fresh mask
  return check_launch("dropout_advance");
}

extern "C" int sitk_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(dy && mask && dx && n > 0 && p >= 0.f && p < 1.f, "dropout_bwd: bad arguments");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, mask,
                     dx, n, 1.0f / (1.0f - p));
  return check_launch("dropout_bwd");
}

extern "C" int sitk_gelu_fwd(const float* u, float* g, int64_t n, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(u && g && n > 0, "gelu_fwd: bad arguments");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), u, g, n);
  return check_launch("gelu_fwd");
}

extern "C" int sitk_gelu_bwd(const float* dg, const float* u, float* du, int64_t n, sitk_stream_t stream) {
  using namespace sitk;
  SITK_REQUIRE(dg && u && du && n > 0, "gelu_bwd: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dg, u, du, n);
  return check_launch("gelu_bwd");
}
