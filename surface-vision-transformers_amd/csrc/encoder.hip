// sitk encoder: the whole vit_pytorch.vit.Transformer (models/sit.py:57,76; models/mpp.py:128) as
// one host call per direction.  The host side only sequences kernel launches on the caller's stream
// (no allocation, no sync).  Launches: one weight-staging launch per 12 layers; per layer forward 2 with the fused
// kernels (attention; to_out + norm + MLP + next block's norm + to_qkv) and 7 on the generic path (norm, to_qkv,
// attention, to_out, norm, net.0, net.3); per layer backward 4 fused (MLP backward, attention query side + key side,
// to_qkv backward + norm) and 9 generic, plus ONE weight-gradient launch and ONE LayerNorm-gradient reduction per
// backward slice.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace sitk {

// ---- weight staging: fp32 master -> compute-dtype copy and its transpose, all matrices of up to
// STAGE_MAX_MAT Linears in one launch (32x32 tiles through LDS) ------------------------------------
constexpr int STAGE_MAX_MAT = 48;
struct StageMat {
  const float* src;
  void* dst_c;  // (rows, cols) or null
  void* dst_t;  // (cols, rows)
  int rows, cols, tile_begin, tiles_c;
};
struct StageArgs {
  StageMat m[STAGE_MAX_MAT];
  int count;
};

template <typename T>
__global__ __launch_bounds__(256) void stage_weights_kernel(StageArgs a) {
  __shared__ float tile[32][33];
  int mi = 0;
  const int bid = blockIdx.x;
  while (mi + 1 < a.count && bid >= a.m[mi + 1].tile_begin) ++mi;
  const StageMat& M = a.m[mi];
  const int t = bid - M.tile_begin;
  const int r0 = (t / M.tiles_c) * 32, c0 = (t % M.tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  T* dc = reinterpret_cast<T*>(M.dst_c);
  T* dt = reinterpret_cast<T*>(M.dst_t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    const bool ok = r < M.rows && c < M.cols;
    const float v = ok ? M.src[(size_t)r * M.cols + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (ok && dc) dc[(size_t)r * M.cols + c] = from_f32<T>(v);
  }
  __syncthreads();
  if (dt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (c < M.cols && r < M.rows) dt[(size_t)c * M.rows + r] = from_f32<T>(tile[tx][ty + 8 * i]);
    }
  }
}

// bf16, rows % 4 == 0 and cols % 4 == 0 (every encoder Linear): 64 x 64 tiles, 16-byte loads, 8-byte stores in both
// orientations (the 32 x 32 scalar kernel above took 21.6 us for the 48 matrices of SiT-tiny: 2-byte stores)
__global__ __launch_bounds__(256) void stage_weights_vec_kernel(StageArgs a) {
  __shared__ float tile[64][65];
  int mi = 0;
  const int bid = blockIdx.x;
  while (mi + 1 < a.count && bid >= a.m[mi + 1].tile_begin) ++mi;
  const StageMat& M = a.m[mi];
  const int t = bid - M.tile_begin;
  const int r0 = (t / M.tiles_c) * 64, c0 = (t % M.tiles_c) * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  h16* dc = reinterpret_cast<h16*>(M.dst_c);
  h16* dt = reinterpret_cast<h16*>(M.dst_t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < M.rows && c < M.cols) {
      v = *reinterpret_cast<const f32x4*>(M.src + (size_t)r * M.cols + c);
      if (dc) {
        h16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (h16)v[e];
        *reinterpret_cast<h16x4*>(dc + (size_t)r * M.cols + c) = o;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[ty + 16 * i][4 * tx + e] = v[e];
  }
  __syncthreads();
  if (dt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 16 * i, r = r0 + 4 * tx;
      if (c < M.cols && r < M.rows) {
        h16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (h16)tile[4 * tx + e][ty + 16 * i];
        *reinterpret_cast<h16x4*>(dt + (size_t)c * M.rows + r) = o;
      }
    }
  }
}

template <typename T>
static int launch_stage(const StageArgs& a, int total_tiles, hipStream_t s) {
  hipLaunchKernelGGL((stage_weights_kernel<T>), dim3(total_tiles), dim3(256), 0, s, a);
  return check_launch("stage_weights");
}

// ---- workspace layout --------------------------------------------------------------------------
struct LayerActs {
  char *wqkv_c, *wqkv_t, *wo_c, *wo_t, *w1_c, *w1_t, *w2_c, *w2_t;  // compute dtype
  float *x_in;                                                       // (R, D) fp32 (layers >= 1)
  float *mean1, *rstd1, *mean2, *rstd2, *lse, *xmid;
  char *h1, *qkv, *o, *h2, *u, *g;
};
struct Scratch {
  // Operands of the weight gradients that backward produces (du, dqkv, the compute-dtype residual gradients) exist
  // once per layer when the weight gradients of a whole backward slice run as ONE launch at its end (wg_batch).
  std::vector<char*> du, dqkv, dxAc, dxBc;
  std::vector<char*> g;  // gelu(u), only when the fused forward does not save it
  char *dh, *d_o;
  bool wg_batch;
  float *delta, *ping, *pong;
  // backward: residual-gradient ping-pong (fp32 dxB besides the caller's dx) and compute-dtype copies
  float* dxB;
  float* ln_partials;
  size_t ln_partial_floats;
  char* wgrad_ws;
  size_t wgrad_ws_bytes;
  char* wgrad_ws_side;          // slab of the side-stream weight-gradient launches (sitk_encoder_bwd_overlap)
  size_t wgrad_ws_side_bytes;
};

struct Layout {
  std::vector<LayerActs> layers;
  Scratch scratch;
  size_t acts_bytes = 0, scratch_bytes = 0;
};

// the fused LayerNorm + MLP kernels (mlp_fused.hip) cover the bf16 tiny shape
static bool mlp_fused(const sitk_encoder_cfg& c) { return sitk_mlp_fused_supported(c.dim, c.mlp_dim, c.dtype) != 0; }
// gelu(u) is stored by the fused forward (the stores hide under its VALU-bound loop) rather than recomputed
// and re-stored by the fused backward, whose loop carries more memory traffic: 0.7 % of the step, measured.
static bool g_in_fwd() { return true; }
// the fused LayerNorm + to_qkv kernels (ln_gemm_fused.hip)
static bool qkv_fused(const sitk_encoder_cfg& c) { return sitk_ln_gemm_fused_supported(c.dim, 3 * c.heads * 64, c.dtype) != 0; }

static Layout make_layout(const sitk_encoder_cfg& c, char* acts, char* scratch) {
  Layout L;
  const size_t es = c.dtype == SITK_H16 ? 2 : 4;
  const size_t R = (size_t)c.B * c.N, D = c.dim, I = (size_t)c.heads * 64, M = c.mlp_dim;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = acts ? acts + off : nullptr; off += align_up(bytes, 256); return p; };
  L.layers.resize(c.depth);
  for (int l = 0; l < c.depth; ++l) {
    LayerActs& a = L.layers[l];
    a.wqkv_c = take(3 * I * D * es); a.wqkv_t = take(3 * I * D * es);
    a.wo_c = take(D * I * es);       a.wo_t = take(D * I * es);
    a.w1_c = take(M * D * es);       a.w1_t = take(M * D * es);
    a.w2_c = take(M * D * es);       a.w2_t = take(M * D * es);
  }
  for (int l = 0; l < c.depth; ++l) {
    LayerActs& a = L.layers[l];
    a.x_in = l > 0 ? (float*)take(R * D * 4) : nullptr;
    a.mean1 = (float*)take(R * 4); a.rstd1 = (float*)take(R * 4);
    a.mean2 = (float*)take(R * 4); a.rstd2 = (float*)take(R * 4);
    a.lse = (float*)take((size_t)c.B * c.heads * c.N * 4);
    a.xmid = (float*)take(R * D * 4);
    a.h1 = take(R * D * es); a.qkv = take(R * 3 * I * es); a.o = take(R * I * es);
    a.h2 = take(R * D * es); a.u = take(R * M * es);
    a.g = take(R * M * es);
  }
  L.acts_bytes = off;
  off = 0;
  auto stake = [&](size_t bytes) { char* p = scratch ? scratch + off : nullptr; off += align_up(bytes, 256); return p; };
  sitk_wgrad_desc wg4[4] = {};
  {
    const int dims[4][2] = {{(int)D, (int)M}, {(int)M, (int)D}, {(int)D, (int)I}, {3 * (int)I, (int)D}};
    for (int i = 0; i < 4; ++i) {
      wg4[i].M = (int)R; wg4[i].N = dims[i][0]; wg4[i].K = dims[i][1]; wg4[i].lddy = dims[i][0]; wg4[i].ldx = dims[i][1];
    }
  }
  L.scratch.wg_batch = sitk_gemm_wgrad_group_ws_bytes(wg4, 4, c.dtype) > 0 && 4 * c.depth <= 48;
  const int nslot = L.scratch.wg_batch ? c.depth : 1;
  L.scratch.du.resize(nslot); L.scratch.g.resize(nslot); L.scratch.dqkv.resize(nslot);
  L.scratch.dxAc.resize(nslot); L.scratch.dxBc.resize(nslot);
  for (int i = 0; i < nslot; ++i) {
    L.scratch.du[i] = stake(R * M * es);
    L.scratch.g[i] = nullptr;
    L.scratch.dqkv[i] = stake(R * 3 * I * es);
    L.scratch.dxAc[i] = stake(R * D * es);
    L.scratch.dxBc[i] = stake(R * D * es);
  }
  L.scratch.dh = stake(R * D * es);
  L.scratch.d_o = stake(R * I * es);
  L.scratch.delta = (float*)stake((size_t)c.B * c.heads * c.N * 4);
  L.scratch.ping = (float*)stake(R * D * 4);
  L.scratch.pong = (float*)stake(R * D * 4);
  L.scratch.dxB = (float*)stake(R * D * 4);
  {  // slab of the large-tile weight-gradient path (0 bytes when the shapes are not eligible)
    std::vector<sitk_wgrad_desc> wg(4 * nslot);
    for (int i = 0; i < 4 * nslot; ++i) wg[i] = wg4[i % 4];
    // The slab is sized for the WORST slice length, not for the whole depth: how many token splits a launch takes depends on how
    // its tiles fill the chip's rounds, and a middle-sized slice can need more slab than all layers together (SiT-base: 4 layers
    // = 2.25 rounds of tile pairs -> four token splits -> 453 MB, 12 layers = 6.75 rounds -> no split -> 340 MB).  Until round 6
    // only the full depth and one layer were priced, and a data-parallel step of SiT-base in its default three slices fell back to
    // the generic 64 x 64 weight-gradient tiles for two of them without a word: 49.1 instead of 39.9 ms per step.
    L.scratch.wgrad_ws_bytes = 0;
    for (int k = 1; k <= nslot; ++k)
      L.scratch.wgrad_ws_bytes = std::max(L.scratch.wgrad_ws_bytes, sitk_gemm_wgrad_group_ws_bytes(wg.data(), 4 * k, c.dtype));
    if (L.scratch.wgrad_ws_bytes) L.scratch.wgrad_ws_bytes += (size_t)128 * 128 * 192 * 4;   // + the patch embedding's and the caller's extra tiles (sitk_encoder_bwd_extra)
    L.scratch.wgrad_ws = stake(L.scratch.wgrad_ws_bytes);
    {
      sitk_wgrad_desc wg12[12];
      for (int i = 0; i < 12; ++i) wg12[i] = wg4[i % 4];
      L.scratch.wgrad_ws_side_bytes = std::max(std::max(sitk_gemm_wgrad_group_ws_bytes(wg4, 4, c.dtype), sitk_gemm_wgrad_group_ws_bytes(wg12, 8, c.dtype)),
                                               sitk_gemm_wgrad_group_ws_bytes(wg12, 12, c.dtype));     // one, two or three layers per side launch
    }
    L.scratch.wgrad_ws_side = stake(L.scratch.wgrad_ws_side_bytes);
  }
  L.scratch.ln_partial_floats = sitk_layernorm_bwd_partial_floats((int64_t)R, (int)D);
  L.scratch.ln_partials = (float*)stake(L.scratch.ln_partial_floats * 4 * 2 * c.depth);   // one region per LayerNorm
  L.scratch_bytes = off;
  return L;
}

// one timeline mark behind a launch (no-op without a timeline)
#define SITK_MARK(label) SITK_TRY(sitk_timeline_mark(c.timeline, label, stream))

static int check_cfg(const sitk_encoder_cfg* c) {
  SITK_REQUIRE(c != nullptr, "encoder: null config");
  SITK_REQUIRE(c->B > 0 && c->N > 0 && c->depth > 0 && c->heads > 0, "encoder: bad shape");
  SITK_REQUIRE(c->dim > 0 && c->dim % 8 == 0 && c->dim <= 1024, "encoder: dim=%d must be a multiple of 8 and <= 1024", c->dim);
  SITK_REQUIRE(c->mlp_dim > 0 && c->mlp_dim % 8 == 0, "encoder: mlp_dim=%d must be a multiple of 8", c->mlp_dim);
  SITK_REQUIRE(c->dtype == SITK_H16 || c->dtype == SITK_F32, "encoder: bad dtype %d", c->dtype);
  return SITK_OK;
}

static sitk_gemm_desc gemm_desc(int M, int N, int K, const void* A, int lda, int a_f32, const void* W, int epi,
                                void* out, int ldo, int out_f32) {
  sitk_gemm_desc d = {};
  d.M = M; d.N = N; d.K = K; d.A = A; d.lda = lda; d.a_is_f32 = a_f32; d.W = W; d.ldw = K;
  d.epilogue = epi; d.out = out; d.ldo = ldo; d.out_is_f32 = out_f32;
  return d;
}

static sitk_wgrad_desc wgrad_desc(int M, int N, int K, const void* dY, int dy_f32, const void* X, float* dW, float* db) {
  sitk_wgrad_desc d = {};
  d.M = M; d.N = N; d.K = K; d.dY = dY; d.lddy = N; d.dy_is_f32 = dy_f32; d.X = X; d.ldx = K; d.dW = dW; d.lddw = K; d.db = db;
  return d;
}

static int stage_all(const sitk_encoder_cfg& c, const sitk_layer_params* P, const Layout& L, hipStream_t s) {
  const int D = c.dim, I = c.heads * 64, M = c.mlp_dim;
  const bool f32 = c.dtype == SITK_F32;
  StageArgs a;
  a.count = 0;
  int tiles = 0;
  const bool vec = !f32 && D % 4 == 0 && I % 4 == 0 && M % 4 == 0;   // 64 x 64 tiles, vector accesses
  const int ts = vec ? 64 : 32;
  auto flush = [&]() -> int {
    if (a.count == 0) return SITK_OK;
    int e;
    if (vec) {
      hipLaunchKernelGGL(stage_weights_vec_kernel, dim3(tiles), dim3(256), 0, s, a);
      e = check_launch("stage_weights_vec");
    } else {
      e = f32 ? launch_stage<float>(a, tiles, s) : launch_stage<h16>(a, tiles, s);
    }
    a.count = 0;
    tiles = 0;
    return e;
  };
  auto add = [&](const float* src, void* dc, void* dt, int rows, int cols) {
    StageMat& m = a.m[a.count++];
    m.src = src; m.dst_c = f32 ? nullptr : dc; m.dst_t = dt; m.rows = rows; m.cols = cols;
    m.tile_begin = tiles; m.tiles_c = cdiv(cols, ts);
    tiles += cdiv(rows, ts) * m.tiles_c;
  };
  for (int l = 0; l < c.depth; ++l) {
    if (a.count + 4 > STAGE_MAX_MAT) SITK_TRY(flush());
    const LayerActs& w = L.layers[l];
    add(P[l].wqkv, w.wqkv_c, w.wqkv_t, 3 * I, D);
    add(P[l].wo, w.wo_c, w.wo_t, D, I);
    add(P[l].w1, w.w1_c, w.w1_t, M, D);
    add(P[l].w2, w.w2_c, w.w2_t, D, M);
  }
  return flush();
}

}  // namespace sitk

using namespace sitk;

SITK_F16_TWIN(sitk_encoder_acts_bytes)
extern "C" size_t sitk_encoder_acts_bytes(const sitk_encoder_cfg* cfg) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_acts_bytes, cfg);
  if (check_cfg(cfg)) return 0;
  return make_layout(*cfg, nullptr, nullptr).acts_bytes;
}
SITK_F16_TWIN(sitk_encoder_scratch_bytes)
extern "C" size_t sitk_encoder_scratch_bytes(const sitk_encoder_cfg* cfg) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_scratch_bytes, cfg);
  if (check_cfg(cfg)) return 0;
  return make_layout(*cfg, nullptr, nullptr).scratch_bytes;
}

// the slab make_layout reserves for the one weight-gradient launch of a backward slice (tests: every slice length must fit)
SITK_F16_TWIN(sitk_encoder_wgrad_slab_bytes)
extern "C" size_t sitk_encoder_wgrad_slab_bytes(const sitk_encoder_cfg* cfg) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_wgrad_slab_bytes, cfg);
  if (check_cfg(cfg)) return 0;
  return make_layout(*cfg, nullptr, nullptr).scratch.wgrad_ws_bytes;
}

SITK_F16_TWIN(sitk_encoder_stage_weights)
extern "C" int sitk_encoder_stage_weights(const sitk_encoder_cfg* cfg, const sitk_layer_params* P, void* acts, size_t acts_bytes,
                                          sitk_stream_t stream) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_stage_weights, cfg, P, acts, acts_bytes, stream);
  SITK_TRY(check_cfg(cfg));
  SITK_REQUIRE(P && acts, "encoder_stage_weights: null pointer");
  Layout L = make_layout(*cfg, (char*)acts, nullptr);
  SITK_REQUIRE(acts_bytes >= L.acts_bytes, "encoder_stage_weights: acts workspace %zu < %zu", acts_bytes, L.acts_bytes);
  return stage_all(*cfg, P, L, reinterpret_cast<hipStream_t>(stream));
}

SITK_F16_TWIN(sitk_encoder_fwd)
extern "C" int sitk_encoder_fwd(const sitk_encoder_cfg* cfg, const sitk_layer_params* P, const float* x_in, float* x_out,
                                void* acts, size_t acts_bytes, void* scratch, size_t scratch_bytes, int save, sitk_stream_t stream) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_fwd, cfg, P, x_in, x_out, acts, acts_bytes, scratch, scratch_bytes, save, stream);
  SITK_TRY(check_cfg(cfg));
  SITK_REQUIRE(P && x_in && x_out && acts && scratch, "encoder_fwd: null pointer");
  const sitk_encoder_cfg& c = *cfg;
  Layout L = make_layout(c, (char*)acts, (char*)scratch);
  SITK_REQUIRE(acts_bytes >= L.acts_bytes, "encoder_fwd: acts workspace %zu < %zu", acts_bytes, L.acts_bytes);
  SITK_REQUIRE(scratch_bytes >= L.scratch_bytes, "encoder_fwd: scratch workspace %zu < %zu", scratch_bytes, L.scratch_bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int R = c.B * c.N, D = c.dim, I = c.heads * 64, M = c.mlp_dim, dt = c.dtype;
  const bool f32 = dt == SITK_F32;
  const float scale = 0.125f;  // dim_head ** -0.5, dim_head = 64

  SITK_MARK("begin");
  if (!(save & 2)) SITK_TRY(stage_all(c, P, L, s));
  SITK_MARK("stage_weights");
  save &= 1;

  const float* x = x_in;
  bool have_qkv = false;   // the previous block's fused kernel already produced this block's h1 / statistics / qkv
  for (int l = 0; l < c.depth; ++l) {
    const LayerActs& a = L.layers[save ? l : 0];
    const LayerActs& w = L.layers[l];
    const void* wqkv = f32 ? (const void*)P[l].wqkv : w.wqkv_c;
    const void* wo = f32 ? (const void*)P[l].wo : w.wo_c;
    const void* w1 = f32 ? (const void*)P[l].w1 : w.w1_c;
    const void* w2 = f32 ? (const void*)P[l].w2 : w.w2_c;
    float* xnext = (l == c.depth - 1) ? x_out : (save ? L.layers[l + 1].x_in : ((l & 1) ? L.scratch.pong : L.scratch.ping));

    if (have_qkv) {
      // this block's LayerNorm + to_qkv were appended to the previous block's fused kernel
    } else if (qkv_fused(c)) {
      SITK_TRY(sitk_ln_gemm_fwd(x, P[l].ln1_w, P[l].ln1_b, wqkv, a.h1, a.mean1, a.rstd1, a.qkv, R, D, 3 * I, dt, stream));
      SITK_MARK("ln_gemm_fwd");
    } else {
      SITK_TRY(sitk_layernorm_fwd(x, P[l].ln1_w, P[l].ln1_b, a.h1, a.mean1, a.rstd1, R, D, dt, stream));
      SITK_MARK("layernorm_fwd");
      sitk_gemm_desc g1 = gemm_desc(R, 3 * I, D, a.h1, D, 0, wqkv, SITK_EPI_STORE, a.qkv, 3 * I, 0);
      SITK_TRY(sitk_gemm_nt(&g1, dt, stream));
      SITK_MARK("gemm:to_qkv");
    }
    have_qkv = false;
    SITK_TRY(sitk_attention_fwd(a.qkv, a.o, a.lse, c.B, c.N, c.heads, scale, dt, stream));
    SITK_MARK("attn_fwd");
    if (mlp_fused(c) && sitk_attn_out_mlp_fused_supported(R, D, I, M, dt)) {   // to_out + residual + norm + MLP + residual
      if (l + 1 < c.depth && qkv_fused(c)) {                                    // ... + the next block's norm + to_qkv
        const LayerActs& an = L.layers[save ? l + 1 : 0];
        const void* wqkv_n = L.layers[l + 1].wqkv_c;
        SITK_TRY(sitk_attn_out_mlp_next_fwd(a.o, wo, P[l].bo, x, a.xmid, P[l].ln2_w, P[l].ln2_b, w1, P[l].b1, w2, P[l].b2, a.h2,
                                            a.mean2, a.rstd2, a.u, save ? a.g : nullptr, xnext, P[l + 1].ln1_w, P[l + 1].ln1_b,
                                            wqkv_n, an.h1, an.mean1, an.rstd1, an.qkv, 3 * I, R, D, I, M, dt, stream));
        SITK_MARK("block_tail_next");
        have_qkv = true;
      } else {
        SITK_TRY(sitk_attn_out_mlp_fwd(a.o, wo, P[l].bo, x, a.xmid, P[l].ln2_w, P[l].ln2_b, w1, P[l].b1, w2, P[l].b2, a.h2,
                                       a.mean2, a.rstd2, a.u, save ? a.g : nullptr, xnext, R, D, I, M, dt, stream));
        SITK_MARK("block_tail");
      }
      x = xnext;
      continue;
    }
    sitk_gemm_desc g2 = gemm_desc(R, D, I, a.o, I, 0, wo, SITK_EPI_BIAS_RES, a.xmid, D, 1);
    g2.bias = P[l].bo; g2.aux = x; g2.ldaux = D;
    SITK_TRY(sitk_gemm_nt(&g2, dt, stream));
    SITK_MARK("gemm:to_out");
    if (mlp_fused(c)) {
      SITK_TRY(sitk_mlp_fwd(a.xmid, P[l].ln2_w, P[l].ln2_b, w1, P[l].b1, w2, P[l].b2, a.h2, a.mean2, a.rstd2, a.u, save ? a.g : nullptr,
                            xnext, R, D, M, dt, stream));
      SITK_MARK("mlp_fwd");
      x = xnext;
      continue;
    }
    SITK_TRY(sitk_layernorm_fwd(a.xmid, P[l].ln2_w, P[l].ln2_b, a.h2, a.mean2, a.rstd2, R, D, dt, stream));
    SITK_MARK("layernorm_fwd");
    sitk_gemm_desc g3 = gemm_desc(R, M, D, a.h2, D, 0, w1, SITK_EPI_BIAS_GELU, a.u, M, 0);
    g3.bias = P[l].b1; g3.out2 = a.g;
    SITK_TRY(sitk_gemm_nt(&g3, dt, stream));
    SITK_MARK("gemm:net0");
    sitk_gemm_desc g4 = gemm_desc(R, D, M, a.g, M, 0, w2, SITK_EPI_BIAS_RES, xnext, D, 1);
    g4.bias = P[l].b2; g4.aux = a.xmid; g4.ldaux = D;
    SITK_TRY(sitk_gemm_nt(&g4, dt, stream));
    SITK_MARK("gemm:net3");
    x = xnext;
  }
  (void)s;
  return SITK_OK;
}

SITK_F16_TWIN(sitk_encoder_bwd)
extern "C" int sitk_encoder_bwd(const sitk_encoder_cfg* cfg, const sitk_layer_params* P, const sitk_layer_params* G,
                                const float* x_in, float* dx, void* acts, size_t acts_bytes, void* scratch,
                                size_t scratch_bytes, int layer_begin, int layer_end, sitk_stream_t stream) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_bwd, cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, stream);
  return sitk_encoder_bwd_embed(cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, nullptr,
                                nullptr, nullptr, stream);
}

SITK_F16_TWIN(sitk_encoder_bwd_embed)
extern "C" int sitk_encoder_bwd_embed(const sitk_encoder_cfg* cfg, const sitk_layer_params* P, const sitk_layer_params* G,
                                      const float* x_in, float* dx, void* acts, size_t acts_bytes, void* scratch,
                                      size_t scratch_bytes, int layer_begin, int layer_end, const sitk_wgrad_desc* embed,
                                      void* dx_c, int* embed_done, sitk_stream_t stream) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_bwd_embed, cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, embed, dx_c, embed_done, stream);
  return sitk_encoder_bwd_extra(cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, embed, dx_c,
                                embed_done, nullptr, 0, nullptr, stream);
}

SITK_F16_TWIN(sitk_encoder_bwd_extra)
extern "C" int sitk_encoder_bwd_extra(const sitk_encoder_cfg* cfg, const sitk_layer_params* P, const sitk_layer_params* G,
                                      const float* x_in, float* dx, void* acts, size_t acts_bytes, void* scratch,
                                      size_t scratch_bytes, int layer_begin, int layer_end, const sitk_wgrad_desc* embed,
                                      void* dx_c, int* embed_done, const sitk_wgrad_desc* extra, int n_extra, int* extra_done,
                                      sitk_stream_t stream) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_bwd_extra, cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, embed, dx_c, embed_done, extra, n_extra, extra_done, stream);
  return sitk_encoder_bwd_overlap(cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, embed, dx_c,
                                  embed_done, extra, n_extra, extra_done, nullptr, stream);
}

// accessors of the overlap object (core.hip, compiled once)
extern "C" void* sitk_overlap_stream_(sitk_overlap* o);
extern "C" void* sitk_overlap_event_(sitk_overlap* o, int i);
extern "C" int sitk_overlap_layers_(const sitk_overlap* o);
extern "C" int sitk_overlap_cus_(const sitk_overlap* o);
extern "C" int sitk_overlap_caller_joins_(const sitk_overlap* o);
extern "C" int sitk_overlap_max_layers_(const sitk_overlap* o);
extern "C" int sitk_overlap_tail_cus_(const sitk_overlap* o);
extern "C" int sitk_overlap_group_(const sitk_overlap* o);
extern "C" void sitk_overlap_reset_done_(sitk_overlap* o);
extern "C" void* sitk_overlap_next_done_(sitk_overlap* o);

SITK_F16_TWIN(sitk_encoder_bwd_overlap)
extern "C" int sitk_encoder_bwd_overlap(const sitk_encoder_cfg* cfg, const sitk_layer_params* P, const sitk_layer_params* G,
                                        const float* x_in, float* dx, void* acts, size_t acts_bytes, void* scratch,
                                        size_t scratch_bytes, int layer_begin, int layer_end, const sitk_wgrad_desc* embed,
                                        void* dx_c, int* embed_done, const sitk_wgrad_desc* extra, int n_extra, int* extra_done,
                                        sitk_overlap* overlap, sitk_stream_t stream) {
  SITK_FORWARD_F16(cfg ? cfg->dtype : -1, sitk_encoder_bwd_overlap, cfg, P, G, x_in, dx, acts, acts_bytes, scratch, scratch_bytes, layer_begin, layer_end, embed, dx_c, embed_done, extra, n_extra, extra_done, overlap, stream);
  if (embed_done) *embed_done = 0;
  if (extra_done) *extra_done = 0;
  SITK_TRY(check_cfg(cfg));
  SITK_REQUIRE(P && G && x_in && dx && acts && scratch, "encoder_bwd: null pointer");
  const sitk_encoder_cfg& c = *cfg;
  SITK_REQUIRE(0 <= layer_begin && layer_begin < layer_end && layer_end <= c.depth, "encoder_bwd: bad layer range [%d, %d)", layer_begin, layer_end);
  Layout L = make_layout(c, (char*)acts, (char*)scratch);
  SITK_REQUIRE(acts_bytes >= L.acts_bytes, "encoder_bwd: acts workspace %zu < %zu", acts_bytes, L.acts_bytes);
  SITK_REQUIRE(scratch_bytes >= L.scratch_bytes, "encoder_bwd: scratch workspace %zu < %zu", scratch_bytes, L.scratch_bytes);
  const int R = c.B * c.N, D = c.dim, I = c.heads * 64, M = c.mlp_dim, dt = c.dtype;
  const float scale = 0.125f;
  const Scratch& S = L.scratch;

  // The residual gradient lives in two fp32 buffers (A = the caller's dx, B = scratch) plus a copy of
  // each in the compute dtype (Ac, Bc) that feeds the GEMMs' operand loads: LN2' reads A and writes
  // B, LN1' reads B and writes A, so d(x_out) [A] and d(x_mid) [B] both survive until the layer's
  // four weight gradients run as ONE grouped launch.
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  hipStream_t side = reinterpret_cast<hipStream_t>(sitk_overlap_stream_(overlap));
  // (a timeline times ONE stream: side launches are off while one is attached -- except in the diagnostic build under
  // SITK_TIMELINE_SIDE=1, where tools/dp_cu_budget.py wants the main chain's marks WITH the side stream running)
  static const int timeline_side = sitk_ab_switch("SITK_TIMELINE_SIDE", 0);
  const int side_max = (overlap && S.wg_batch && S.wgrad_ws_side_bytes && (!c.timeline || timeline_side))
                           ? std::min(sitk_overlap_layers_(overlap), layer_end - layer_begin) : 0;
  const int side_cus = sitk_overlap_cus_(overlap);
  const int side_group = sitk_overlap_group_(overlap);      // layers per side launch (2; sitk_overlap_set_group)
  sitk_overlap_reset_done_(overlap);
  int n_side = 0;
  std::vector<sitk_wgrad_desc> wg_side;
  std::vector<LnFinalizeEntry> ln_entries;
  ln_entries.reserve(2 * (layer_end - layer_begin));
  // slot of a layer's weight-gradient operands: its own when the slice's weight gradients are batched, else shared
  auto slot = [&](int l) { return S.wg_batch ? l : 0; };
  // the patch embedding's weight gradient joins this slice's launch when the slice ends at layer 0, its shape takes the
  // large-tile path and the slab has room for its tiles (sized for them in make_layout)
  sitk_wgrad_desc emb = {};
  bool take_embed = false;
  if (embed && dx_c && embed_done && layer_begin == 0 && S.wg_batch) {
    emb = *embed;
    emb.dY = dx_c; emb.lddy = D; emb.dy_is_f32 = 0;
    take_embed = sitk_gemm_wgrad_group_ws_bytes(&emb, 1, dt) > 0;
  }
  std::vector<sitk_wgrad_desc> wg_all;
  wg_all.reserve(4 * (layer_end - layer_begin));
  SITK_MARK("begin");
  // the compute-dtype copy of the incoming dx: written by the first MLP backward launch itself where that is the fused kernel
  static const int cast_in_mlp = sitk_ab_switch("SITK_BWD_CAST_FOLD", 1);
  bool dx_cast_pending = mlp_fused(c) && cast_in_mlp;
  if (!dx_cast_pending) {
    SITK_TRY(sitk_cast_rows(dx, D, S.dxAc[slot(layer_end - 1)], D, R, D, dt, stream));
    SITK_MARK("cast_rows");
  }
  bool mlp_bwd_done = false;
  for (int l = layer_end - 1; l >= layer_begin; --l) {
    const LayerActs& a = L.layers[l];
    const float* xl = l == 0 ? x_in : a.x_in;
    const int sl = slot(l);
    // where LN1' leaves the compute-dtype copy of dx for the MLP backward of layer l - 1 (never this layer's own
    // slot while its weight gradients are still pending; layer 0 of a batched slice needs no copy at all)
    char* dxc_next = l > 0 ? S.dxAc[slot(l - 1)] : (take_embed ? (char*)dx_c : (S.wg_batch ? nullptr : S.dxAc[0]));
    char* dxAc = S.dxAc[sl];
    char* dxBc = S.dxBc[sl];
    char* du = S.du[sl];
    char* dqkv = S.dqkv[sl];
    // ---- MLP branch: x_out = xmid + W2 gelu(W1 LN2(xmid) + b1) + b2 ----
    float* part2 = S.ln_partials + (size_t)(2 * l + 1) * S.ln_partial_floats;
    const void* gact = a.g;
    if (mlp_fused(c)) {
      if (!mlp_bwd_done) {      // (done already when the previous layer's d to_qkv launch carried it: the pair kernel below)
        if (dx_cast_pending) {
          SITK_TRY(sitk_mlp_bwd_cast(dx, dxAc, a.xmid, a.mean2, a.rstd2, P[l].ln2_w, a.w2_t, a.w1_t, a.u, du, S.dxB, dxBc, part2, R, D,
                                     M, dt, stream));
        } else {
          SITK_TRY(sitk_mlp_bwd(dx, dxAc, a.xmid, a.mean2, a.rstd2, P[l].ln2_w, a.w2_t, a.w1_t, a.u, du, S.dxB, dxBc, part2, R, D, M, dt,
                                stream));
        }
        dx_cast_pending = false;
        SITK_MARK("mlp_bwd");
      }
      mlp_bwd_done = false;
      ln_entries.push_back(LnFinalizeEntry{part2, G[l].ln2_w, G[l].ln2_b, (int)(sitk_mlp_bwd_partial_floats(R) / (2 * D))});
      gact = a.g ? a.g : S.g[sl];
    } else {
      sitk_gemm_desc d1 = gemm_desc(R, M, D, dxAc, D, 0, a.w2_t, SITK_EPI_DGELU, du, M, 0);
      d1.aux = a.u; d1.ldaux = M;
      SITK_TRY(sitk_gemm_nt(&d1, dt, stream));
      SITK_MARK("gemm:dnet3");
      sitk_gemm_desc d2 = gemm_desc(R, D, M, du, M, 0, a.w1_t, SITK_EPI_STORE, S.dh, D, 0);
      SITK_TRY(sitk_gemm_nt(&d2, dt, stream));
      SITK_MARK("gemm:dnet0");
      SITK_TRY(layernorm_bwd_deferred(S.dh, a.xmid, a.mean2, a.rstd2, P[l].ln2_w, dx, S.dxB, dxBc, part2, R, D, dt, hs));
      SITK_MARK("layernorm_bwd");
      ln_entries.push_back(LnFinalizeEntry{part2, G[l].ln2_w, G[l].ln2_b, 0});
    }
    // ---- attention branch: xmid = x + Wo attn(Wqkv LN1(x)) + bo ----
    if (sitk_attention_bwd_proj_supported(c.N, D, dt)) {     // d_o = dx_mid Wo inside the query-side kernel
      // (query side + key side: ONE launch where the sequence is LDS-resident, two otherwise)
      SITK_TRY(sitk_attention_bwd_phases(a.qkv, a.o, nullptr, dxBc, a.wo_t, S.d_o, a.lse, S.delta, dqkv, c.B, c.N, c.heads, D, scale,
                                         dt, 3, stream));
      SITK_MARK("attn_bwd");
    } else {
      sitk_gemm_desc d3 = gemm_desc(R, I, D, dxBc, D, 0, a.wo_t, SITK_EPI_STORE, S.d_o, I, 0);
      SITK_TRY(sitk_gemm_nt(&d3, dt, stream));
      SITK_MARK("gemm:dto_out");
      SITK_TRY(sitk_attention_bwd_phases(a.qkv, a.o, S.d_o, nullptr, nullptr, nullptr, a.lse, S.delta, dqkv, c.B, c.N, c.heads, D, scale,
                                         dt, 3, stream));
      SITK_MARK("attn_bwd");
    }
    // ---- the four weight (+ bias) gradients of the layer, one launch ----
    sitk_wgrad_desc wg[4] = {
        wgrad_desc(R, D, M, dxAc, 0, gact, G[l].w2, G[l].b2),
        wgrad_desc(R, M, D, du, 0, a.h2, G[l].w1, G[l].b1),
        wgrad_desc(R, D, I, dxBc, 0, a.o, G[l].wo, G[l].bo),
        wgrad_desc(R, 3 * I, D, dqkv, 0, a.h1, G[l].wqkv, nullptr),
    };
    if (S.wg_batch && n_side < side_max) {
      // This layer's weight gradients run beside the rest of the chain, on the side stream.  Two layers per launch: their 20
      // double tiles (256 x 192) in two token halves are 40 workgroups for the CUs the chain leaves idle, followed by a slab
      // reduction of 12 - 30 us on the same stream (three or four layers per launch need no split but start later: measured
      // 2.59 / 2.50 ms per step against 2.43, profiles/README.md round 3).
      wg_side.insert(wg_side.end(), wg, wg + 4);
      ++n_side;
      if ((int)wg_side.size() == 4 * side_group || n_side == side_max) {
        hipEvent_t ev = reinterpret_cast<hipEvent_t>(sitk_overlap_event_(overlap, n_side - 1));
        if (hipEventRecord(ev, hs) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess) {
          set_error("encoder_bwd_overlap: fork of the side stream failed");
          return SITK_ERR_LAUNCH;
        }
        SITK_TRY(sitk_gemm_wgrad_group_ws_cus(wg_side.data(), (int)wg_side.size(), dt, S.wgrad_ws_side, S.wgrad_ws_side_bytes,
                                              side_cus, side));
        wg_side.clear();
        // the weight / bias gradients of this launch's layers are final behind it: a data-parallel caller reduces them from
        // here on (sitk_overlap_wait_side_launch), beside the rest of the chain
        hipEvent_t done = reinterpret_cast<hipEvent_t>(sitk_overlap_next_done_(overlap));
        if (done && hipEventRecord(done, side) != hipSuccess) {
          set_error("encoder_bwd_overlap: event record on the side stream failed");
          return SITK_ERR_LAUNCH;
        }
      }
    }
    else if (S.wg_batch) wg_all.insert(wg_all.end(), wg, wg + 4);       // launched once, after the slice's last layer
    else { SITK_TRY(sitk_gemm_wgrad_group_ws(wg, 4, dt, S.wgrad_ws, S.wgrad_ws_bytes, stream)); SITK_MARK("wgrad"); }
    float* part1 = S.ln_partials + (size_t)(2 * l) * S.ln_partial_floats;
    if (qkv_fused(c) && l > layer_begin && mlp_fused(c) && sitk_ln_gemm_mlp_bwd_supported(R, D, 3 * I, M, dt)) {
      // d to_qkv + LayerNorm backward of THIS layer and the MLP backward of the NEXT one (l - 1) in one launch: same 96-row
      // workgroups, the second half reads back the rows its own workgroup has just written (dx, dxc_next)
      // ALIASING, intended: `dres` of the first half (the residual gradient d x_mid of layer l, read) and `dx_mid` of the second
      // half (d x_mid of layer l - 1, written) are BOTH S.dxB, and the second half's `dy` / `dy_c` are the `dx` / `dxc_next` the
      // first half has just stored.  This is safe because the two halves give a workgroup the SAME 96 rows: a workgroup reads
      // its rows of S.dxB in the first half, stores its rows of dx / dxc_next, drains them (`s_waitcnt vmcnt(0)` + the workgroup
      // barrier between the halves), and only then overwrites its own rows of S.dxB; no workgroup touches another's rows.  A
      // change of either half's row partition breaks this (test_ln_gemm_mlp_bwd_pair_launch_is_bitwise_the_two_launches).
      const LayerActs& an = L.layers[l - 1];
      const int sn = slot(l - 1);
      float* part2n = S.ln_partials + (size_t)(2 * (l - 1) + 1) * S.ln_partial_floats;
      SITK_TRY(sitk_ln_gemm_mlp_bwd(dqkv, a.wqkv_t, xl, a.mean1, a.rstd1, P[l].ln1_w, S.dxB, dx, dxc_next, part1, 3 * I, an.xmid,
                                    an.mean2, an.rstd2, P[l - 1].ln2_w, an.w2_t, an.w1_t, an.u, S.du[sn], S.dxB, S.dxBc[sn], part2n, R,
                                    D, M, dt, stream));
      SITK_MARK("ln_gemm_mlp_bwd");
      mlp_bwd_done = true;
      ln_entries.push_back(LnFinalizeEntry{part1, G[l].ln1_w, G[l].ln1_b, (int)(sitk_ln_gemm_bwd_partial_floats(R) / (2 * D))});
    } else if (qkv_fused(c)) {
      SITK_TRY(sitk_ln_gemm_bwd(dqkv, a.wqkv_t, xl, a.mean1, a.rstd1, P[l].ln1_w, S.dxB, dx, dxc_next, part1, R, D, 3 * I, dt,
                                stream));
      SITK_MARK("ln_gemm_bwd");
      ln_entries.push_back(LnFinalizeEntry{part1, G[l].ln1_w, G[l].ln1_b, (int)(sitk_ln_gemm_bwd_partial_floats(R) / (2 * D))});
    } else {
      sitk_gemm_desc d4 = gemm_desc(R, D, 3 * I, dqkv, 3 * I, 0, a.wqkv_t, SITK_EPI_STORE, S.dh, D, 0);
      SITK_TRY(sitk_gemm_nt(&d4, dt, stream));
      SITK_MARK("gemm:dqkv");
      SITK_TRY(layernorm_bwd_deferred(S.dh, xl, a.mean1, a.rstd1, P[l].ln1_w, S.dxB, dx, dxc_next, part1, R, D, dt, hs));
      SITK_MARK("layernorm_bwd");
      ln_entries.push_back(LnFinalizeEntry{part1, G[l].ln1_w, G[l].ln1_b, 0});
    }
  }
  hipEvent_t ev_chain = nullptr;
  const bool use_side = overlap != nullptr;     // (the caller of a caller_joins object relies on the fork below, side launches or not)
  if (use_side) {            // the chain is complete here: everything behind this point may run beside the tail launch
    ev_chain = reinterpret_cast<hipEvent_t>(sitk_overlap_event_(overlap, sitk_overlap_max_layers_(overlap)));
    if (hipEventRecord(ev_chain, hs) != hipSuccess) { set_error("encoder_bwd_overlap: event record failed"); return SITK_ERR_LAUNCH; }
  }
  // The weight (+ bias) gradients of every layer of the slice in ONE launch: nothing downstream of a layer reads
  // its parameter gradients, and with 21 tiles per layer a whole slice brings enough tiles to give each workgroup
  // a long token run (12 layers: 252 tiles = one tile over ALL tokens per workgroup -- no token split, one slab
  // write and one reduction per step instead of twelve).  The operands stayed in their per-layer slots.
  if (take_embed) {
    wg_all.push_back(emb);
    take_embed = sitk_gemm_wgrad_group_ws_bytes(wg_all.data(), (int)wg_all.size(), dt) <= S.wgrad_ws_bytes;
    if (!take_embed) wg_all.pop_back();          // (cannot happen with make_layout's sizing; the caller then runs it)
    else *embed_done = 1;
  }
  // caller-supplied problems (to_original of the MPP head): taken when the slice ends at layer 0, all of them take the
  // large-tile path and the slab still has room (make_layout sizes it for the encoder's own tiles + 64 more)
  if (extra && n_extra > 0 && extra_done && layer_begin == 0 && S.wg_batch && (int)wg_all.size() + n_extra <= 52) {
    const size_t before = wg_all.size();
    wg_all.insert(wg_all.end(), extra, extra + n_extra);
    if (sitk_gemm_wgrad_group_ws_bytes(extra, n_extra, dt) > 0 &&
        sitk_gemm_wgrad_group_ws_bytes(wg_all.data(), (int)wg_all.size(), dt) <= S.wgrad_ws_bytes)
      *extra_done = 1;
    else
      wg_all.resize(before);
  }
  if (S.wg_batch && !wg_all.empty()) {
    // (make_layout sizes the slab for every slice length; a launch that does not fit would fall back to the generic tiles
    // without a word -- round 6 found SiT-base's three-slice data-parallel step doing exactly that, 25 % slower)
    const size_t need = sitk_gemm_wgrad_group_ws_bytes(wg_all.data(), (int)wg_all.size(), dt);
    SITK_REQUIRE(need == 0 || need <= S.wgrad_ws_bytes, "encoder_bwd: weight-gradient slab of %zu bytes, slice [%d, %d) needs %zu",
                 S.wgrad_ws_bytes, layer_begin, layer_end, need);
    // (a data-parallel caller leaves the all-reduce channels' CUs out of the tail launch: sitk_overlap_set_tail_cus)
    SITK_TRY(sitk_gemm_wgrad_group_ws_cus(wg_all.data(), (int)wg_all.size(), dt, S.wgrad_ws, S.wgrad_ws_bytes,
                                          overlap ? sitk_overlap_tail_cus_(overlap) : 256, stream));
    SITK_MARK("wgrad");
  }
  // every LayerNorm parameter gradient of the slice in one reduction launch; with a side stream it runs there, behind the
  // chain's last kernel (event ev_chain, recorded in front of the tail weight-gradient launch) and beside that launch
  if (use_side) {
    if (hipStreamWaitEvent(side, ev_chain, 0) != hipSuccess) { set_error("encoder_bwd_overlap: fork of the side stream failed"); return SITK_ERR_LAUNCH; }
    SITK_TRY(layernorm_finalize_multi(ln_entries.data(), (int)ln_entries.size(), R, D, side));
    if (!sitk_overlap_caller_joins_(overlap)) {   // join: whatever follows on the caller's stream sees the side stream's gradients
      hipEvent_t ev = reinterpret_cast<hipEvent_t>(sitk_overlap_event_(overlap, sitk_overlap_max_layers_(overlap)));
      if (hipEventRecord(ev, side) != hipSuccess || hipStreamWaitEvent(hs, ev, 0) != hipSuccess) {
        set_error("encoder_bwd_overlap: join of the side stream failed");
        return SITK_ERR_LAUNCH;
      }
    }
  } else {
    SITK_TRY(layernorm_finalize_multi(ln_entries.data(), (int)ln_entries.size(), R, D, hs));
    SITK_MARK("ln_finalize");
  }
  return SITK_OK;
}
