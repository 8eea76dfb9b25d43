// sitk fused multi-head self-attention (dim_head = 64) for gfx950, forward and backward.
// softmax((q k^T) * scale) v without materialising the (B, H, N, N) score matrix
// (vit_pytorch.vit.Attention as used by models/sit.py:57,76 and models/mpp.py:128).
//
// All three kernels are built from two MFMA products over 64-row LDS tiles ([rows][64] of T):
//   row_mma : S[t](16 tile rows x 16 lanes) = sum_d tile[16t+i][d] * frag(lane)[d]      (row reads)
//   tr_mma64: O[dt][d=16dt+i][lane] += sum_r tile[r][16dt+i] * P[r][lane]                (transposed reads)
// where P is the accumulator of a row_mma (lane l holds rows 16t + 4*(l>>4) + jj of column l&15):
// the accumulator is reused in registers as the next product's B operand, the LDS tile supplies the
// A operand through ds_read_b64_tr_b16 (bf16) or ds_read_b32 (f32), so the query (forward, dQ) or
// key (dK/dV) index stays on lane&15 through the whole kernel and per-row softmax state is per-lane.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace sitk {

template <typename T>
struct AttnGeom {
  static constexpr int RB = 64 * (int)sizeof(T);  // bytes per head row: 128 (bf16) / 256 (f32)
  static constexpr int KS = RB / 64;              // mma steps across dim_head
  static constexpr int NPAN = RB / 128;           // 128-byte panels per row
  static constexpr int CPR = RB / 16;             // 16-byte chunks per row
  static constexpr int CH = 64 * CPR / 256;       // chunks per thread per 64-row tile
  static constexpr int EPV = 16 / (int)sizeof(T);
  static constexpr int TILE_BYTES = 64 * RB;
};

// Swizzle of the bf16 attention tiles (tiled and sequence-resident kernels).  The transposed reads below take 4-row blocks that are FOUR rows apart
// in the two 16-lane groups of a half wave (rows 4 fq + q: the accumulator layout of the probabilities), for which
// the generic key of common.h (row bits 1 and 3, built for blocks eight rows apart) leaves rows r and r + 4 on the
// same banks: a 2-way conflict on every ds_read_b64_tr_b16 (SQ_LDS_BANK_CONFLICT = 27 % of the LDS cycles of the
// backward kernels, profiles/r01_pmc_attention_bwd.txt).  Row bits 1 and 2 give the 8 rows of a half-wave read 8
// distinct (parity, window) slots and keep the 16-row ds_read_b128 pattern conflict-free.
SITK_DEV int attn_res_key(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 1); }
SITK_DEV int attn_res_off(int row, int byte_in_row) { return row * 128 + (byte_in_row ^ (attn_res_key(row) << 5)); }

// byte offset of (row, byte) inside one 64-row x 128-byte panel: bf16 panels use the attention swizzle above, the f32
// panels (ds_read_b32 transposed reads, a different access pattern) the generic one of common.h
template <typename T>
SITK_DEV int tile_off(int row, int byte_in_row) {
  if constexpr (sizeof(T) == 2) return attn_res_off(row, byte_in_row);
  else return lds_off(row, byte_in_row);
}

// stage 64 rows x 64 elements (row r <- src + r*ld, zero when row0 + r >= nrows) into an LDS tile
template <typename T>
SITK_DEV void stage_tile(char* tile, const T* __restrict__ src, size_t ld, int row0, int nrows, int tid) {
  using G = AttnGeom<T>;
#pragma unroll
  for (int i = 0; i < G::CH; ++i) {
    const int c = tid + 256 * i, row = c / G::CPR, cc = c % G::CPR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row0 + row < nrows) v = *reinterpret_cast<const u32x4*>(src + (size_t)(row0 + row) * ld + cc * G::EPV);
    *reinterpret_cast<u32x4*>(tile + ((cc * 16) / 128) * (64 * 128) + tile_off<T>(row, (cc * 16) % 128)) = v;
  }
}

template <typename T>
SITK_DEV u32x4 row_frag(const char* tile, int row, int ks, int fq) {
  return *reinterpret_cast<const u32x4*>(tile + (ks >> 1) * (64 * 128) + tile_off<T>(row, (ks & 1) * 64 + fq * 16));
}

// s[t] += tile rows (16t + i) . frag       (frag: this lane's 16-byte chunks of its own row)
template <typename T>
SITK_DEV void row_mma(f32x4 (&s)[4], const char* tile, const u32x4 (&frag)[AttnGeom<T>::KS], int lane) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < AttnGeom<T>::KS; ++ks)
      s[t] = Mma<T>::mma(row_frag<T>(tile, 16 * t + fr, ks, fq), frag[ks], s[t]);
}

template <typename T>
struct TrMma;

template <>
struct TrMma<h16> {
  static SITK_DEV void run(f32x4 (&o)[4], const f32x4 (&p)[4], const char* tile, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      h16x8 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pb[e] = (h16)p[2 * s2][e]; pb[e + 4] = (h16)p[2 * s2 + 1][e]; }
      const u32x4 pf = __builtin_bit_cast(u32x4, pb);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int row = 32 * s2 + 4 * g + q, cb = (16 * dt + 4 * pp) * 2;
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) i16x4*)(tile + attn_res_off(row, cb)));
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) i16x4*)(tile + attn_res_off(row + 16, cb)));
        u32x4 vf;
        vf[0] = __builtin_bit_cast(u32x2, lo)[0];
        vf[1] = __builtin_bit_cast(u32x2, lo)[1];
        vf[2] = __builtin_bit_cast(u32x2, hi)[0];
        vf[3] = __builtin_bit_cast(u32x2, hi)[1];
        o[dt] = Mma<h16>::mma(vf, pf, o[dt]);
      }
    }
  }
};

template <>
struct TrMma<float> {
  static SITK_DEV void run(f32x4 (&o)[4], const f32x4 (&p)[4], const char* tile, int lane) {
    const int g = lane >> 4, fr = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int row = 16 * t + 4 * g + jj;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int col = 16 * dt + fr;  // 32 floats per panel
          const float a = *reinterpret_cast<const float*>(tile + (col >> 5) * (64 * 128) + lds_off(row, (col & 31) * 4));
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[t][jj], o[dt], 0, 0, 0);
        }
      }
  }
};

SITK_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// Reductions over the 4 lanes that share lane&15 (lanes l, l^16, l^32, l^48) with the gfx950 half/row
// swaps instead of ds_bpermute: v_permlane16_swap(v, v) leaves {row0,row0,row2,row2} / {row1,row1,row3,row3},
// v_permlane32_swap(v, v) leaves {lo,lo} / {hi,hi}; combining the two results pairs every lane with its
// xor-16 / xor-32 partner.  Pure VALU: no LDS round trip on the softmax critical path.
// Written as inline asm: with hipcc 7.2 the __builtin_amdgcn_permlane{16,32}_swap forms were folded
// incorrectly once both inputs carry the same value (checked on hardware: 200/256 lanes wrong in a
// 4-iteration loop, while the asm form below is exact).  The instruction modifies BOTH registers in
// place; s_nop 1 covers the VALU-write -> permlane-read hazard on either side.
SITK_DEV void permlane16_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
SITK_DEV void permlane32_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
SITK_DEV float xor_max4(float v) {
  float a = v, b = v;
  permlane16_swap(a, b);
  a = b = fmaxf(a, b);
  permlane32_swap(a, b);
  return fmaxf(a, b);
}
SITK_DEV float xor_sum4(float v) {
  float a = v, b = v;
  permlane16_swap(a, b);
  a = b = a + b;
  permlane32_swap(a, b);
  return a + b;
}

// 1-D grid, XCD-aware: the ceil(N/64) blocks of one (batch, head) are consecutive logical ids, so
// they run on one XCD and its L2 serves their shared K/V (or Q/dO) tiles.
struct BlockCoord {
  int x, h, b;
};
SITK_DEV BlockCoord attn_block(int N, int H) {
  const int nb = (N + 63) / 64;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  return BlockCoord{L % nb, (L / nb) % H, L / (nb * H)};
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ------------------------------------------------------------------------------------------
// forward: grid (ceil(N/64), H, B); wave w owns query rows 64*bx + 16w + (lane&15)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                       float* __restrict__ lse, int N, int H, float scale) {
  using G = AttnGeom<T>;
  __shared__ __attribute__((aligned(256))) char smem[2 * G::TILE_BYTES];
  char* sK = smem;
  char* sV = smem + G::TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const BlockCoord bc = attn_block(N, H);
  const int h = bc.h, b = bc.b, I = H * 64;
  const size_t ld = (size_t)3 * I;
  const int q = bc.x * 64 + 16 * wave + fr, qc = min(q, N - 1);
  const T* base = qkv + (size_t)b * N * ld;

  u32x4 qf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
    qf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + (ks * 64 + fq * 16) / (int)sizeof(T));

  const float c = scale * kLog2e;
  float m = -1e30f, l = 0.f;
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < N; k0 += 64) {
    __syncthreads();
    stage_tile<T>(sK, base + I + h * 64, ld, k0, N, tid);
    stage_tile<T>(sV, base + 2 * I + h * 64, ld, k0, N, tid);
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    row_mma<T>(s, sK, qf, lane);
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int key = k0 + 16 * t + 4 * fq + jj;
        s[t][jj] = key < N ? s[t][jj] * c : -INFINITY;
        mx = fmaxf(mx, s[t][jj]);
      }
    mx = xor_max4(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = fast_exp2(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float pv = fast_exp2(s[t][jj] - mn);
        s[t][jj] = pv;
        ps += pv;
      }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
    TrMma<T>::run(oacc, s, sV, lane);
  }
  const float lt = xor_sum4(l);
  const float inv = 1.0f / lt;
  if (q < N) {
    T* orow = o + ((size_t)b * N + q) * I + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt + 4 * fq, oacc[dt] * inv);
    if (fq == 0) lse[((size_t)b * H + h) * N + q] = (m + __log2f(lt)) * kLn2;
  }
}

// ------------------------------------------------------------------------------------------
// backward, query side: dQ (and delta = rowsum(dO * O), written for the key-side kernel).
// Same decomposition as forward.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                          const T* __restrict__ d_o, const float* __restrict__ lse,
                                                          float* __restrict__ delta, T* __restrict__ dqkv, int N,
                                                          int H, float scale) {
  using G = AttnGeom<T>;
  __shared__ __attribute__((aligned(256))) char smem[2 * G::TILE_BYTES];
  char* sK = smem;
  char* sV = smem + G::TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const BlockCoord bc = attn_block(N, H);
  const int h = bc.h, b = bc.b, I = H * 64;
  const size_t ld = (size_t)3 * I;
  const int q = bc.x * 64 + 16 * wave + fr, qc = min(q, N - 1);
  const T* base = qkv + (size_t)b * N * ld;

  u32x4 qf[G::KS], dof[G::KS];
  float dpart = 0.f;
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    const int eo = (ks * 64 + fq * 16) / (int)sizeof(T);
    qf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + eo);
    const T* dop = d_o + ((size_t)b * N + qc) * I + h * 64 + eo;
    const T* op = o + ((size_t)b * N + qc) * I + h * 64 + eo;
    dof[ks] = *reinterpret_cast<const u32x4*>(dop);
#pragma unroll
    for (int e = 0; e < G::EPV; ++e) dpart += to_f32(dop[e]) * to_f32(op[e]);
  }
  const float dl = xor_sum4(dpart);
  const size_t ridx = ((size_t)b * H + h) * N + qc;
  if (q < N && fq == 0) delta[ridx] = dl;
  const float Lq = lse[ridx] * kLog2e;
  const float c = scale * kLog2e;

  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < N; k0 += 64) {
    __syncthreads();
    stage_tile<T>(sK, base + I + h * 64, ld, k0, N, tid);
    stage_tile<T>(sV, base + 2 * I + h * 64, ld, k0, N, tid);
    __syncthreads();
    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    row_mma<T>(s, sK, qf, lane);
    row_mma<T>(dp, sV, dof, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int key = k0 + 16 * t + 4 * fq + jj;
        const float pv = key < N ? fast_exp2(s[t][jj] * c - Lq) : 0.f;
        s[t][jj] = pv * (dp[t][jj] - dl) * scale;
      }
    TrMma<T>::run(dq, s, sK, lane);
  }
  if (q < N) {
    T* row = dqkv + ((size_t)b * N + q) * ld + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(row + 16 * dt + 4 * fq, dq[dt]);
  }
}

// ------------------------------------------------------------------------------------------
// backward, key side: dK, dV.  grid (ceil(N/64), H, B); wave w owns keys 64*bx + 16w + (lane&15)
// and sweeps all queries in 64-row stages (Q and dO tiles in LDS, K and V fragments in registers).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dqkv, int N, int H, float scale) {
  using G = AttnGeom<T>;
  __shared__ __attribute__((aligned(256))) char smem[2 * G::TILE_BYTES + 2 * 64 * 4];
  char* sQ = smem;
  char* sDO = smem + G::TILE_BYTES;
  float* sL = reinterpret_cast<float*>(smem + 2 * G::TILE_BYTES);
  float* sD = sL + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const BlockCoord bc = attn_block(N, H);
  const int h = bc.h, b = bc.b, I = H * 64;
  const size_t ld = (size_t)3 * I;
  const int key = bc.x * 64 + 16 * wave + fr, kc = min(key, N - 1);
  const T* base = qkv + (size_t)b * N * ld;
  const T* dobase = d_o + (size_t)b * N * I;

  u32x4 kf[G::KS], vf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    const int eo = (ks * 64 + fq * 16) / (int)sizeof(T);
    kf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)kc * ld + I + h * 64 + eo);
    vf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)kc * ld + 2 * I + h * 64 + eo);
  }
  const float c = scale * kLog2e;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int q0 = 0; q0 < N; q0 += 64) {
    __syncthreads();
    stage_tile<T>(sQ, base + h * 64, ld, q0, N, tid);
    stage_tile<T>(sDO, dobase + h * 64, (size_t)I, q0, N, tid);
    if (tid < 64) {
      const int qq = q0 + tid;
      const size_t ridx = ((size_t)b * H + h) * N + min(qq, N - 1);
      sL[tid] = qq < N ? lse[ridx] * kLog2e : INFINITY;
      sD[tid] = qq < N ? delta[ridx] : 0.f;
    }
    __syncthreads();
    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    row_mma<T>(s, sQ, kf, lane);
    row_mma<T>(dp, sDO, vf, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 Lr = *reinterpret_cast<const f32x4*>(sL + 16 * t + 4 * fq);
      const f32x4 Dr = *reinterpret_cast<const f32x4*>(sD + 16 * t + 4 * fq);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float pv = fast_exp2(s[t][jj] * c - Lr[jj]);
        s[t][jj] = pv;
        dp[t][jj] = pv * (dp[t][jj] - Dr[jj]) * scale;
      }
    }
    TrMma<T>::run(dv, s, sDO, lane);
    TrMma<T>::run(dk, dp, sQ, lane);
  }
  if (key < N) {
    T* row = dqkv + ((size_t)b * N + key) * ld + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4(row + I + 16 * dt + 4 * fq, dk[dt]);
      store4(row + 2 * I + 16 * dt + 4 * fq, dv[dt]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Sequence-resident variants (bf16, N <= 384: the 81- and 321-token configurations of the reference).
// K and V of one (batch, head) are 2 * 384 * 128 B = 96 KB: the whole pair is brought into LDS ONCE
// by LDS-DMA (one workgroup of 8 waves per (batch, head)), then every wave walks its 16-row query
// (or key) tiles over all of it with no further barrier or staging.  Removes the 6x K/V re-reads, the
// 12 barriers and the exposed global-load latency per tile of the tiled kernels above.
// ------------------------------------------------------------------------------------------
constexpr int RES_MAX_N = 384;
__device__ u32x4 g_zero_page_attn[4];

// Per-lane byte offsets into a 64-row bf16 tile, computed ONCE per kernel: with them every fragment
// read is `tile + lane offset + compile-time immediate` (the XOR swizzle of lds_off() depends only on
// lane bits here; recomputing it per read cost more VALU than the softmax itself).
struct LaneOffs {
  int row[2];  // row-read (ds_read_b128) offset of k-step ks for row (lane&15):  + t * 2048 per 16-row block
  int tr[4];   // transposed-read offset of column block dt for row 4*(lane>>4) + ((lane>>2)&3): + s2*4096, + 2048 (second half)
};
SITK_DEV LaneOffs lane_offs_h16(int lane) {
  LaneOffs o;
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) o.row[ks] = attn_res_off(fr, ks * 64 + fq * 16);
  const int r = 4 * fq + ((lane >> 2) & 3);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o.tr[dt] = attn_res_off(r, (16 * dt + 4 * (lane & 3)) * 2);
  return o;
}
// NT: 16-row blocks of the tile that hold rows (4 = whole tile; the last tile of N = 64 k + 1 tokens holds one)
template <int NT = 4>
SITK_DEV void row_mma_o(f32x4 (&s)[4], const char* tile, const u32x4 (&frag)[2], const LaneOffs& o) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      s[t] = Mma<h16>::mma(*reinterpret_cast<const u32x4*>(tile + o.row[ks] + t * 2048), frag[ks], s[t]);
}
template <int NS2 = 2>
SITK_DEV void tr_mma_o(f32x4 (&acc)[4], const f32x4 (&p)[4], const char* tile, const LaneOffs& o) {
#pragma unroll
  for (int s2 = 0; s2 < NS2; ++s2) {
    h16x8 pb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pb[e] = (h16)p[2 * s2][e]; pb[e + 4] = (h16)p[2 * s2 + 1][e]; }
    const u32x4 pf = __builtin_bit_cast(u32x4, pb);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) i16x4*)(tile + o.tr[dt] + s2 * 4096));
      const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) i16x4*)(tile + o.tr[dt] + s2 * 4096 + 2048));
      u32x4 vf;
      vf[0] = __builtin_bit_cast(u32x2, lo)[0];
      vf[1] = __builtin_bit_cast(u32x2, lo)[1];
      vf[2] = __builtin_bit_cast(u32x2, hi)[0];
      vf[3] = __builtin_bit_cast(u32x2, hi)[1];
      acc[dt] = Mma<h16>::mma(vf, pf, acc[dt]);
    }
  }
}

// The same with the row sums of P taken by the matrix pipe: one more MFMA per 32-key half whose A operand is all ones
// (bf16 1.0), so lsum[.] += sum_k P[k][lane & 15] in every register of lsum -- no VALU adds, no cross-lane reduction,
// and the sum is taken over the bf16-rounded P that multiplies V.
#ifdef SITK_TU_F16
constexpr uint32_t kOnesH16x2 = 0x3C003C00u;   // two f16 ones
#else
constexpr uint32_t kOnesH16x2 = 0x3F803F80u;   // two bf16 ones
#endif
template <int NS2 = 2>
SITK_DEV void tr_mma_o_sum(f32x4 (&acc)[4], f32x4& lsum, const f32x4 (&p)[4], const char* tile, const LaneOffs& o) {
  const u32x4 ones = {kOnesH16x2, kOnesH16x2, kOnesH16x2, kOnesH16x2};
#pragma unroll
  for (int s2 = 0; s2 < NS2; ++s2) {
    h16x8 pb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pb[e] = (h16)p[2 * s2][e]; pb[e + 4] = (h16)p[2 * s2 + 1][e]; }
    const u32x4 pf = __builtin_bit_cast(u32x4, pb);
    lsum = Mma<h16>::mma(ones, pf, lsum);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) i16x4*)(tile + o.tr[dt] + s2 * 4096));
      const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) i16x4*)(tile + o.tr[dt] + s2 * 4096 + 2048));
      u32x4 vf;
      vf[0] = __builtin_bit_cast(u32x2, lo)[0];
      vf[1] = __builtin_bit_cast(u32x2, lo)[1];
      vf[2] = __builtin_bit_cast(u32x2, hi)[0];
      vf[3] = __builtin_bit_cast(u32x2, hi)[1];
      acc[dt] = Mma<h16>::mma(vf, pf, acc[dt]);
    }
  }
}

// one operand fragment (8 bf16) times a scalar, one rounding: the softmax scale (times log2 e) is folded into the Q (or K)
// fragments a wave keeps in registers, so the score MFMAs deliver log2-domain scores and the elementwise part starts at exp2
SITK_DEV u32x4 scale_frag(u32x4 f, float c) {
  h16x8 v = __builtin_bit_cast(h16x8, f);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * c);
  return __builtin_bit_cast(u32x4, v);
}
SITK_DEV f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
SITK_DEV u32x4 pack_pair_h16(const f32x4& a, const f32x4& b) {
  h16x8 pb;
#pragma unroll
  for (int e = 0; e < 4; ++e) { pb[e] = (h16)a[e]; pb[e + 4] = (h16)b[e]; }
  return __builtin_bit_cast(u32x4, pb);
}
// Online softmax, VALU-lean form (the attention kernels are bound by vector-instruction ISSUE, profiles/README):
//   * scores arrive as s' = c q.k - m: the running maximum m is the INITIAL ACCUMULATOR of the score MFMAs;
//   * the maximum is re-examined per lane only (8 v_max3 + one vote); the cross-lane reduction, the rescale of O and l
//     and the shift of the tile's scores run only when some score exceeds m by more than kRescaleThr (log2 units): until
//     then probabilities may reach 2^kRescaleThr, harmless in f32 / bf16 (same exponent range);
//   * the first tile of a sweep (m unknown) takes that path unconditionally with a zero initial accumulator.
constexpr float kRescaleThr = 8.0f;

// rows [0, ntiles*64) x 64 bf16 columns of `src` (leading dim ld) -> LDS tiles [t][64][128 B]; rows >= nrows are zero
SITK_DEV void dma_pieces_h16(char* dst, const h16* __restrict__ src, size_t ld, int nrows, int npieces, int wave, int lane, int nwaves);
SITK_DEV void dma_rows_h16(char* dst, const h16* __restrict__ src, size_t ld, int nrows, int ntiles, int wave, int lane, int nwaves) {
  dma_pieces_h16(dst, src, ld, nrows, ntiles * 8, wave, lane, nwaves);
}
// the same for an image of `npieces` pieces of 8 rows (a compact image ends with the last 32-row pair that holds rows)
SITK_DEV void dma_pieces_h16(char* dst, const h16* __restrict__ src, size_t ld, int nrows, int npieces, int wave, int lane, int nwaves) {
  const char* zero = reinterpret_cast<const char*>(g_zero_page_attn);
  for (int q = wave; q < npieces; q += nwaves) {         // one piece = 8 rows x 128 B
    const int row = q * 8 + (lane >> 3), r64 = row & 63;
    const int chunk = (lane & 7) ^ (attn_res_key(r64) << 1);
    const char* g = row < nrows ? reinterpret_cast<const char*>(src + (size_t)row * ld + chunk * 8) : zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
  }
}

// the last tile of a sequence holds `rows` in 1..63 rows: run f with the number of 16-row blocks to compute (1, 2 or 4;
// N = 321 = 5 * 64 + 1 and N = 81 = 64 + 17 are the reference's shapes: without this a sixth / a quarter of the score
// work of every kernel is spent on padding)
template <typename F>
SITK_DEV void tail_dispatch(int rows, F&& f) {
  if (rows <= 16) f(std::integral_constant<int, 1>{});
  else if (rows <= 32) f(std::integral_constant<int, 2>{});
  else f(std::integral_constant<int, 4>{});
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_fwd_res_kernel(const h16* __restrict__ qkv, h16* __restrict__ o,
                                                           float* __restrict__ lse, int N, int H, float scale) {
  using T = h16;
  __shared__ __attribute__((aligned(256))) char smem[2 * (RES_MAX_N / 64) * 8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int h = blockIdx.x % H, b = blockIdx.x / H, I = H * 64, nkt = (N + 63) / 64;
  const size_t ld = (size_t)3 * I;
  const T* base = qkv + (size_t)b * N * ld;
  char* sK = smem;
  char* sV = smem + (RES_MAX_N / 64) * 8192;   // fixed distance: one address register serves both tiles
  dma_rows_h16(sK, base + I + h * 64, ld, N, nkt, wave, lane, WAVES);
  dma_rows_h16(sV, base + 2 * I + h * 64, ld, N, nkt, wave, lane, WAVES);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float c = scale * kLog2e;
  const LaneOffs lo = lane_offs_h16(lane);
  for (int qt = wave; qt * 16 < N; qt += WAVES) {
    const int q = qt * 16 + fr, qc = min(q, N - 1);
    u32x4 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[ks] = scale_frag(*reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + ks * 32 + fq * 8), c);
    float m = 0.f;
    f32x4 negm = splat4(0.f), lacc = splat4(0.f);
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nfull = N >> 6;   // key tiles without padding: no masking instructions in their body
    // masked = 0: full key tile; masked = NT > 0: the last, partly filled tile, NT of its 16-key blocks computed
    auto kv_tile = [&](int t, auto masked, auto first_c) {
      constexpr int MK = decltype(masked)::value, NT = MK ? MK : 4, NS2 = (NT + 1) / 2;
      constexpr bool FIRST = decltype(first_c)::value;
      f32x4 s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] = FIRST ? splat4(0.f) : negm;
      row_mma_o<NT>(s, sK + t * 8192, qf, lo);       // s' = c q.k - m
      float mxl = -INFINITY;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if constexpr (MK != 0) {
            if (t * 64 + 16 * i + 4 * fq + jj >= N) s[i][jj] = -INFINITY;
          }
          mxl = fmaxf(mxl, s[i][jj]);
        }
      if (FIRST || __any(mxl > kRescaleThr)) {
        const float mx = xor_max4(mxl);
        const float delta = FIRST ? mx : fmaxf(mx, 0.f);
        if constexpr (!FIRST) {
          const float alpha = fast_exp2(-delta);
          lacc *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
        }
        m += delta;
        negm = splat4(-m);
#pragma unroll
        for (int i = 0; i < NT; ++i) s[i] -= delta;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) s[i][jj] = i < NT ? fast_exp2(s[i][jj]) : 0.f;
      tr_mma_o_sum<NS2>(oacc, lacc, s, sV + t * 8192, lo);
    };
    constexpr std::integral_constant<int, 0> full{};
    if (nfull > 0) {
      kv_tile(0, full, std::true_type{});
      for (int t = 1; t < nfull; ++t) kv_tile(t, full, std::false_type{});
      if (nfull < nkt) tail_dispatch(N - 64 * nfull, [&](auto nt) { kv_tile(nfull, nt, std::false_type{}); });
    } else {
      tail_dispatch(N, [&](auto nt) { kv_tile(0, nt, std::true_type{}); });
    }
    const float lt = lacc[0];                      // every register / lane of a column holds the column's full sum
    const float inv = 1.0f / lt;
    if (q < N) {
      T* orow = o + ((size_t)b * N + q) * I + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt + 4 * fq, oacc[dt] * inv);
      if (fq == 0) lse[((size_t)b * H + h) * N + q] = (m + __log2f(lt)) * kLn2;
    }
  }
}

#ifdef SITK_AB
// (diagnostic build only: QT query tiles per wave in the resident forward -- measured slower, see the file's header)
#include "experimental/attn_qt.inc"
#endif

// FOLD (D = 192): the gradient of the attention output is computed here instead of by a GEMM launch of its own:
//   dO[q, 64 h + c] = sum_j dxmid[q, j] * Wo[j, 64 h + c]      (to_out backward, utils/utils.py:26 layout: wo_t = Wo^T, (I, D))
// The head's 64 rows of wo_t (24 KB) sit in LDS next to K and V; per 16-query tile the product is taken TRANSPOSED
// (A = wo_t rows, B = the lane's own dxmid row, 24 MFMAs), with the A rows of block ct taken in the order
// c = 32 (ct >> 1) + 8 (m >> 2) + 4 (ct & 1) + (m & 3): the accumulators of blocks 2 ks and 2 ks + 1 then ARE the
// lane's 16-byte fragment ks of its dO row (columns 32 ks + 8 fq .. + 8) -- no shuffle -- and go to `d_o_out` for the
// key-side kernel in the same form.
#ifdef SITK_AB
// diagnostic build: s_memtime stamps of the merged backward kernel's first workgroups (tools/res_stamps.py)
constexpr int RES_STAMP_WGS = 4, RES_STAMP_N = 12;
__device__ unsigned long long g_res_stamps[RES_STAMP_WGS][16][RES_STAMP_N];
#define RES_ST_PARAM , unsigned long long* st
#define RES_ST_ARG , st
#define RES_STAMP(i) do { if (st) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RES_ST_PARAM
#define RES_ST_ARG
#define RES_STAMP(i) do {} while (0)
#endif
constexpr int FOLD_D = 192, FOLD_KS = FOLD_D / 32;
constexpr int RES_DQ_SMEM = 2 * (RES_MAX_N / 64) * 8192, RES_DQ_SMEM_FOLD = RES_DQ_SMEM + 64 * FOLD_D * 2;
// QRES (round 5; merged FOLD kernel, 320 < N <= 352: the 321-token configurations of dim 192): the operand images are COMPACT
// -- 22 blocks of 16 rows = 45 056 B instead of six 64-row tiles = 49 152 B; the sixth tile of 321 tokens holds ONE row, and the
// transposed reads touch 32-row pairs -- so that K, V, the head's Wo^T slice AND Q fit the CU together:
//   [ statistics 3 072 | K 45 056 | Q 45 056 | V 45 056 | Wo^T 24 576 ] = 162 816 B of 163 840.
// The query side reads its q fragments from the Q image (no global q reads), the key side finds Q already there (no second
// transfer), reads its k fragments from the K image, which nothing overwrites, and its v fragments from registers (first tile,
// read before the barrier) or from the head of the V image (second tile: the waves' second tiles are key tiles 0 .. n2 - 1, and
// dO lands on the LAST 45 056 bytes of [V | Wo^T], so V's first 24 576 bytes = 192 keys survive).  What crosses HBM / L2 twice
// in the classic plan and once here: q (7.9 MB per launch at B = 64), k and v (15.8 MB).  Same arithmetic, same bits.
constexpr int QRES_NB = 22, QRES_IMG = QRES_NB * 2048, QRES_STATS = 2 * RES_MAX_N * 4, QRES_W = 64 * FOLD_D * 2;
constexpr int QRES_MAX_N = QRES_NB * 16, QRES_MIN_N = 321;
constexpr int QRES_SMEM = QRES_STATS + 3 * QRES_IMG + QRES_W;
static_assert(QRES_SMEM <= 160 * 1024, "QRES LDS plan");
static_assert((QRES_NB - 16) * 2048 <= QRES_W, "the second tiles' v rows (key tiles 0 .. QRES_NB - 17) must survive the dO image");
template <int WAVES, bool FOLD, bool QRES = false>
SITK_DEV void attn_bwd_dq_res_body(char* smem, int bid, const h16* __restrict__ qkv, const h16* __restrict__ o,
                                   const h16* __restrict__ d_o, const float* __restrict__ lse, float* __restrict__ delta,
                                   h16* __restrict__ dqkv, int N, int H, float scale, const h16* __restrict__ dxmid,
                                   const h16* __restrict__ wo_t, h16* __restrict__ d_o_out RES_ST_PARAM) {
  using T = h16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int h = bid % H, b = bid / H, I = H * 64, nkt = (N + 63) / 64;
  const size_t ld = (size_t)3 * I;
  const T* base = qkv + (size_t)b * N * ld;
  static_assert(!QRES || FOLD, "QRES is a plan of the FOLD kernel");
  char* sK = smem + (QRES ? QRES_STATS : 0);
  char* sV = QRES ? sK + 2 * QRES_IMG : smem + (RES_MAX_N / 64) * 8192;   // fixed distance: one address register serves both tiles
  char* sQ = sK + QRES_IMG;                        // QRES only
  if constexpr (QRES) {
    dma_pieces_h16(sK, base + I + h * 64, ld, N, 2 * QRES_NB, wave, lane, WAVES);
    dma_pieces_h16(sV, base + 2 * I + h * 64, ld, N, 2 * QRES_NB, wave, lane, WAVES);
    dma_pieces_h16(sQ, base + h * 64, ld, N, 2 * QRES_NB, wave, lane, WAVES);
  } else {
    dma_rows_h16(sK, base + I + h * 64, ld, N, nkt, wave, lane, WAVES);
    dma_rows_h16(sV, base + 2 * I + h * 64, ld, N, nkt, wave, lane, WAVES);
  }
  char* sW = QRES ? sV + QRES_IMG : smem + 2 * (RES_MAX_N / 64) * 8192;   // FOLD: [k-step panel 0..5][64 rows][64 B], 16-B slots XOR (row >> 2) & 3
  if constexpr (FOLD) {
    for (int pc = wave; pc < 4 * FOLD_KS; pc += WAVES) {     // one piece = 16 rows x 64 B of one panel
      const int panel = pc >> 2, row = (pc & 3) * 16 + (lane >> 2), kq = (lane & 3) ^ ((row >> 2) & 3);
      const h16* g = wo_t + (size_t)(h * 64 + row) * FOLD_D + panel * 32 + kq * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(sW + pc * 1024), 16, 0, 0);
    }
  }
  RES_STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RES_STAMP(2);
  const float c = scale * kLog2e;
  const LaneOffs lo = lane_offs_h16(lane);
  // FOLD: A-fragment offsets of blocks ct = 0 / 1 (+ 2048 for ct = 2 / 3, + 4096 per k-step): row c(ct, m), chunk fq
  int wofs[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int crow = 8 * (fr >> 2) + 4 * ct + (fr & 3);
    wofs[ct] = crow * 64 + ((fq ^ ((crow >> 2) & 3)) << 4);
  }
  for (int qt = wave; qt * 16 < N; qt += WAVES) {
    const int q = qt * 16 + fr, qc = min(q, N - 1);
    u32x4 qf[2], dof[2];
    float dpart = 0.f;
    if constexpr (FOLD) {
      u32x4 dxf[FOLD_KS];
#pragma unroll
      for (int ks = 0; ks < FOLD_KS; ++ks)
        dxf[ks] = *reinterpret_cast<const u32x4*>(dxmid + ((size_t)b * N + qc) * FOLD_D + ks * 32 + fq * 8);
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < FOLD_KS; ++ks)
          acc[ct] = Mma<h16>::mma(*reinterpret_cast<const u32x4*>(sW + wofs[ct & 1] + (ct >> 1) * 2048 + ks * 4096),
                                   dxf[ks], acc[ct]);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h16x8 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pk[e] = (h16)acc[2 * ks][e]; pk[e + 4] = (h16)acc[2 * ks + 1][e]; }
        dof[ks] = __builtin_bit_cast(u32x4, pk);
        const int eo = ks * 32 + fq * 8;
        if constexpr (QRES) qf[ks] = *reinterpret_cast<const u32x4*>(sQ + qt * 2048 + lo.row[ks]);   // (rows >= N: zeros, never stored)
        else qf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + eo);
        const h16x8 ov = *reinterpret_cast<const h16x8*>(o + ((size_t)b * N + qc) * I + h * 64 + eo);
#pragma unroll
        for (int e = 0; e < 8; ++e) dpart += (float)pk[e] * (float)ov[e];
        if (q < N) *reinterpret_cast<u32x4*>(d_o_out + ((size_t)b * N + q) * I + h * 64 + eo) = dof[ks];
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int eo = ks * 32 + fq * 8;
        qf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + eo);
        const T* dop = d_o + ((size_t)b * N + qc) * I + h * 64 + eo;
        const T* op = o + ((size_t)b * N + qc) * I + h * 64 + eo;
        dof[ks] = *reinterpret_cast<const u32x4*>(dop);
#pragma unroll
        for (int e = 0; e < 8; ++e) dpart += (float)dop[e] * (float)op[e];
      }
    }
    const float dl = xor_sum4(dpart);
    const size_t ridx = ((size_t)b * H + h) * N + qc;
    if (q < N && fq == 0) delta[ridx] = dl;
    const float nLq = -lse[ridx] * kLog2e, ndl = -dl;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = scale_frag(qf[ks], c);   // scores in log2 units straight from the MFMA
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nfull = N >> 6;
    auto kv_tile = [&](int t, auto masked) {
      constexpr int MK = decltype(masked)::value, NT = MK ? MK : 4, NS2 = (NT + 1) / 2;
      f32x4 s[4], dp[4];                               // row constants start the accumulators: s' = c q.k - lse, dp' = dO.v - delta
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[i] = i < NT ? splat4(nLq) : splat4(0.f); dp[i] = i < NT ? splat4(ndl) : splat4(0.f); }
      row_mma_o<NT>(s, sK + t * 8192, qf, lo);
      row_mma_o<NT>(dp, sV + t * 8192, dof, lo);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        f32x4 x;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          x[jj] = fast_exp2(s[i][jj]);
          if constexpr (MK != 0) {
            if (t * 64 + 16 * i + 4 * fq + jj >= N) x[jj] = 0.f;
          }
        }
        s[i] = x * dp[i];                              // dS / scale: the scale is applied once, to dQ
      }
      tr_mma_o<NS2>(dq, s, sK + t * 8192, lo);        // s[i >= NT] = 0
    };
    for (int t = 0; t < nfull; ++t) kv_tile(t, std::integral_constant<int, 0>{});
    if (nfull < nkt) tail_dispatch(N - 64 * nfull, [&](auto nt) { kv_tile(nfull, nt); });
    if (q < N) {
      T* row = dqkv + ((size_t)b * N + q) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(row + 16 * dt + 4 * fq, dq[dt] * scale);
    }
    RES_STAMP(qt < WAVES ? 3 : 4);
  }
}

template <int WAVES, bool FOLD = false>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dq_res_kernel(const h16* __restrict__ qkv, const h16* __restrict__ o,
                                                              const h16* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ delta, h16* __restrict__ dqkv, int N,
                                                              int H, float scale, const h16* __restrict__ dxmid = nullptr,
                                                              const h16* __restrict__ wo_t = nullptr,
                                                              h16* __restrict__ d_o_out = nullptr) {
  __shared__ __attribute__((aligned(256))) char smem[FOLD ? RES_DQ_SMEM_FOLD : RES_DQ_SMEM];
  // FOLD: the H workgroups of one sample all read that sample's dxmid rows -- keep them on one XCD (one L2)
  const int bid = FOLD ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
#ifdef SITK_AB
  unsigned long long* st = nullptr;
#endif
  attn_bwd_dq_res_body<WAVES, FOLD>(smem, bid, qkv, o, d_o, lse, delta, dqkv, N, H, scale, dxmid, wo_t, d_o_out RES_ST_ARG);
}

constexpr int RES_DKV_SMEM = 2 * (RES_MAX_N / 64) * 8192 + 2 * RES_MAX_N * 4;
// QRES (see above): Q is already in LDS, k fragments come from the K image, v fragments from `vpre` (the wave's first tile) or
// from the surviving head of the V image (its second tile); a wave's FIRST tile is key tile n2 + wave, its second tile `wave`
// (n2 = tiles - WAVES of them), so that the second tiles are the first n2.
template <int WAVES, bool QRES = false>
SITK_DEV void attn_bwd_dkv_res_body(char* smem, int bid, const h16* __restrict__ qkv, const h16* __restrict__ d_o,
                                    const float* __restrict__ lse, const float* __restrict__ delta, h16* __restrict__ dqkv,
                                    int N, int H, float scale, const u32x4 (&vpre)[2] RES_ST_PARAM) {
  using T = h16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int h = bid % H, b = bid / H, I = H * 64, nqt = (N + 63) / 64;
  const size_t ld = (size_t)3 * I;
  const T* base = qkv + (size_t)b * N * ld;
  // row statistics FIRST: every LDS read of the sweep then is `address register + 16-bit immediate`
  float* sL = reinterpret_cast<float*>(smem);
  float* sD = sL + RES_MAX_N;
  char* sQ = QRES ? smem + QRES_STATS + QRES_IMG : smem + 2 * RES_MAX_N * 4;
  char* sDO = QRES ? smem + QRES_STATS + 2 * QRES_IMG + QRES_W : sQ + (RES_MAX_N / 64) * 8192;    // fixed distance: one address register serves both tiles
  const char* sKf = smem + QRES_STATS;         // QRES: the K image of the query side, untouched
  const char* sVf = smem + QRES_STATS + 2 * QRES_IMG;   // QRES: the first QRES_W bytes of the V image (keys 0 .. 191) survive the dO image
  if constexpr (QRES) {
    dma_pieces_h16(sDO, d_o + (size_t)b * N * I + h * 64, (size_t)I, N, 2 * QRES_NB, wave, lane, WAVES);
  } else {
    dma_rows_h16(sQ, base + h * 64, ld, N, nqt, wave, lane, WAVES);
    dma_rows_h16(sDO, d_o + (size_t)b * N * I + h * 64, (size_t)I, N, nqt, wave, lane, WAVES);
  }
  for (int r = tid; r < nqt * 64; r += WAVES * 64) {
    const size_t ridx = ((size_t)b * H + h) * N + min(r, N - 1);
    sL[r] = r < N ? -lse[ridx] * kLog2e : -INFINITY;   // negated (added below); exp2(x - inf) = 0 for padded query rows
    sD[r] = r < N ? -delta[ridx] : 0.f;
  }
  RES_STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  RES_STAMP(7);
  const float c = scale * kLog2e;
  const LaneOffs lo = lane_offs_h16(lane);
  const int n16 = (N + 15) >> 4, n2 = QRES ? max(n16 - WAVES, 0) : 0;      // QRES: tiles of the second round = key tiles 0 .. n2 - 1
  for (int round = 0;; ++round) {
    int kt;
    if constexpr (QRES) {
      kt = round == 0 ? n2 + wave : wave;
      if (round > 1 || (round == 1 && wave >= n2) || kt >= n16) break;
    } else {
      kt = wave + round * WAVES;
      if (kt >= n16) break;
    }
    const int key = kt * 16 + fr, kc = min(key, N - 1);
    u32x4 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int eo = ks * 32 + fq * 8;
      if constexpr (QRES) {
        kf[ks] = scale_frag(*reinterpret_cast<const u32x4*>(sKf + kt * 2048 + lo.row[ks]), c);              // (rows >= N: zeros, never stored)
        vf[ks] = round == 0 ? vpre[ks] : *reinterpret_cast<const u32x4*>(sVf + kt * 2048 + lo.row[ks]);
      } else {
        kf[ks] = scale_frag(*reinterpret_cast<const u32x4*>(base + (size_t)kc * ld + I + h * 64 + eo), c);   // log2-domain scores
        vf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)kc * ld + 2 * I + h * 64 + eo);
      }
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int nfull = N >> 6;
    // nt = 0: full query tile; nt > 0: the last, partly filled tile (its padded rows have L = inf, so p = 0 there)
    auto q_tile = [&](int t, auto nt) {
      constexpr int NT = decltype(nt)::value ? decltype(nt)::value : 4, NS2 = (NT + 1) / 2;
      f32x4 s[4], dp[4];                               // the per-query statistics start the accumulators
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i] = i < NT ? *reinterpret_cast<const f32x4*>(sL + t * 64 + 16 * i + 4 * fq) : splat4(0.f);    // -lse log2 e
        dp[i] = i < NT ? *reinterpret_cast<const f32x4*>(sD + t * 64 + 16 * i + 4 * fq) : splat4(0.f);   // -delta
      }
      row_mma_o<NT>(s, sQ + t * 8192, kf, lo);
      row_mma_o<NT>(dp, sDO + t * 8192, vf, lo);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        f32x4 x;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) x[jj] = fast_exp2(s[i][jj]);
        s[i] = x;
        dp[i] = x * dp[i];                            // dS / scale: the scale is applied once, to dK
      }
      tr_mma_o<NS2>(dv, s, sDO + t * 8192, lo);       // s, dp [i >= NT] = 0
      tr_mma_o<NS2>(dk, dp, sQ + t * 8192, lo);
    };
    for (int t = 0; t < nfull; ++t) q_tile(t, std::integral_constant<int, 0>{});
    if (nfull < nqt) tail_dispatch(N - 64 * nfull, [&](auto nt) { q_tile(nfull, nt); });
    if (key < N) {
      T* row = dqkv + ((size_t)b * N + key) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store4(row + I + 16 * dt + 4 * fq, dk[dt] * scale);
        store4(row + 2 * I + 16 * dt + 4 * fq, dv[dt]);
      }
    }
    RES_STAMP(round == 0 && kt < WAVES + n2 ? 8 : 9);
  }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dkv_res_kernel(const h16* __restrict__ qkv, const h16* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               h16* __restrict__ dqkv, int N, int H, float scale) {
  __shared__ __attribute__((aligned(256))) char smem[RES_DKV_SMEM];
#ifdef SITK_AB
  unsigned long long* st = nullptr;
#endif
  const u32x4 none[2] = {};
  attn_bwd_dkv_res_body<WAVES>(smem, (int)blockIdx.x, qkv, d_o, lse, delta, dqkv, N, H, scale, none RES_ST_ARG);
}

// The two sides in ONE launch (round 4): a workgroup owns its (sample, head) through both -- the query side first (dQ, delta,
// and with FOLD the dO rows), then, behind a workgroup barrier, the key side on the dO / delta rows the same workgroup has
// just written (L2) and on q, k, v that the first half has just pulled through this XCD's L2.  One launch boundary per layer
// less, and q, k, v cross HBM once instead of twice (section 8, round 4).  The single-side kernels above stay for profiling
// (sitk_attention_bwd_phases with phases 1 or 2).
template <bool FOLD, bool QRES = false>
__global__ __launch_bounds__(1024) void attn_bwd_res_kernel(const h16* __restrict__ qkv, const h16* __restrict__ o,
                                                            const h16* __restrict__ d_o, const float* __restrict__ lse,
                                                            float* __restrict__ delta, h16* __restrict__ dqkv, int N, int H,
                                                            float scale, const h16* __restrict__ dxmid, const h16* __restrict__ wo_t,
                                                            h16* __restrict__ d_o_out) {
  constexpr int SMEM0 = (FOLD ? RES_DQ_SMEM_FOLD : RES_DQ_SMEM) > RES_DKV_SMEM ? (FOLD ? RES_DQ_SMEM_FOLD : RES_DQ_SMEM) : RES_DKV_SMEM;
  constexpr int SMEM = QRES ? QRES_SMEM : SMEM0;
  __shared__ __attribute__((aligned(256))) char smem[SMEM];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);      // the H workgroups of a sample on one XCD (they share its dxmid rows)
#ifdef SITK_AB
  unsigned long long stamps[RES_STAMP_N] = {}, *st = stamps;
  st[0] = __builtin_amdgcn_s_memtime();
#endif
  attn_bwd_dq_res_body<16, FOLD, QRES>(smem, bid, qkv, o, d_o, lse, delta, dqkv, N, H, scale, dxmid, wo_t, d_o_out RES_ST_ARG);
  u32x4 vpre[2] = {};
  if constexpr (QRES) {       // the v fragments of this wave's first key tile, before the dO image lands on those rows of V
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = (N + 15) >> 4, kt = max(n16 - 16, 0) + wave;
    if (kt < n16) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        vpre[ks] = *reinterpret_cast<const u32x4*>(smem + QRES_STATS + 2 * QRES_IMG + kt * 2048 + attn_res_off(lane & 15, ks * 64 + (lane >> 4) * 16));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's dO / delta stores have left the CU ...
  __syncthreads();                                       // ... and nobody reads the K / V image any more
  RES_STAMP(5);
  attn_bwd_dkv_res_body<16, QRES>(smem, bid, qkv, FOLD ? d_o_out : d_o, lse, delta, dqkv, N, H, scale, vpre RES_ST_ARG);
#ifdef SITK_AB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st[10] = __builtin_amdgcn_s_memtime();
  if (blockIdx.x >= gridDim.x / 2 && blockIdx.x < gridDim.x / 2 + RES_STAMP_WGS && (threadIdx.x & 63) == 0)
    for (int i = 0; i < RES_STAMP_N; ++i) g_res_stamps[blockIdx.x - gridDim.x / 2][threadIdx.x >> 6][i] = stamps[i];
#endif
}


#ifdef SITK_AB
// (diagnostic build only: the unit-packed kernels of round 4 -- an experiment that measured no gain -- and their launch branches)
#include "experimental/attn_pk.inc"
#endif

// ------------------------------------------------------------------------------------------
// Ring variants (bf16, 384 < N <= 2048: the 1281-token configurations of BASELINE configs 3 and 5).
// One workgroup = WAVES waves = 2 * WAVES consecutive 16-row query (or key) tiles of one (batch, head); every wave keeps
// TWO tiles in registers, so each fragment it reads from LDS feeds two MFMAs.  The other operand pair of the sweep
// (K and V tiles for forward / dQ; Q and dO tiles for dK / dV) streams through a 3-stage LDS ring by LDS-DMA, one
// barrier per tile.  Tile t + 2 is issued in iteration t right after the iteration's LAST LDS read that hipcc can see
// (the transposed fragment reads, which stay builtins so that their two halves are allocated as one MFMA operand):
// hipcc drains the DMA queue in front of compiler-visible LDS reads, so that read of iteration t + 1 is where tile
// t + 2 is waited for -- a whole iteration after its issue -- and tile t + 1, which iteration t + 1 computes on, was
// waited for an iteration earlier.  The row-fragment reads at the top of an iteration are inline asm (invisible to that
// logic; a visible read there would wait for a DMA issued moments before).  Blocks of one (batch, head) are consecutive
// logical ids (one XCD's L2 serves their shared tiles).  Launch geometry (4 waves x 2 tiles): ring_blocks() below.  The
// last, partly filled tile of the sweep runs a separate instantiation of the loop body (masking; zero page for the
// missing rows).
// ------------------------------------------------------------------------------------------
constexpr int RING_STAGES = 3, RING_STAGE = 16384, RING_MAX_N = 2048;

// 8 row fragments (4 sixteen-row blocks x 2 k-steps) of the 64-row tile at byte addresses a0 / a1 (k-step 0 / 1)
#define SITK_RING_ROWS8(F, A0, A1, OFF)                                                                                   \
  asm volatile("ds_read_b128 %0, %8 offset:" #OFF "\n\tds_read_b128 %1, %9 offset:" #OFF "\n\t"                           \
               "ds_read_b128 %2, %8 offset:" #OFF "+2048\n\tds_read_b128 %3, %9 offset:" #OFF "+2048\n\t"                 \
               "ds_read_b128 %4, %8 offset:" #OFF "+4096\n\tds_read_b128 %5, %9 offset:" #OFF "+4096\n\t"                 \
               "ds_read_b128 %6, %8 offset:" #OFF "+6144\n\tds_read_b128 %7, %9 offset:" #OFF "+6144\n\t"                 \
               "s_waitcnt lgkmcnt(0)"                                                                                     \
               : "=&v"(F[0][0]), "=&v"(F[0][1]), "=&v"(F[1][0]), "=&v"(F[1][1]), "=&v"(F[2][0]), "=&v"(F[2][1]),          \
                 "=&v"(F[3][0]), "=&v"(F[3][1])                                                                           \
               : "v"(A0), "v"(A1)                                                                                         \
               : "memory")

SITK_DEV u32x4 pack_pair(const f32x4& a, const f32x4& b) {
  h16x8 pb;
#pragma unroll
  for (int e = 0; e < 4; ++e) { pb[e] = (h16)a[e]; pb[e + 4] = (h16)b[e]; }
  return __builtin_bit_cast(u32x4, pb);
}
// transposed fragment (column block dt, row half s2) of the tile at `tile`: two compiler-visible ds_read_b64_tr_b16
SITK_DEV u32x4 ring_tr_frag(const char* tile, const LaneOffs& o, int s2, int dt) {
  const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(tile + o.tr[dt] + s2 * 4096));
  const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(tile + o.tr[dt] + s2 * 4096 + 2048));
  return u32x4{__builtin_bit_cast(u32x2, lo)[0], __builtin_bit_cast(u32x2, lo)[1], __builtin_bit_cast(u32x2, hi)[0],
               __builtin_bit_cast(u32x2, hi)[1]};
}

// The DMA of one stage: pieces 0..7 = rows of tile A, 8..15 = rows of tile B (8 rows x 128 B each), 64 rows from row
// t * 64.  Every wave issues exactly PPW instructions (piece index clamped: duplicates rewrite the same bytes).  Source =
// wave-uniform tile base + 32-bit lane offset; rows >= nrows (last tile only) come from a zero page.
template <int WAVES>
struct RingLoader {
  static constexpr int PPW = (16 + WAVES - 1) / WAVES;
  const char *base_a, *base_b;           // row 0 of the two operands for this (batch, head)
  size_t step_a, step_b;                 // bytes per 64 rows
  uint32_t off[PPW];                     // lane offset of piece i (bytes, from the tile's first row)
  int prow[PPW], pdst[PPW];
  bool is_b[PPW];
  SITK_DEV void init(const h16* src_a, size_t ld_a, const h16* src_b, size_t ld_b, int wave, int lane) {
    base_a = reinterpret_cast<const char*>(src_a);
    base_b = reinterpret_cast<const char*>(src_b);
    step_a = 128 * ld_a;
    step_b = 128 * ld_b;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = min(wave * PPW + i, 15);
      const int row = (p & 7) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (attn_res_key(row) << 1);
      is_b[i] = p >= 8;
      off[i] = (uint32_t)((size_t)row * (p >= 8 ? ld_b : ld_a) * 2 + chunk * 16);
      prow[i] = row;
      pdst[i] = p * 1024;
    }
  }
  template <bool TAIL>
  SITK_DEV void issue(char* smem, int t, int nrows) const {
    char* dst = smem + (t % RING_STAGES) * RING_STAGE;   // (scalar arithmetic)
    const char* ta = base_a + (size_t)t * step_a;
    const char* tb = base_b + (size_t)t * step_b;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const char* gp = (is_b[i] ? tb : ta) + off[i];
      if constexpr (TAIL) {
        if (t * 64 + prow[i] >= nrows) gp = reinterpret_cast<const char*>(g_zero_page_attn);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                       (__attribute__((address_space(3))) void*)(dst + pdst[i]), 16, 0, 0);
    }
  }
  SITK_DEV void issue_any(char* smem, int t, int nrows) const {
    if ((t + 1) * 64 > nrows) issue<true>(smem, t, nrows);
    else issue<false>(smem, t, nrows);
  }
};

struct RingBlock {
  int x, h, b;
};
SITK_DEV RingBlock ring_block(int nxb, int H) {
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  return RingBlock{L % nxb, (L / nxb) % H, L / (nxb * H)};
}

template <int WAVES, int QT, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void attn_fwd_ring_kernel(const h16* __restrict__ qkv, h16* __restrict__ o,
                                                                   float* __restrict__ lse, int N, int H, float scale, int nqb) {
  __shared__ __attribute__((aligned(256))) char smem[RING_STAGES * RING_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const RingBlock bc = ring_block(nqb, H);
  const int h = bc.h, b = bc.b, I = H * 64, nkt = (N + 63) / 64;
  const size_t ld = (size_t)3 * I;
  const h16* base = qkv + (size_t)b * N * ld;
  RingLoader<WAVES> dma;
  dma.init(base + I + h * 64, ld, base + 2 * I + h * 64, ld, wave, lane);
  dma.issue_any(smem, 0, N);
  if (nkt > 1) dma.issue_any(smem, 1, N);

  const int qt0 = (bc.x * WAVES + wave) * QT;             // first 16-row query tile of this wave
  const bool active = qt0 * 16 < N;                       // wave-uniform: padding waves only move tiles
  u32x4 qf[QT][2];
  int qrow[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    qrow[j] = (qt0 + j) * 16 + fr;
    const int qc = min(qrow[j], N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[j][ks] = scale_frag(*reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + ks * 32 + fq * 8), scale * kLog2e);
  }
  const LaneOffs lo = lane_offs_h16(lane);
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const u32x4 ones = {kOnesH16x2, kOnesH16x2, kOnesH16x2, kOnesH16x2};
  float m[QT];
  f32x4 negm[QT], lacc[QT];
  f32x4 oacc[QT][4];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    m[j] = 0.f; negm[j] = splat4(0.f); lacc[j] = splat4(0.f);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tiles 0 and 1 (and the Q fragments) have landed

  int ring_stage = 0;
  // tail_c: 0 = full key tile; NT > 0 = the last, partly filled tile, of which only NT 16-key blocks are computed
  auto body = [&](int t, auto tail_c, auto first_c) {
    constexpr int NT = decltype(tail_c)::value ? decltype(tail_c)::value : 4;
    constexpr bool TAIL = decltype(tail_c)::value != 0, FIRST = decltype(first_c)::value;
    __builtin_amdgcn_s_barrier();                         // every wave's pieces of tile t; stage (t + 2) % 3 no longer read
    const int stage = ring_stage;
    ring_stage = ring_stage == RING_STAGES - 1 ? 0 : ring_stage + 1;
    const char* tile = smem + stage * RING_STAGE;
    if (!active) {                                        // padding wave: moves its share of the tiles, computes nothing
      if (t + 2 < nkt) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dma.issue_any(smem, t + 2, N); }
      return;
    }
    const uint32_t sb = lbase + stage * RING_STAGE;
    u32x4 kf[4][2];
    SITK_RING_ROWS8(kf, sb + lo.row[0], sb + lo.row[1], 0);
    f32x4 s[QT][4];                                       // s' = c q.k - m: the running maximum starts the accumulators
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) s[j][i] = FIRST ? splat4(0.f) : negm[j];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < QT; ++j) s[j][i] = Mma<h16>::mma(kf[i][ks], qf[j][ks], s[j][i]);
    u32x4 pf[QT][2];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float mxl = -INFINITY;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if constexpr (TAIL) {
            if (t * 64 + 16 * i + 4 * fq + jj >= N) s[j][i][jj] = -INFINITY;
          }
          mxl = fmaxf(mxl, s[j][i][jj]);
        }
      if (FIRST || __any(mxl > kRescaleThr)) {            // see kRescaleThr: rare after the first tile
        const float mx = xor_max4(mxl);
        const float delta = FIRST ? mx : fmaxf(mx, 0.f);
        if constexpr (!FIRST) {
          const float alpha = fast_exp2(-delta);
          lacc[j] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[j][dt] *= alpha;
        }
        m[j] += delta;
        negm[j] = splat4(-m[j]);
#pragma unroll
        for (int i = 0; i < NT; ++i) s[j][i] -= delta;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) s[j][i][jj] = i < NT ? fast_exp2(s[j][i][jj]) : 0.f;
      pf[j][0] = pack_pair(s[j][0], s[j][1]);
      pf[j][1] = pack_pair(s[j][2], s[j][3]);
    }
    u32x4 vf[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[s2][dt] = ring_tr_frag(tile + 8192, lo, s2, dt);
    __builtin_amdgcn_sched_barrier(0);                    // the DMA issue stays behind the last LDS read of the iteration
    if (t + 2 < nkt) dma.issue_any(smem, t + 2, N);
#pragma unroll
    for (int s2 = 0; s2 < (NT + 1) / 2; ++s2) {
#pragma unroll
      for (int j = 0; j < QT; ++j) lacc[j] = Mma<h16>::mma(ones, pf[j][s2], lacc[j]);    // row sums of P on the matrix pipe
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < QT; ++j) oacc[j][dt] = Mma<h16>::mma(vf[s2][dt], pf[j][s2], oacc[j][dt]);
    }
  };
  const int nfull = N >> 6;                               // >= 1 (ring kernels take N >= 64)
  constexpr std::integral_constant<int, 0> full{};
  body(0, full, std::true_type{});
  for (int t = 1; t < nfull; ++t) body(t, full, std::false_type{});
  if (nfull < nkt) tail_dispatch(N - 64 * nfull, [&](auto nt) { body(nfull, nt, std::false_type{}); });
  if (!active) return;
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    const float lt = lacc[j][0];
    const float inv = 1.0f / lt;
    const int q = qrow[j];
    if (q < N) {
      h16* orow = o + ((size_t)b * N + q) * I + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt + 4 * fq, oacc[j][dt] * inv);
      if (fq == 0) lse[((size_t)b * H + h) * N + q] = (m[j] + __log2f(lt)) * kLn2;
    }
  }
}

// backward, query side: dQ, delta; same sweep as forward (K and V tiles in the ring), three products per tile
template <int WAVES, int QT, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void attn_bwd_dq_ring_kernel(const h16* __restrict__ qkv, const h16* __restrict__ o,
                                                                      const h16* __restrict__ d_o, const float* __restrict__ lse,
                                                                      float* __restrict__ delta, h16* __restrict__ dqkv, int N,
                                                                      int H, float scale, int nqb) {
  __shared__ __attribute__((aligned(256))) char smem[RING_STAGES * RING_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const RingBlock bc = ring_block(nqb, H);
  const int h = bc.h, b = bc.b, I = H * 64, nkt = (N + 63) / 64;
  const size_t ld = (size_t)3 * I;
  const h16* base = qkv + (size_t)b * N * ld;
  RingLoader<WAVES> dma;
  dma.init(base + I + h * 64, ld, base + 2 * I + h * 64, ld, wave, lane);
  dma.issue_any(smem, 0, N);
  if (nkt > 1) dma.issue_any(smem, 1, N);

  const int qt0 = (bc.x * WAVES + wave) * QT;
  const bool active = qt0 * 16 < N;
  u32x4 qf[QT][2], dof[QT][2];
  int qrow[QT];
  float Lq[QT], ndl[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    qrow[j] = (qt0 + j) * 16 + fr;
    const int qc = min(qrow[j], N - 1);
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int eo = ks * 32 + fq * 8;
      qf[j][ks] = *reinterpret_cast<const u32x4*>(base + (size_t)qc * ld + h * 64 + eo);
      const h16* dop = d_o + ((size_t)b * N + qc) * I + h * 64 + eo;
      const h16* op = o + ((size_t)b * N + qc) * I + h * 64 + eo;
      dof[j][ks] = *reinterpret_cast<const u32x4*>(dop);
      const h16x8 dv8 = __builtin_bit_cast(h16x8, dof[j][ks]);
      const h16x8 ov8 = *reinterpret_cast<const h16x8*>(op);
#pragma unroll
      for (int e = 0; e < 8; ++e) dpart += (float)dv8[e] * (float)ov8[e];
    }
    const float dl = xor_sum4(dpart);
    const size_t ridx = ((size_t)b * H + h) * N + qc;
    if (active && qrow[j] < N && fq == 0) delta[ridx] = dl;
    Lq[j] = -lse[ridx] * kLog2e;                          // (negated: the initial accumulator of the score product)
    ndl[j] = -dl;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[j][ks] = scale_frag(qf[j][ks], scale * kLog2e);
  }
  const LaneOffs lo = lane_offs_h16(lane);
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  f32x4 dq[QT][4];
#pragma unroll
  for (int j = 0; j < QT; ++j)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int ring_stage = 0;
  auto body = [&](int t, auto tail_c) {
    constexpr int NT = decltype(tail_c)::value ? decltype(tail_c)::value : 4;
    constexpr bool TAIL = decltype(tail_c)::value != 0;
    __builtin_amdgcn_s_barrier();
    const int stage = ring_stage;
    ring_stage = ring_stage == RING_STAGES - 1 ? 0 : ring_stage + 1;
    const char* tile = smem + stage * RING_STAGE;
    if (!active) {
      if (t + 2 < nkt) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dma.issue_any(smem, t + 2, N); }
      return;
    }
    const uint32_t sb = lbase + stage * RING_STAGE;
    u32x4 pf[QT][2];
    {
      u32x4 kf[4][2], vf[4][2];
      SITK_RING_ROWS8(kf, sb + lo.row[0], sb + lo.row[1], 0);
      SITK_RING_ROWS8(vf, sb + lo.row[0], sb + lo.row[1], 8192);
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        f32x4 s[4], dp[4];                                // row constants start the accumulators
        const f32x4 s0 = splat4(Lq[j]), d0 = splat4(ndl[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i >= NT) { s[i] = splat4(0.f); continue; }
          s[i] = Mma<h16>::mma(kf[i][0], qf[j][0], s0);
          dp[i] = Mma<h16>::mma(vf[i][0], dof[j][0], d0);
          s[i] = Mma<h16>::mma(kf[i][1], qf[j][1], s[i]);
          dp[i] = Mma<h16>::mma(vf[i][1], dof[j][1], dp[i]);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          f32x4 x;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            x[jj] = fast_exp2(s[i][jj]);
            if constexpr (TAIL) {
              if (t * 64 + 16 * i + 4 * fq + jj >= N) x[jj] = 0.f;
            }
          }
          s[i] = x * dp[i];                               // dS / scale: the scale is applied once, to dQ
        }
        pf[j][0] = pack_pair(s[0], s[1]);
        pf[j][1] = pack_pair(s[2], s[3]);
      }
    }
    u32x4 kt[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) kt[s2][dt] = ring_tr_frag(tile, lo, s2, dt);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nkt) dma.issue_any(smem, t + 2, N);
#pragma unroll
    for (int s2 = 0; s2 < (NT + 1) / 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < QT; ++j) dq[j][dt] = Mma<h16>::mma(kt[s2][dt], pf[j][s2], dq[j][dt]);
  };
  const int nfull = N >> 6;
  for (int t = 0; t < nfull; ++t) body(t, std::integral_constant<int, 0>{});
  if (nfull < nkt) tail_dispatch(N - 64 * nfull, [&](auto nt) { body(nfull, nt); });
  if (!active) return;
#pragma unroll
  for (int j = 0; j < QT; ++j)
    if (qrow[j] < N) {
      h16* row = dqkv + ((size_t)b * N + qrow[j]) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(row + 16 * dt + 4 * fq, dq[j][dt] * scale);
    }
}

// backward, key side: dK, dV.  Wave = two 16-key tiles (K and V fragments in registers); Q and dO tiles in the ring; the
// per-query statistics of the whole sequence (-lse / scale and -delta) sit in LDS and START the accumulators of the
// S and dP products (S' = Q K^T - lse / scale, p = exp2(c S'); dP' = dO V^T - delta), so the elementwise part is one
// multiply + exp2 and one multiply per element.  Padded query rows carry -inf there: p = 0.
template <int WAVES, int QT, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void attn_bwd_dkv_ring_kernel(const h16* __restrict__ qkv, const h16* __restrict__ d_o,
                                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                                       h16* __restrict__ dqkv, int N, int H, float scale, int nkb) {
  __shared__ __attribute__((aligned(256))) char smem[RING_STAGES * RING_STAGE + 2 * RING_MAX_N * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const RingBlock bc = ring_block(nkb, H);
  const int h = bc.h, b = bc.b, I = H * 64, nqt = (N + 63) / 64;
  const size_t ld = (size_t)3 * I;
  const h16* base = qkv + (size_t)b * N * ld;
  RingLoader<WAVES> dma;
  dma.init(base + h * 64, ld, d_o + (size_t)b * N * I + h * 64, (size_t)I, wave, lane);
  float* sL = reinterpret_cast<float*>(smem + RING_STAGES * RING_STAGE);
  float* sD = sL + RING_MAX_N;
  for (int r = tid; r < nqt * 64; r += WAVES * 64) {
    const size_t ridx = ((size_t)b * H + h) * N + min(r, N - 1);
    sL[r] = r < N ? -lse[ridx] * kLog2e : -INFINITY;
    sD[r] = r < N ? -delta[ridx] : 0.f;
  }
  dma.issue_any(smem, 0, N);
  if (nqt > 1) dma.issue_any(smem, 1, N);

  const int kt0 = (bc.x * WAVES + wave) * QT;
  const bool active = kt0 * 16 < N;
  u32x4 kf[QT][2], vf[QT][2];
  int krow[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    krow[j] = (kt0 + j) * 16 + fr;
    const int kc = min(krow[j], N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int eo = ks * 32 + fq * 8;
      kf[j][ks] = scale_frag(*reinterpret_cast<const u32x4*>(base + (size_t)kc * ld + I + h * 64 + eo), scale * kLog2e);
      vf[j][ks] = *reinterpret_cast<const u32x4*>(base + (size_t)kc * ld + 2 * I + h * 64 + eo);
    }
  }
  const LaneOffs lo = lane_offs_h16(lane);
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lstat = lbase + RING_STAGES * RING_STAGE + 16 * fq;      // + t * 256 + i * 64 ; sD at + RING_MAX_N * 4
  f32x4 dk[QT][4], dv[QT][4];
#pragma unroll
  for (int j = 0; j < QT; ++j)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                        // statistics visible

  int ring_stage = 0;
  // nt_c: 0 = full query tile; NT > 0 = the last, partly filled tile (only NT of its 16-query blocks are computed)
  auto body = [&](int t, auto nt_c) {
    constexpr int NT = decltype(nt_c)::value ? decltype(nt_c)::value : 4, NS2 = (NT + 1) / 2;
    __builtin_amdgcn_s_barrier();
    const int stage = ring_stage;
    ring_stage = ring_stage == RING_STAGES - 1 ? 0 : ring_stage + 1;
    const char* tile = smem + stage * RING_STAGE;
    if (!active) {
      if (t + 2 < nqt) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dma.issue_any(smem, t + 2, N); }
      return;
    }
    const uint32_t sb = lbase + stage * RING_STAGE;
    const uint32_t st = lstat + t * 256;
    u32x4 pp[QT][2], pds[QT][2];
    {
      u32x4 qr[4][2], dor[4][2];
      SITK_RING_ROWS8(qr, sb + lo.row[0], sb + lo.row[1], 0);
      SITK_RING_ROWS8(dor, sb + lo.row[0], sb + lo.row[1], 8192);
      f32x4 sl[4], sd[4];                                 // -lse log2 e and -delta of the tile's 64 queries: the initial accumulators
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\t"
                   "ds_read_b128 %3, %8 offset:192\n\tds_read_b128 %4, %8 offset:8192\n\tds_read_b128 %5, %8 offset:8256\n\t"
                   "ds_read_b128 %6, %8 offset:8320\n\tds_read_b128 %7, %8 offset:8384\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(sl[0]), "=&v"(sl[1]), "=&v"(sl[2]), "=&v"(sl[3]), "=&v"(sd[0]), "=&v"(sd[1]), "=&v"(sd[2]), "=&v"(sd[3])
                   : "v"(st)
                   : "memory");
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        f32x4 s[4], dp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i >= NT) { s[i] = splat4(0.f); dp[i] = splat4(0.f); continue; }
          s[i] = Mma<h16>::mma(qr[i][0], kf[j][0], sl[i]);
          dp[i] = Mma<h16>::mma(dor[i][0], vf[j][0], sd[i]);
          s[i] = Mma<h16>::mma(qr[i][1], kf[j][1], s[i]);
          dp[i] = Mma<h16>::mma(dor[i][1], vf[j][1], dp[i]);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          f32x4 x;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) x[jj] = fast_exp2(s[i][jj]);
          s[i] = x;
          dp[i] = x * dp[i];                              // dS / scale: the scale is applied once, to dK
        }
        pp[j][0] = pack_pair(s[0], s[1]);
        pp[j][1] = pack_pair(s[2], s[3]);
        pds[j][0] = pack_pair(dp[0], dp[1]);
        pds[j][1] = pack_pair(dp[2], dp[3]);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 f = ring_tr_frag(tile + 8192, lo, s2, dt);          // dO^T
#pragma unroll
        for (int j = 0; j < QT; ++j) dv[j][dt] = Mma<h16>::mma(f, pp[j][s2], dv[j][dt]);
      }
    u32x4 qt_[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) qt_[s2][dt] = ring_tr_frag(tile, lo, s2, dt);   // Q^T
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nqt) dma.issue_any(smem, t + 2, N);
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < QT; ++j) dk[j][dt] = Mma<h16>::mma(qt_[s2][dt], pds[j][s2], dk[j][dt]);
  };
  const int nfull = N >> 6;
  for (int t = 0; t < nfull; ++t) body(t, std::integral_constant<int, 0>{});
  if (nfull < nqt) tail_dispatch(N - 64 * nfull, [&](auto nt) { body(nfull, nt); });
  if (!active) return;
#pragma unroll
  for (int j = 0; j < QT; ++j)
    if (krow[j] < N) {
      h16* row = dqkv + ((size_t)b * N + krow[j]) * ld + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store4(row + I + 16 * dt + 4 * fq, dk[j][dt] * scale);
        store4(row + 2 * I + 16 * dt + 4 * fq, dv[j][dt]);
      }
    }
}

// Launch geometry of the ring kernels: workgroups of FOUR waves (one per SIMD), two 16-row tiles per wave.  Measured at
// B = 32, N = 1281, H = 6 (tools/attn_bench.py; forward / query side / key side, us): 4 waves x 2 tiles 123 / 158 / 200;
// 7 x 2 (least padding) 152 / 188 / 232; 8 x 2 150 / 184 / 237; 5 x 2 188 / 216 / 289; 8 x 1 140 / 209 / 250.  Small
// workgroups win although they re-stream K / V more often and pad more (88 tile slots for 81 tiles): three (forward) or two
// of them share a CU, each SIMD then holds waves of DIFFERENT barrier domains, and one workgroup's MFMA phase runs under
// another's softmax and waits.  (At N = 321 the sequence-resident kernels stay ahead: 17 / 22 / 27 us against 19 / 26 / 33.)
constexpr int RING_W = 4, RING_QT = 2;
static int ring_blocks(int N) { return ((N + 15) / 16 + RING_W * RING_QT - 1) / (RING_W * RING_QT); }
static bool ring_supported(int N) { return N > RES_MAX_N && N <= RING_MAX_N; }

template <typename T>
static int run_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
#ifdef SITK_AB
    { int rc; if (pk_try_fwd(qkv, o, lse, B, N, H, scale, s, &rc)) return rc; }
#endif
    if (N <= RES_MAX_N) {
#ifdef SITK_AB
      static const int qt_var = sitk_ab_switch("SITK_ATTN_FWD_QT", 0);     // 82 = 8 waves x 2 tiles, 83 = 8 x 3, 122 = 12 x 2
      if (qt_var == 82) hipLaunchKernelGGL((attn_fwd_resq_kernel<8, 2>), dim3(B * H), dim3(512), 0, s, reinterpret_cast<const h16*>(qkv), reinterpret_cast<h16*>(o), lse, N, H, scale);
      else if (qt_var == 83) hipLaunchKernelGGL((attn_fwd_resq_kernel<8, 3>), dim3(B * H), dim3(512), 0, s, reinterpret_cast<const h16*>(qkv), reinterpret_cast<h16*>(o), lse, N, H, scale);
      else if (qt_var == 122) hipLaunchKernelGGL((attn_fwd_resq_kernel<12, 2>), dim3(B * H), dim3(768), 0, s, reinterpret_cast<const h16*>(qkv), reinterpret_cast<h16*>(o), lse, N, H, scale);
      else
#endif
      hipLaunchKernelGGL(attn_fwd_res_kernel<16>, dim3(B * H), dim3(1024), 0, s, reinterpret_cast<const h16*>(qkv),
                         reinterpret_cast<h16*>(o), lse, N, H, scale);
      return check_launch("attention_fwd_res");
    }
    if (ring_supported(N)) {
      const int nqb = ring_blocks(N);
      hipLaunchKernelGGL((attn_fwd_ring_kernel<RING_W, RING_QT, 3>), dim3(nqb * H * B), dim3(RING_W * 64), 0, s,
                         reinterpret_cast<const h16*>(qkv), reinterpret_cast<h16*>(o), lse, N, H, scale, nqb);
      return check_launch("attention_fwd_ring");
    }
  }
  dim3 grid(cdiv(N, 64) * H * B);
  hipLaunchKernelGGL((attn_fwd_kernel<T>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(qkv),
                     reinterpret_cast<T*>(o), lse, N, H, scale);
  return check_launch("attention_fwd");
}

static bool bwd_proj_supported(int N, int D, int dtype) { return dtype == SITK_H16 && N <= RES_MAX_N && D == FOLD_D; }

// phases: bit 0 = the query-side kernel (dQ, delta[, dO]), bit 1 = the key-side kernel (dK, dV); 3 = the whole backward.
// (Single phases exist so that a profiler / bench.py can time each kernel of the pair by itself.)
static int run_bwd_proj(const void* qkv, const void* o, const void* dxmid, const void* wo_t, void* d_o, const float* lse,
                        float* delta, void* dqkv, int B, int N, int H, float scale, hipStream_t s, int phases = 3) {
#ifdef SITK_AB
  { int rc; if (pk_try_bwd_proj(qkv, o, dxmid, wo_t, d_o, lse, delta, dqkv, B, N, H, scale, s, phases, &rc)) return rc; }
#endif
  if (phases == 3 && N >= QRES_MIN_N && N <= QRES_MAX_N && sitk_ab_switch("SITK_ATTN_QRES", 1)) {    // ... with Q resident (the 321-token shapes)
    hipLaunchKernelGGL((attn_bwd_res_kernel<true, true>), dim3(B * H), dim3(1024), 0, s, reinterpret_cast<const h16*>(qkv),
                       reinterpret_cast<const h16*>(o), (const h16*)nullptr, lse, delta, reinterpret_cast<h16*>(dqkv), N, H,
                       scale, reinterpret_cast<const h16*>(dxmid), reinterpret_cast<const h16*>(wo_t),
                       reinterpret_cast<h16*>(d_o));
    return check_launch("attention_bwd_proj_res_q");
  }
  if (phases == 3 && sitk_ab_switch("SITK_ATTN_MERGED", 1)) {    // both sides in one launch
    hipLaunchKernelGGL((attn_bwd_res_kernel<true>), dim3(B * H), dim3(1024), 0, s, reinterpret_cast<const h16*>(qkv),
                       reinterpret_cast<const h16*>(o), (const h16*)nullptr, lse, delta, reinterpret_cast<h16*>(dqkv), N, H,
                       scale, reinterpret_cast<const h16*>(dxmid), reinterpret_cast<const h16*>(wo_t),
                       reinterpret_cast<h16*>(d_o));
    return check_launch("attention_bwd_proj_res");
  }
  if (phases & 1)
  hipLaunchKernelGGL((attn_bwd_dq_res_kernel<16, true>), dim3(B * H), dim3(1024), 0, s, reinterpret_cast<const h16*>(qkv),
                     reinterpret_cast<const h16*>(o), (const h16*)nullptr, lse, delta, reinterpret_cast<h16*>(dqkv), N, H,
                     scale, reinterpret_cast<const h16*>(dxmid), reinterpret_cast<const h16*>(wo_t),
                     reinterpret_cast<h16*>(d_o));
  SITK_LAUNCH_CHECK("attention_bwd_proj_dq_res");
  if (phases & 2)
  hipLaunchKernelGGL(attn_bwd_dkv_res_kernel<8>, dim3(B * H), dim3(512), 0, s, reinterpret_cast<const h16*>(qkv),
                     reinterpret_cast<const h16*>(d_o), lse, delta, reinterpret_cast<h16*>(dqkv), N, H, scale);
  return check_launch("attention_bwd_dkv_res");
}

template <typename T>
static int run_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                   int B, int N, int H, float scale, hipStream_t s, int phases = 3) {
  if constexpr (sizeof(T) == 2) {
#ifdef SITK_AB
    { int rc; if (pk_try_bwd(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, s, phases, &rc)) return rc; }
#endif
    if (N <= RES_MAX_N && phases == 3 && sitk_ab_switch("SITK_ATTN_MERGED", 1)) {    // both sides in one launch
      hipLaunchKernelGGL((attn_bwd_res_kernel<false>), dim3(B * H), dim3(1024), 0, s, reinterpret_cast<const h16*>(qkv),
                         reinterpret_cast<const h16*>(o), reinterpret_cast<const h16*>(d_o), lse, delta,
                         reinterpret_cast<h16*>(dqkv), N, H, scale, (const h16*)nullptr, (const h16*)nullptr, (h16*)nullptr);
      return check_launch("attention_bwd_res");
    }
    if (N <= RES_MAX_N) {
      if (phases & 1)
      hipLaunchKernelGGL(attn_bwd_dq_res_kernel<16>, dim3(B * H), dim3(1024), 0, s, reinterpret_cast<const h16*>(qkv),
                         reinterpret_cast<const h16*>(o), reinterpret_cast<const h16*>(d_o), lse, delta,
                         reinterpret_cast<h16*>(dqkv), N, H, scale);
      SITK_LAUNCH_CHECK("attention_bwd_dq_res");
      if (phases & 2)
      hipLaunchKernelGGL(attn_bwd_dkv_res_kernel<8>, dim3(B * H), dim3(512), 0, s, reinterpret_cast<const h16*>(qkv),
                         reinterpret_cast<const h16*>(d_o), lse, delta, reinterpret_cast<h16*>(dqkv), N, H, scale);
      return check_launch("attention_bwd_dkv_res");
    }
    if (ring_supported(N)) {
      const int nb = ring_blocks(N);
      const h16 *q_ = reinterpret_cast<const h16*>(qkv), *o_ = reinterpret_cast<const h16*>(o), *do_ = reinterpret_cast<const h16*>(d_o);
      h16* dq_ = reinterpret_cast<h16*>(dqkv);
      if (phases & 1)
        hipLaunchKernelGGL((attn_bwd_dq_ring_kernel<RING_W, RING_QT, 2>), dim3(nb * H * B), dim3(RING_W * 64), 0, s, q_, o_, do_, lse,
                           delta, dq_, N, H, scale, nb);
      SITK_LAUNCH_CHECK("attention_bwd_dq_ring");
      if (phases & 2)
        hipLaunchKernelGGL((attn_bwd_dkv_ring_kernel<RING_W, RING_QT, 2>), dim3(nb * H * B), dim3(RING_W * 64), 0, s, q_, do_, lse,
                           delta, dq_, N, H, scale, nb);
      return check_launch("attention_bwd_dkv_ring");
    }
  }
  dim3 grid(cdiv(N, 64) * H * B);
  if (phases & 1)
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(qkv),
                     reinterpret_cast<const T*>(o), reinterpret_cast<const T*>(d_o), lse, delta,
                     reinterpret_cast<T*>(dqkv), N, H, scale);
  SITK_LAUNCH_CHECK("attention_bwd_dq");
  if (phases & 2)
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(qkv),
                     reinterpret_cast<const T*>(d_o), lse, delta, reinterpret_cast<T*>(dqkv), N, H, scale);
  return check_launch("attention_bwd_dkv");
}

}  // namespace sitk

SITK_F16_TWIN(sitk_attention_fwd)
extern "C" int sitk_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, int dtype,
                                  sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_attention_fwd, qkv, o, lse, B, N, H, scale, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(qkv && o && lse, "attention_fwd: null pointer");
  SITK_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attention_fwd: bad shape B=%d N=%d H=%d", B, N, H);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return run_fwd<h16>(qkv, o, lse, B, N, H, scale, s);
  if (dtype == SITK_F32) return run_fwd<float>(qkv, o, lse, B, N, H, scale, s);
  set_error("attention_fwd: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_attention_bwd)
extern "C" int sitk_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                                  void* dqkv, int B, int N, int H, float scale, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_attention_bwd, qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(qkv && o && d_o && lse && delta && dqkv, "attention_bwd: null pointer");
  SITK_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attention_bwd: bad shape B=%d N=%d H=%d", B, N, H);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SITK_H16) return run_bwd<h16>(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, s);
  if (dtype == SITK_F32) return run_bwd<float>(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, s);
  set_error("attention_bwd: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

SITK_F16_TWIN(sitk_attention_bwd_proj_supported)
extern "C" int sitk_attention_bwd_proj_supported(int N, int D, int dtype) {
  SITK_FORWARD_F16(dtype, sitk_attention_bwd_proj_supported, N, D, dtype); return sitk::bwd_proj_supported(N, D, dtype) ? 1 : 0; }

SITK_F16_TWIN(sitk_attention_bwd_proj)
extern "C" int sitk_attention_bwd_proj(const void* qkv, const void* o, const void* dxmid, const void* wo_t, void* d_o,
                                       const float* lse, float* delta, void* dqkv, int B, int N, int H, int D, float scale,
                                       int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_attention_bwd_proj, qkv, o, dxmid, wo_t, d_o, lse, delta, dqkv, B, N, H, D, scale, dtype, stream);
  using namespace sitk;
  SITK_REQUIRE(qkv && o && dxmid && wo_t && d_o && lse && delta && dqkv, "attention_bwd_proj: null pointer");
  SITK_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attention_bwd_proj: bad shape B=%d N=%d H=%d", B, N, H);
  SITK_REQUIRE(bwd_proj_supported(N, D, dtype), "attention_bwd_proj: unsupported N=%d D=%d dtype=%d (h16, N <= %d, D == %d)", N,
               D, dtype, RES_MAX_N, FOLD_D);
  return run_bwd_proj(qkv, o, dxmid, wo_t, d_o, lse, delta, dqkv, B, N, H, scale, reinterpret_cast<hipStream_t>(stream));
}

SITK_F16_TWIN(sitk_attention_bwd_phases)
extern "C" int sitk_attention_bwd_phases(const void* qkv, const void* o, const void* d_o_in, const void* dxmid, const void* wo_t,
                                         void* d_o_out, const float* lse, float* delta, void* dqkv, int B, int N, int H, int D,
                                         float scale, int dtype, int phases, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_attention_bwd_phases, qkv, o, d_o_in, dxmid, wo_t, d_o_out, lse, delta, dqkv, B, N, H, D, scale, dtype, phases, stream);
  using namespace sitk;
  SITK_REQUIRE(qkv && o && lse && delta && dqkv && phases >= 1 && phases <= 3, "attention_bwd_phases: bad arguments");
  SITK_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attention_bwd_phases: bad shape B=%d N=%d H=%d", B, N, H);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (wo_t) {
    SITK_REQUIRE(dxmid && d_o_out && bwd_proj_supported(N, D, dtype), "attention_bwd_phases: projection fold unsupported here");
    return run_bwd_proj(qkv, o, dxmid, wo_t, d_o_out, lse, delta, dqkv, B, N, H, scale, s, phases);
  }
  SITK_REQUIRE(d_o_in, "attention_bwd_phases: null d_o");
  if (dtype == SITK_H16) return run_bwd<h16>(qkv, o, d_o_in, lse, delta, dqkv, B, N, H, scale, s, phases);
  if (dtype == SITK_F32) return run_bwd<float>(qkv, o, d_o_in, lse, delta, dqkv, B, N, H, scale, s, phases);
  set_error("attention_bwd_phases: bad dtype %d", dtype);
  return SITK_ERR_INVALID;
}

#if defined(SITK_AB) && !defined(SITK_TU_F16)
// diagnostic build: the s_memtime stamps of the merged sequence-resident backward kernel ([workgroup][wave][12])
extern "C" int sitk_debug_res_stamps(void* out, size_t bytes) {
  if (bytes != sizeof(sitk::g_res_stamps)) return (int)sizeof(sitk::g_res_stamps);
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sitk::g_res_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif
