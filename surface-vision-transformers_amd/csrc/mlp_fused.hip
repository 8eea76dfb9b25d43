// sitk fused MLP half of an encoder block (bf16 compute mode, dim = 192), forward and backward.
//
//   forward   h   = LayerNorm(x)                          layers.i.1.norm
//             u   = h W1^T + b1 ;  g = gelu_erf(u)        layers.i.1.fn.net.0 / GELU
//             out = g W2^T + b2 + x                       layers.i.1.fn.net.3 + residual
//             saved: g (operand of net.3's weight gradient) and gd = gelu'(u) - 1/2 -- NOT u: nothing in backward needs the
//             pre-activation except through gelu', forward has Phi(u) in hand (one more interpolation and one FMA per
//             element give gelu' = Phi + u phi), and backward's elementwise phase shrinks from table look-ups and 11
//             vector instructions per element to one multiplication (round 4: these kernels are bound by vector +
//             matrix ISSUE at three waves per SIMD, profiles/r02_mlp_stamps.txt)
//
//   backward  du  = (dy W2) * (gd + 1/2)
//             dh  = du W1
//             dx  = dy + LayerNorm'(dh)                   + per-workgroup dgamma / dbeta partials
//
// Both directions are the same two chained GEMMs, so they share one kernel body.  As separate
// launches each direction costs a LayerNorm pass and two GEMMs whose (tokens x mlp_dim) operand makes
// an extra HBM round trip; here it never leaves the registers between the two products.
//
// A workgroup owns 16 TT TG tokens: TG token groups of TT MFMA column tiles, two waves per group.  Two
// geometries are launched: (TG, TT) = (6, 1), 96 tokens as 12 waves of 16 tokens -- 3 waves on every SIMD --
// whenever 96-token workgroups cover the problem in one round (fused_block_rows()), else (4, 2), 128 tokens
// as 8 waves of 32.  [Measured at the BASELINE shape, 20 544 tokens: 6 waves of 32 tokens sit 2,2,1,1 on the
// SIMDs and the loaded pair sets the pace; 12 x 16 re-reads every weight fragment from LDS per 16 instead of
// per 32 tokens (LDS array 36 % -> ~60 % busy in the loop) and still wins 3.6 % of the whole training step.]
// The text below is written for TT = 2.  Wave w = (tg, hh): token group tg = w >> 1 (32
// tokens = 2 MFMA column tiles) and half hh = w & 1 of every 64-unit chunk of the hidden dimension.
// The wave keeps the 32 x 192 operand of the first product (h, or dy) in registers for the whole
// kernel (12 fragments) and walks the hidden dimension in chunks of 64 units:
//     uacc (32 hidden x 32 tokens) = Wa[chunk half] . operand      6 k-steps x 4 MFMAs
//     elementwise (bias + GELU, or GELU' with the saved u) in registers; u / du / g go to HBM once
//     yacc (192 x 32 tokens)      += Wb[:, chunk half] . uacc      12 x 2 MFMAs; the accumulator pair of
//                                     the first product IS the B operand of the second
// Wa / Wb chunks (24 KB each) stream global -> LDS by LDS-DMA, double buffered, one raw barrier per
// chunk.  The rows of the Wa chunk are stored permuted (slot 16 i + r <-> hidden 8 (r >> 2) + 4 i +
// (r & 3)) so that a lane's 2 x 4 accumulator values are 8 CONSECUTIVE hidden units: they form the
// natural-order B fragment of the second product and one 16-byte global store of u.  Fragment reads
// sit in asm blocks (a compiler-visible LDS read would drain the DMA queue).  At the end the two
// halves of a wave pair exchange partial sums through LDS and each finishes 96 of the 192 features.
#include "common.h"
#include "fused_epilogue.h"
#include "ln_gemm_bwd_body.h"

namespace sitk {

struct MlpParams {
  // forward                              backward
  const float* x;      // (R,192) residual stream in      | x_mid saved by forward
  const float* gamma;  // LayerNorm weight
  const float* beta;   // LayerNorm bias                   | unused
  const h16* wa;      // W1 (M,192)                       | W2^T (M,192)
  const h16* wb;      // W2 (192,M)                       | W1^T (192,M)
  const float* b1;     // (M)                              | unused
  const float* b2;     // (192)                            | unused
  h16* h;             // (R,192) LN output, saved         | unused
  float* mean;         // (R) written                      | read
  float* rstd;
  h16* u;             // (R,M) gd = gelu'(pre-activation) - 1/2: written | read
  h16* g;             // (R,M) gelu(pre-activation) written (or null) | unused
  float* out;          // (R,192) fp32                     | dx (R,192) fp32
  // backward only
  const h16* dyc;     // (R,192) compute-dtype copy of dy
  h16* dyc_make;      // not null: that copy does not exist yet -- round dy here and WRITE it (the weight gradients read it)
  const float* dy;     // (R,192) fp32 dy (residual gradient)
  h16* du;            // (R,M)
  h16* outc;          // (R,192) compute-dtype copy of dx
  float* partials;     // (gridDim.x, 2, 192) dgamma / dbeta partial sums
  // forward with the attention output projection folded in (PROJ): x_mid = x + o Wo^T + bo is computed here
  const h16* o;       // (R,192) attention output, 'b n (h d)'
  const h16* wo;      // (192,192) to_out weight
  const float* bo;     // (192)
  float* xmid;         // (R,192) fp32, written (saved for backward); the LayerNorm input and the residual of `out`
  // forward with the NEXT block's LayerNorm + to_qkv appended (NEXT): h1 = LN(out), qkv = h1 Wqkv^T
  const float* n_gamma;  // next block's layers.{i+1}.0.norm weight / bias
  const float* n_beta;
  const h16* n_w;       // (N3,192) next block's to_qkv weight
  h16* n_h;             // (R,192) saved LN output of the next block (or null)
  float* n_mean;         // (R)
  float* n_rstd;
  h16* n_y;             // (R,N3) qkv of the next block
  int N3;
  int R, M;
};

__device__ u32x4 g_zero_page_mlp[4];
__device__ unsigned long long g_mlp_stamps[2 * 16 * 8 + 64];   // + [224 .. 224 + 4 waves) of the to_qkv loop (see below: index 224 + 4 wave)   // diagnostic build (SITK_MLP_VAR=6): [wave][phase] cycle sums of workgroup 0

// GELU without transcendentals in the loop.  v_exp_f32 / v_rcp_f32 run at a quarter of the VALU rate (16
// cycles per wave instruction), and with one exp + one rcp per element the elementwise phase, not the
// MFMAs, bounded the kernel.  Instead every workgroup tabulates Phi(x) = 0.5 (1 + erf(x / sqrt2)) -- and,
// for backward, the density exp(-x^2/2)/sqrt(2 pi) -- in LDS on the grid x_i = (i - 384) / 64, i = 0..767
// (values computed in float64 at build time), and the loop interpolates linearly: 7 full-rate VALU operations and one
// ds_read per element.  Interpolation error <= h^2/8 max|f''| = 7.4e-6 (Phi), 1.2e-5 (density); outside
// [-6, 6) the end entries apply (Phi(-6) = 1e-9).  The result is closer to the reference's exact-erf GELU
// than the polynomial erfc of the unfused bf16 epilogues (2.5e-5, gemm.hip).
constexpr int MLP_TAB_N = 768;
constexpr float MLP_TAB_SCALE = 64.0f, MLP_TAB_ZERO = 384.0f, MLP_TAB_TMAX = 767.99f;
// {Phi(x_i), density(x_i)}, i = 0..768, computed in float64 by tools/gen_gelu_table.py (6 KB, L2 resident)
__device__ const float g_gelu_table[MLP_TAB_N + 1][2] = {
#include "gelu_table.inc"
};
// table position of x: t in [0, 768), entry = floor(t), weight of the next entry = fract(t)
SITK_DEV float tab_pos(float x) { return __builtin_amdgcn_fmed3f(fmaf(x, MLP_TAB_SCALE, MLP_TAB_ZERO), 0.0f, MLP_TAB_TMAX); }

// two floats -> one dword of two bf16 (v_cvt_pk_bf16_f32)
SITK_DEV uint32_t pack_h16(float a, float b) {
  h16x2 v;
  v[0] = (h16)a; v[1] = (h16)b;
  return __builtin_bit_cast(uint32_t, v);
}

SITK_DEV __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Fragment reads are software pipelined: every block first drains the reads issued by the PREVIOUS block
// (whose destination registers it carries as "+v" operands, so that no consumer can be scheduled above the
// wait) and then issues the reads of the NEXT batch into the other register set.  The MFMAs of the
// current batch run while those reads are in flight.
#define SITK_MLP_WAIT_ISSUE4(c0, c1, c2, c3, n0, n1, n2, n3, aA, aB, oA0, oA1, oB0, oB1)                    \
  asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                  \
               "ds_read_b128 %4, %8 offset:" #oA0 "\n\tds_read_b128 %5, %8 offset:" #oA1 "\n\t"            \
               "ds_read_b128 %6, %9 offset:" #oB0 "\n\tds_read_b128 %7, %9 offset:" #oB1                   \
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)        \
               : "v"(aA), "v"(aB)                                                                          \
               : "memory")
#define SITK_MLP_ISSUE4(n0, n1, n2, n3, aA, aB, oA0, oA1, oB0, oB1)                                         \
  asm volatile("ds_read_b128 %0, %4 offset:" #oA0 "\n\tds_read_b128 %1, %4 offset:" #oA1 "\n\t"            \
               "ds_read_b128 %2, %5 offset:" #oB0 "\n\tds_read_b128 %3, %5 offset:" #oB1                   \
               : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)                                                \
               : "v"(aA), "v"(aB)                                                                          \
               : "memory")
#define SITK_MLP_WAIT4(c0, c1, c2, c3)                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory")

constexpr int MLP_D = 192;
constexpr int MLP_W1B = 3 * 64 * 128;          // Wa chunk: 3 k-panels x 64 slot rows x 128 B = 24 KB
constexpr int MLP_W2B = MLP_D * 128;           // Wb chunk: 192 rows x 64 hidden (128 B)     = 24 KB
constexpr int MLP_OFF_H = 2 * (MLP_W1B + MLP_W2B);   // operand strip: 3 k-panels x 128 rows x 128 B = 48 KB
constexpr int MLP_OFF_B1 = MLP_OFF_H + 3 * 128 * 128;
constexpr int MLP_MAX_M = 1024;
constexpr int MLP_OFF_TAB_F = MLP_OFF_B1 + MLP_MAX_M * 4;        // forward: {Phi, dPhi, pdf, dpdf} x 768 = 12 KB after the bias
constexpr int MLP_SMEM = MLP_OFF_TAB_F + MLP_TAB_N * 16;         // = 163 840 B: all of a CU's LDS
static_assert(MLP_SMEM <= 163840, "LDS plan");

// TG = token groups (of 16 TT rows = 2 waves) per workgroup; (TG, TT) = (4, 2): 128 rows, 8 waves; (6, 1): 96 rows,
// 12 waves; (3, 2): 96 rows, 6 waves (A/B only).  See fused_block_rows() in fused_epilogue.h.
// The kernel's body as a device function (smem: MLP_SMEM bytes, 256-byte aligned): mlp_kernel below is this and nothing else;
// ln_gemm_mlp_bwd_kernel chains it behind the body of ln_gemm_bwd_kernel.
template <bool BWD, int VAR, int TG, bool PROJ, bool NEXT, int TT>
SITK_DEV void mlp_body(const MlpParams& p, char* smem) {
  static_assert(!NEXT || PROJ, "the appended LayerNorm + to_qkv shares the 96-row LDS plan of the projection prologue");
  constexpr int D = MLP_D, BLK = 16 * TT * TG, NT = 128 * TG, PPW = 24 / TG;   // rows, threads, DMA pieces per wave and chunk
  static_assert(!PROJ || (!BWD && BLK == 96), "the projection prologue needs Wo (72 KB) + a 96-row fp32 row buffer in LDS");
  constexpr int PAR = PROJ ? 1 : 0;                                      // ring slot of chunk 0
  constexpr int W1B = MLP_W1B, W2B = MLP_W2B;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int tg = wave >> 1, hh = wave & 1;
  const int blk0 = blockIdx.x * BLK;
  const int nchunks = p.M / 64;
  const int M = p.M;
  unsigned long long t_kernel0 = 0;
  if constexpr (VAR == 6) t_kernel0 = __builtin_amdgcn_s_memtime();

  // ---- W chunk DMA: 48 pieces of 8 rows x 128 B; waves 0..TG-1 carry Wa (24 pieces), waves TG..2TG-1 Wb ----
  const int r8 = lane >> 3;
  const bool isA = wave < TG;
  const int wsub = isA ? wave : wave - TG;                    // index among the waves that carry the same matrix
  const h16* wsrc = isA ? p.wa : p.wb;
  const int cstep = isA ? 64 * D : 64;                         // element step per chunk
  int soff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int qq = wsub * PPW + i;                             // piece within its matrix, 0..23
    if (isA) {
      const int kt = qq >> 3, s = (qq & 7) * 8 + r8;           // slot row in the 64-row panel
      const int r = s & 15, it = (s >> 4) & 1, hs = s >> 5;
      const int hidden = 32 * hs + 8 * (r >> 2) + 4 * it + (r & 3);
      const int key = ((s >> 1) & 1) | (((s >> 3) & 1) << 1);
      soff[i] = hidden * D + kt * 64 + (((lane & 7) ^ (key << 1)) * 8);
    } else {
      const int row = qq * 8 + r8;
      const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      soff[i] = row * M + (((lane & 7) ^ (key << 1)) * 8);
    }
  }
  auto issue = [&](int c, int buf) {
    char* base = smem + buf * (W1B + W2B) + (isA ? 0 : W1B) + wsub * PPW * 1024;
    const h16* src = wsrc + (size_t)c * cstep;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + soff[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
  };
  if constexpr (!PROJ) issue(0, 0);

  if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 0] = __builtin_amdgcn_s_memtime() - t_kernel0; }
  // ---- GELU tables (see above): entry i = {f(x_i), f(x_i+1) - f(x_i)}; visible to everybody after the
  //      first barrier of the loop.  Two entries per thread.  The global loads are issued here, the LDS stores
  //      (in front of which hipcc drains every outstanding load: LDS-DMA aliasing) after the LayerNorm rows
  //      have been requested too, so that the prologue pays one memory latency, not three. ----
  float tv[2][2][2];
  if constexpr (!BWD) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT * k;
      const int ii = i < MLP_TAB_N ? i : 0;
      tv[k][0][0] = g_gelu_table[ii][0]; tv[k][0][1] = g_gelu_table[ii][1];
      tv[k][1][0] = g_gelu_table[ii + 1][0]; tv[k][1][1] = g_gelu_table[ii + 1][1];
    }
  }
  auto store_tables = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT * k;
      if (i < MLP_TAB_N)
        *reinterpret_cast<f32x4*>(smem + MLP_OFF_TAB_F + i * 16) =
            f32x4{tv[k][0][0], tv[k][1][0] - tv[k][0][0], tv[k][0][1], tv[k][1][1] - tv[k][0][1]};
    }
  };

  // Row-indexed global traffic goes through buffer descriptors of THIS workgroup's rows (base = its first
  // row, num_records = its valid rows): rows past R fall outside num_records, so their loads return 0 and
  // their stores are dropped without any per-lane branch, every wave issues the same number of memory
  // operations (the vmcnt bookkeeping below relies on it), and the chunk offset rides in the scalar offset
  // (no VALU address arithmetic).  The hardware range-checks the VGPR offset only -- never the scalar
  // offset -- which is why the descriptor is per workgroup and the scalar offset stays inside a row.
  const size_t nrows = (size_t)(p.R - blk0 < BLK ? p.R - blk0 : BLK);
  const size_t RD = nrows * D, RM = nrows * M, oD = (size_t)blk0 * D, oM = (size_t)blk0 * M;
  const __amdgpu_buffer_rsrc_t r_u = make_rsrc(p.u + oM, p.u ? RM * 2 : 0);
  const __amdgpu_buffer_rsrc_t r_g = make_rsrc(p.g + oM, p.g ? RM * 2 : 0);
  const __amdgpu_buffer_rsrc_t r_du = make_rsrc(p.du + oM, BWD ? RM * 2 : 0);

  // ---- operand strip: forward = LayerNorm of the block's rows (wave: 16 rows; 16 lanes per row, 4 rows
  //      per pass, all 12 loads in flight together); backward = the compute-dtype copy of dy ----
  char* sH = smem + MLP_OFF_H;
  u32x4 hf[TT][6];
  if constexpr (PROJ) {
    // ---- attention output projection + residual + LayerNorm (layers.i.0.fn.to_out.0, layers.i.1.norm) ----
    // LDS: Wo [0, 72 KB) as [k-panel][192 rows][128 B]; fp32 row buffer [72 KB, 145.5 KB); afterwards the operand
    // strip takes [0, 36 KB) and chunk 0 of the weight ring lands in slot 1 (48 KB ..), so the ring starts odd.
    constexpr int NW = 2 * TG, WPW = 72 / NW;
    float bvals[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) bvals[k] = tid + NT * k < M ? p.b1[tid + NT * k] : 0.f;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int q = wave * WPW + i, kt = q / 24, row = (q % 24) * 8 + r8;
      const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      const h16* src = p.wo + (size_t)row * D + kt * 64 + (((lane & 7) ^ (key << 1)) * 8);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + kt * 24576 + (q % 24) * 1024), 16, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t r_o = make_rsrc(p.o + oD, RD * 2);
    u32x4 of[TT][6];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int k = 0; k < 6; ++k)
        of[t][k] = __builtin_amdgcn_raw_buffer_load_b128(r_o, ((16 * TT * tg + 16 * t + fr) * D + k * 32 + fq * 8) * 2, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 5] = __builtin_amdgcn_s_memtime() - t_kernel0; }
    f32x4 pacc[6][TT];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int t = 0; t < TT; ++t) pacc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int keyp = ((fr >> 1) & 1) | (((fr >> 3) & 1) << 1);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(smem + (k >> 1) * 24576 + (96 * hh + 16 * i + fr) * 128 +
                                                        (((k & 1) * 64 + fq * 16) ^ (keyp << 5)));
#pragma unroll
        for (int t = 0; t < TT; ++t) pacc[i][t] = Mma<h16>::mma(a, of[t][k], pacc[i][t]);
      }
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 1] = __builtin_amdgcn_s_memtime() - t_kernel0; }
    proj_residual_ln_rows<TG, TT>(smem + 73728, smem, pacc, tid, blk0, p.R, p.x, p.bo, p.gamma, p.beta, p.xmid, p.h, p.mean, p.rstd);
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 2] = __builtin_amdgcn_s_memtime() - t_kernel0; }
    issue(0, 1);                                               // the row buffer is dead: ring slot 1 takes chunk 0
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int k = 0; k < 6; ++k)
        hf[t][k] = *reinterpret_cast<const u32x4*>(smem + (k >> 1) * (BLK * 128) +
                                                   lds_off(16 * TT * tg + 16 * t + fr, (k & 1) * 64 + fq * 16));
    store_tables();
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (tid + NT * k < MLP_MAX_M) reinterpret_cast<float*>(smem + MLP_OFF_B1)[tid + NT * k] = bvals[k];
    __syncthreads();                                           // tables visible; every wave holds its fragments of the strip
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 3] = __builtin_amdgcn_s_memtime() - t_kernel0; }
  } else if constexpr (!BWD) {
    float bvals[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) bvals[k] = tid + NT * k < M ? p.b1[tid + NT * k] : 0.f;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(p.x + oD, RD * 4);
    const __amdgpu_buffer_rsrc_t r_h = make_rsrc(p.h + oD, p.h ? RD * 2 : 0);
    const int j = lane & 15, sub = lane >> 4;
    f32x4 gm[3], bt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gm[i] = load4(p.gamma + 4 * (j + 16 * i)); bt[i] = load4(p.beta + 4 * (j + 16 * i)); }
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 1] = __builtin_amdgcn_s_memtime() - t_kernel0; }
    f32x4 v[2 * TT][3];
#pragma unroll
    for (int pass = 0; pass < 2 * TT; ++pass)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        v[pass][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            r_x, ((wave * (8 * TT) + pass * 4 + sub) * D + 4 * (j + 16 * i)) * 4, 0, 0));
    store_tables();
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (tid + NT * k < MLP_MAX_M) reinterpret_cast<float*>(smem + MLP_OFF_B1)[tid + NT * k] = bvals[k];
#pragma unroll
    for (int pass = 0; pass < 2 * TT; ++pass) {
      const int r = wave * (8 * TT) + pass * 4 + sub, row = blk0 + r;
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
        const int c4 = j + 16 * i;                       // float4 index in the row: columns 4*c4 ..
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ok ? (v[pass][i][e] - mu) * rs * gm[i][e] + bt[i][e] : 0.f;
        const int byte = c4 * 8;                         // bf16 byte offset in the 384-byte row
        h16x4 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) ob[e] = (h16)o[e];
        *reinterpret_cast<h16x4*>(sH + (byte >> 7) * (BLK * 128) + lds_off(r, byte & 127)) = ob;
        if (p.h) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), r_h, (r * D + 4 * c4) * 2, 0, 0);
      }
      if (ok && j == 0 && p.mean) { p.mean[row] = mu; p.rstd[row] = rs; }
    }
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 2] = __builtin_amdgcn_s_memtime() - t_kernel0; }
    __syncthreads();
    if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 3] = __builtin_amdgcn_s_memtime() - t_kernel0; }
    // this wave's operand fragments: token tile t, k-step k <-> 16 B of row 32 tg + 16 t + fr
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int k = 0; k < 6; ++k)
        hf[t][k] = *reinterpret_cast<const u32x4*>(sH + (k >> 1) * (BLK * 128) +
                                                   lds_off(16 * TT * tg + 16 * t + fr, (k & 1) * 64 + fq * 16));
  } else {
    if (p.dyc_make) {
      // first backward kernel of a chain: dy exists in fp32 only.  The fragments are rounded from it (as sitk_cast_rows would)
      // and the wave of each pair that carries hh == 0 writes the copy; the other's descriptor has no records, so that both
      // issue the same memory operations.
      const __amdgpu_buffer_rsrc_t r_dyf = make_rsrc(p.dy + oD, RD * 4);
      const __amdgpu_buffer_rsrc_t r_mk = make_rsrc(p.dyc_make + oD, hh == 0 ? RD * 2 : 0);
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int e = (16 * TT * tg + 16 * t + fr) * D + k * 32 + fq * 8;
          const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_dyf, e * 4, 0, 0));
          const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_dyf, e * 4 + 16, 0, 0));
          hf[t][k] = u32x4{pack_h16(lo[0], lo[1]), pack_h16(lo[2], lo[3]), pack_h16(hi[0], hi[1]), pack_h16(hi[2], hi[3])};
          __builtin_amdgcn_raw_buffer_store_b128(hf[t][k], r_mk, e * 2, 0, 0);
        }
    } else {
      const __amdgpu_buffer_rsrc_t r_dyc = make_rsrc(p.dyc + oD, RD * 2);
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int k = 0; k < 6; ++k)
          hf[t][k] = __builtin_amdgcn_raw_buffer_load_b128(r_dyc, ((16 * TT * tg + 16 * t + fr) * D + k * 32 + fq * 8) * 2, 0, 0);
    }
  }

  if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 4] = __builtin_amdgcn_s_memtime() - t_kernel0; }
  // ---- per-lane LDS byte addresses ----
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int keyl = ((fr >> 1) & 1) | (((fr >> 3) & 1) << 1);
  uint32_t aw1[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    aw1[ks] = lbase + (32 * hh + fr) * 128 + ((ks * 64 + fq * 16) ^ (keyl << 5));   // + kt*8192 + i*2048 (+ buffer)
  const uint32_t aw2 = lbase + W1B + fr * 128 + ((hh * 64 + fq * 16) ^ (keyl << 5));  // + dt*2048 (+ buffer)
  const uint32_t ab1 = lbase + MLP_OFF_B1 + (32 * hh + 8 * fq) * 4;                   // + c*256
  const uint32_t ltabf = lbase + MLP_OFF_TAB_F;

  f32x4 yacc[12][TT];                                         // accumulators of the second product
#pragma unroll
  for (int dt = 0; dt < 12; ++dt)
#pragma unroll
    for (int t = 0; t < TT; ++t) yacc[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (rows x M) operands: lane's 16 bytes of token tile t sit at vo[t] + (block, chunk) scalar offset
  int vo[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) vo[t] = ((16 * TT * tg + 16 * t + fr) * M + 32 * hh + 8 * fq) * 2;
  const int so0 = 0;                                           // scalar offset: chunk only (c * 128 bytes)
  // backward: the saved pre-activations u of chunk c + 1 are fetched as soon as chunk c's have been consumed,
  // by buffer loads the COMPILER DOES NOT SEE (inline asm): a compiler-visible load that is live across the
  // loop edge makes hipcc drain vmcnt to 0 -- every store of the chunk included -- twice per iteration.  The
  // counted waits below cover them instead.
  u32x4 uc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) uc[t] = u32x4{0u, 0u, 0u, 0u};
  u32x4 srd_u = {0u, 0u, 0u, 0u};
  if constexpr (BWD) {
    const uint64_t ua = reinterpret_cast<uint64_t>(p.u + oM);
    srd_u = u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ua),
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ua >> 32)) & 0xffffu,
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(RM * 2)), 0x00020000u};
  }
#define SITK_MLP_LOAD_U(SOFF)                                                                              \
  if constexpr (TT == 2)                                                                                   \
    asm volatile("buffer_load_dwordx4 %0, %2, %4, %5 offen\n\tbuffer_load_dwordx4 %1, %3, %4, %5 offen"     \
                 : "=&v"(uc[0]), "=&v"(uc[TT - 1]) : "v"(vo[0]), "v"(vo[TT - 1]), "s"(srd_u), "s"(SOFF) : "memory"); \
  else                                                                                                     \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(uc[0]) : "v"(vo[0]), "s"(srd_u), "s"(SOFF) : "memory")
  if constexpr (BWD) { const int so_first = 0; SITK_MLP_LOAD_U(so_first); }

  // Waves w and w + 4 share a SIMD and the per-chunk barrier keeps them in lock step, so left alone they
  // would fight for the MFMA pipe in the product phases and for VALU issue in the elementwise phase without
  // ever overlapping the two.  A static priority lets one wave of each pair win the matrix pipe: it reaches
  // its elementwise phase while the other is still in its MFMAs, and the phases interleave from there.
  if (wave < TG) __builtin_amdgcn_s_setprio(2);
  // global stores each wave issues per chunk after the next chunk's DMA
  // backward adds the 2 u loads issued at the end of the elementwise phase
  const int nstores = TT * (BWD ? 2 : (p.u ? 1 : 0) + (p.g ? 1 : 0));
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define SITK_STAMP(i)                                                                    \
  if constexpr (VAR == 6) {                                                              \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                          \
    st[i] += tn - tprev;                                                                 \
    tprev = tn;                                                                          \
  }
  if constexpr (VAR == 6) { tprev = __builtin_amdgcn_s_memtime(); st[5] = tprev - t_kernel0; }
  for (int c = 0; c < nchunks; ++c) {
    const int buf = (c + PAR) & 1;
    SITK_STAMP(7)
    // chunk c's DMA (and, backward, its u loads) were issued before the previous iteration's stores
    // (pinned there by the "memory" clobbers of the fragment-read blocks), so only those stores may stay in flight
    if (c == 0 || nstores == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nstores == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (nstores == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (nstores == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (nstores == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    SITK_STAMP(0)
    __builtin_amdgcn_s_barrier();                                   // everybody's pieces landed; buffer buf^1 is free
    SITK_STAMP(1)
    const uint32_t bo = buf * (W1B + W2B);
    const uint32_t a0 = aw1[0] + bo, a1 = aw1[1] + bo, a2 = aw2 + bo;
    // batches of 4 fragments alternate between register sets X and Y.  First product, panel kt:
    // {tile 0, tile 1} x {k-step 2 kt, 2 kt + 1}; second product, group j: feature tiles 4 j .. 4 j + 3.
    u32x4 x0, x1, x2, x3, y0, y1, y2, y3;
    SITK_MLP_ISSUE4(x0, x1, x2, x3, a0, a1, 0, 2048, 0, 2048);
    if (c + 1 < nchunks) issue(c + 1, buf ^ 1);

    // ---- first product: uacc[i][t], hidden tile i (slot rows 32 hh + 16 i ..), token tile t; forward
    //      starts the accumulators at the bias ----
    f32x4 uacc[2][TT];
    if constexpr (!BWD) {
      u32x4 t0, t1;
      const uint32_t ab = ab1 + c * 256;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(t0), "=&v"(t1) : "v"(ab) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "memory");
#pragma unroll
      for (int t = 0; t < TT; ++t) { uacc[0][t] = __builtin_bit_cast(f32x4, t0); uacc[1][t] = __builtin_bit_cast(f32x4, t1); }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TT; ++t) uacc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#define SITK_MLP_FC1_MMAS(KT, f0, f1, f2, f3)                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) uacc[0][t] = Mma<h16>::mma(f0, hf[t][2 * KT], uacc[0][t]);     \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) uacc[1][t] = Mma<h16>::mma(f1, hf[t][2 * KT], uacc[1][t]);     \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) uacc[0][t] = Mma<h16>::mma(f2, hf[t][2 * KT + 1], uacc[0][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) uacc[1][t] = Mma<h16>::mma(f3, hf[t][2 * KT + 1], uacc[1][t]); \
    __builtin_amdgcn_sched_barrier(0);
    SITK_MLP_WAIT_ISSUE4(x0, x1, x2, x3, y0, y1, y2, y3, a0, a1, 8192, 10240, 8192, 10240);
    SITK_MLP_FC1_MMAS(0, x0, x1, x2, x3)
    SITK_MLP_WAIT_ISSUE4(y0, y1, y2, y3, x0, x1, x2, x3, a0, a1, 16384, 18432, 16384, 18432);
    SITK_MLP_FC1_MMAS(1, y0, y1, y2, y3)
    SITK_MLP_WAIT_ISSUE4(x0, x1, x2, x3, y0, y1, y2, y3, a2, a2, 0, 2048, 4096, 6144);   // second product, group 0
    SITK_MLP_FC1_MMAS(2, x0, x1, x2, x3)
#undef SITK_MLP_FC1_MMAS
    SITK_STAMP(2)

    // ---- elementwise; lane holds hidden c*64 + 32 hh + 8 fq + 4 i + e of token 32 tg + 16 t + fr ----
    u32x4 pf[TT];                                              // B fragments of the second product
    u32x4 sd[TT];                                              // the other stored vector (u forward, g backward)
    const int so = so0 + c * 128;
    if constexpr (!BWD) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        float gv[8], dv[8];
#pragma unroll
        for (int hf4 = 0; hf4 < 2; ++hf4) {                     // 4 elements at a time (register budget)
          float fw[4];
          uint32_t ad[4];
          u32x4 te[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float tp = tab_pos(uacc[hf4][t][e]);
            fw[e] = __builtin_amdgcn_fractf(tp);
            ad[e] = ltabf + ((uint32_t)tp << 4);
          }
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(te[0]), "=&v"(te[1]), "=&v"(te[2]), "=&v"(te[3])
                       : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                       : "memory");
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 en = __builtin_bit_cast(f32x4, te[e]);
            const float xv = uacc[hf4][t][e];
            const float cdf = fmaf(fw[e], en[1], en[0]), pdf = fmaf(fw[e], en[3], en[2]);
            gv[4 * hf4 + e] = xv * cdf;                          // gelu(u)
            dv[4 * hf4 + e] = fmaf(xv, pdf, cdf) - 0.5f;         // gelu'(u) - 1/2 = Phi(u) - 1/2 + u phi(u): centred, so that
                                                                 // the 16-bit rounding is finest where most u are (near 0)
          }
        }
        pf[t] = u32x4{pack_h16(gv[0], gv[1]), pack_h16(gv[2], gv[3]), pack_h16(gv[4], gv[5]), pack_h16(gv[6], gv[7])};
        sd[t] = u32x4{pack_h16(dv[0], dv[1]), pack_h16(dv[2], dv[3]), pack_h16(dv[4], dv[5]), pack_h16(dv[6], dv[7])};
        if (p.u) __builtin_amdgcn_raw_buffer_store_b128(sd[t], r_u, vo[t], so, 0);
        if (p.g) __builtin_amdgcn_raw_buffer_store_b128(pf[t], r_g, vo[t], so, 0);
      }
    } else {
      // u(c) was requested at the end of the previous elementwise phase; only this iteration's 6 DMA pieces are
      // younger (chunk 0's were drained by the vmcnt(0) at the top of the first iteration)
      if (c > 0) {
        if (c + 1 < nchunks) {                                  // younger: this iteration's PPW DMA pieces
          if constexpr (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else if constexpr (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // The registers become "defined" for the compiler only HERE, in one unconditional statement behind the
      // waits: tying them to the conditional wait statements themselves made hipcc merge the two branches
      // through register copies placed in front of a wait, i.e. copies of registers whose loads were in flight.
      if constexpr (TT == 2) asm volatile("" : "+v"(uc[0]), "+v"(uc[TT - 1]) : : "memory");
      else asm volatile("" : "+v"(uc[0]) : : "memory");
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        float dv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t w = uc[t][e >> 1];
#ifdef SITK_TU_F16
          const float gd = (float)__builtin_bit_cast(h16x2, w)[e & 1];                        // v_cvt_f32_f16 (word select)
#else
          const float gd = __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));   // bf16 -> f32 is a 16-bit shift
#endif
          const float a = uacc[e >> 2][t][e & 3];
          dv[e] = fmaf(a, gd, 0.5f * a);                         // du = (dy W2) gelu'(u); forward saved gelu'(u) - 1/2
        }
        pf[t] = u32x4{pack_h16(dv[0], dv[1]), pack_h16(dv[2], dv[3]), pack_h16(dv[4], dv[5]), pack_h16(dv[6], dv[7])};
        sd[t] = pf[t];
        __builtin_amdgcn_raw_buffer_store_b128(pf[t], r_du, vo[t], so, 0);
      }
      if (c + 1 < nchunks) { const int so_next = so + 128; SITK_MLP_LOAD_U(so_next); }
    }

    SITK_STAMP(3)
    // ---- second product: yacc[dt][t] += Wb[16 dt .., chunk half] . pf[t] ----
#define SITK_MLP_FC2_MMAS(J, f0, f1, f2, f3)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 0][t] = Mma<h16>::mma(f0, pf[t], yacc[4 * J + 0][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 1][t] = Mma<h16>::mma(f1, pf[t], yacc[4 * J + 1][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 2][t] = Mma<h16>::mma(f2, pf[t], yacc[4 * J + 2][t]); \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) yacc[4 * J + 3][t] = Mma<h16>::mma(f3, pf[t], yacc[4 * J + 3][t]); \
    __builtin_amdgcn_sched_barrier(0);
    SITK_MLP_WAIT_ISSUE4(y0, y1, y2, y3, x0, x1, x2, x3, a2, a2, 8192, 10240, 12288, 14336);
    SITK_MLP_FC2_MMAS(0, y0, y1, y2, y3)
    // Keep-alive: the vector memory pipeline reads the data registers of a 16-byte store well after the store
    // issues when stores queue back to back (observed on gfx950: a VALU result written three instructions
    // after the second store of a pair reached memory).  Holding the stored vectors live across the first
    // MFMA group keeps the allocator from recycling their registers while the stores may still be reading.
    asm volatile("" : : "v"(sd[0]), "v"(sd[TT - 1]), "v"(pf[0]), "v"(pf[TT - 1]));
    SITK_MLP_WAIT_ISSUE4(x0, x1, x2, x3, y0, y1, y2, y3, a2, a2, 16384, 18432, 20480, 22528);
    SITK_MLP_FC2_MMAS(1, x0, x1, x2, x3)
    SITK_MLP_WAIT4(y0, y1, y2, y3);
    SITK_MLP_FC2_MMAS(2, y0, y1, y2, y3)
#undef SITK_MLP_FC2_MMAS
    SITK_STAMP(4)
  }
  const unsigned long long t_loop_end = VAR == 6 ? __builtin_amdgcn_s_memtime() : 0;

  // ---- pair exchange: wave hh finishes features [96 hh, 96 hh + 96); the other half's partial sums
  //      cross through LDS (12 tiles x 1 KB per wave, in the W buffers that nobody reads any more) ----
  __syncthreads();
  {
    char* mine = smem + wave * (6144 * TT);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int t = 0; t < TT; ++t)   // the half this wave does NOT finish (register indices stay compile-time constants)
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
        // yacc index must be a compile-time constant: select by hh without dynamic indexing
        v[i][t] = (hh ? yacc[6 + i][t] : yacc[i][t]) + o;
      }
  }
  const int n0 = 96 * hh + 4 * fq;                             // + 16 i : this lane's 4 features of tile i

  if constexpr (!BWD) {
    if constexpr (!NEXT) {
      // out = v + b2 + x, in row layout (fused_epilogue.h)
      residual_rows_epilogue<TG, TT>(smem, v, tid, blk0, p.R, PROJ ? p.xmid : p.x, p.b2, p.out);
    } else {
      // ---- out = v + b2 + x_mid (stored), then the NEXT block's attention input: h1 = LN(out), qkv = h1 Wqkv^T.
      //      The finished rows never leave the chip between the two blocks' kernels: rows -> fp32 row buffer
      //      [72 KB ..) -> LayerNorm -> bf16 operand strip [0, 36 KB) -> 12 register fragments per wave; Wqkv
      //      streams in 24-KB chunks through a 2-slot ring at [48 KB, 96 KB) exactly as in ln_gemm_fused.hip. ----
      constexpr int NW = 2 * TG, QPW = 24 / NW;
      if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 6] = __builtin_amdgcn_s_memtime() - t_loop_end; }
      proj_residual_ln_rows<TG, TT>(smem + 73728, smem, v, tid, blk0, p.R, p.xmid, p.b2, p.n_gamma, p.n_beta, p.out, p.n_h,
                                p.n_mean, p.n_rstd);
      if constexpr (VAR == 6) { if (blockIdx.x == 80 && lane == 0) g_mlp_stamps[128 + wave * 8 + 7] = __builtin_amdgcn_s_memtime() - t_loop_end; }
      // Lane-derived addresses of this phase are rebuilt from an opaque copy of the lane id: derived from `lane`
      // itself, hipcc computes them at kernel entry and carries them through the main loop, whose register budget
      // (254 of 256) has no room for them.
      int lane_q = lane;
      asm volatile("" : "+v"(lane_q));
      const int fr = lane_q & 15, fq = lane_q >> 4, r8 = lane_q >> 3;
      const int keyl = ((fr >> 1) & 1) | (((fr >> 3) & 1) << 1);
      const int N3 = p.N3, nq = N3 / 64;
      int qoff[QPW];
#pragma unroll
      for (int i = 0; i < QPW; ++i) {
        const int qq = wave * QPW + i;
        const int kt = qq >> 3, sr = (qq & 7) * 8 + r8;
        const int r = sr & 15, it = (sr >> 4) & 1, hs = sr >> 5;
        const int feat = 32 * hs + 8 * (r >> 2) + 4 * it + (r & 3);
        const int key = ((sr >> 1) & 1) | (((sr >> 3) & 1) << 1);
        qoff[i] = feat * D + kt * 64 + (((lane_q & 7) ^ (key << 1)) * 8);
      }
      auto qissue = [&](int c, int slot) {
        char* base = smem + 49152 + slot * 24576 + wave * QPW * 1024;
        const h16* src = p.n_w + (size_t)c * 64 * D;
#pragma unroll
        for (int i = 0; i < QPW; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + qoff[i]),
                                           (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
      };
      // FOUR ring slots [48 KB, 144 KB) (the fp32 row buffer is dead), three chunks in flight: with two slots and one chunk in
      // flight this loop spent 17 k of its 26 k cycles waiting for the next 24 KB (profiles/r04_mlp_stamps_tail.txt): every CU
      // asks L2 for the same chunk at the same time and one chunk per round trip is 8.6 B / clk / CU
      constexpr int QNS = 4;
      qissue(0, 0);
      if (nq > 1) qissue(1, 1);
      if (nq > 2) qissue(2, 2);
      u32x4 qf[TT][6];
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int k = 0; k < 6; ++k)
          qf[t][k] = *reinterpret_cast<const u32x4*>(smem + (k >> 1) * (BLK * 128) +
                                                     lds_off(16 * TT * tg + 16 * t + fr, (k & 1) * 64 + fq * 16));
      const __amdgpu_buffer_rsrc_t r_y = make_rsrc(p.n_y + (size_t)blk0 * N3, nrows * N3 * 2);
      int qvo[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) qvo[t] = ((16 * TT * tg + 16 * t + fr) * N3 + 32 * hh + 8 * fq) * 2;
      uint32_t qa[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qa[ks] = lbase + 49152 + (32 * hh + fr) * 128 + ((ks * 64 + fq * 16) ^ (keyl << 5));
      unsigned long long qst[4] = {0, 0, 0, 0}, qtp = 0;     // diagnostic build: vmcnt wait / barrier / reads + MFMAs / pack + store
#define SITK_QSTAMP(i)                                                                   \
      if constexpr (VAR == 6) {                                                          \
        const unsigned long long tn = __builtin_amdgcn_s_memtime();                      \
        qst[i] += tn - qtp;                                                              \
        qtp = tn;                                                                        \
      }
      if constexpr (VAR == 6) qtp = __builtin_amdgcn_s_memtime();
      for (int c = 0; c < nq; ++c) {
        // chunk c must have landed; YOUNGER than its DMA and allowed to stay in flight: the DMAs of the chunks c + 1, c + 2 that
        // exist (QPW pieces each) and the TT stores of each of the last two iterations
        {
          const int nd = nq - 1 - c < 2 ? nq - 1 - c : 2, keep = nd * QPW + (c < 2 ? c : 2) * TT;
          switch (keep) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;      // (keep <= 2 QPW + 2 TT = 8 at TT = 2)
          }
        }
        SITK_QSTAMP(0)
        __builtin_amdgcn_s_barrier();                                 // everybody's pieces of chunk c; the slot of chunk c - 1 is free
        SITK_QSTAMP(1)
        const uint32_t qbo = (c & (QNS - 1)) * 24576;
        const uint32_t a0 = qa[0] + qbo, a1 = qa[1] + qbo;
        u32x4 x0, x1, x2, x3, y0, y1, y2, y3;
        SITK_MLP_ISSUE4(x0, x1, x2, x3, a0, a1, 0, 2048, 0, 2048);
        if (c + 3 < nq) qissue(c + 3, (c + 3) & (QNS - 1));
        f32x4 qacc[2][TT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < TT; ++t) qacc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#define SITK_MLP_Q_MMAS(KT, f0, f1, f2, f3)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) qacc[0][t] = Mma<h16>::mma(f0, qf[t][2 * KT], qacc[0][t]);     \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) qacc[1][t] = Mma<h16>::mma(f1, qf[t][2 * KT], qacc[1][t]);     \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) qacc[0][t] = Mma<h16>::mma(f2, qf[t][2 * KT + 1], qacc[0][t]); \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) qacc[1][t] = Mma<h16>::mma(f3, qf[t][2 * KT + 1], qacc[1][t]); \
        __builtin_amdgcn_sched_barrier(0);
        SITK_MLP_WAIT_ISSUE4(x0, x1, x2, x3, y0, y1, y2, y3, a0, a1, 8192, 10240, 8192, 10240);
        SITK_MLP_Q_MMAS(0, x0, x1, x2, x3)
        SITK_MLP_WAIT_ISSUE4(y0, y1, y2, y3, x0, x1, x2, x3, a0, a1, 16384, 18432, 16384, 18432);
        SITK_MLP_Q_MMAS(1, y0, y1, y2, y3)
        SITK_MLP_WAIT4(x0, x1, x2, x3);
        SITK_MLP_Q_MMAS(2, x0, x1, x2, x3)
#undef SITK_MLP_Q_MMAS
        SITK_QSTAMP(2)
        u32x4 qsd[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const f32x4 v0 = qacc[0][t], v1 = qacc[1][t];
          qsd[t] = u32x4{pack_h16(v0[0], v0[1]), pack_h16(v0[2], v0[3]), pack_h16(v1[0], v1[1]), pack_h16(v1[2], v1[3])};
          __builtin_amdgcn_raw_buffer_store_b128(qsd[t], r_y, qvo[t], c * 128, 0);
        }
        asm volatile("" : : "v"(qsd[0]), "v"(qsd[TT - 1]));         // store keep-alive (see the main loop)
        SITK_QSTAMP(3)
      }
#undef SITK_QSTAMP
      if constexpr (VAR == 6) {
        if (blockIdx.x == 80 && lane == 0)
          for (int i = 0; i < 4; ++i) g_mlp_stamps[224 + wave * 4 + i] = qst[i];
      }
    }
  } else {
    // LayerNorm backward on dh = v, in row layout (fused_epilogue.h)
    ln_bwd_rows_epilogue<TG, TT>(smem, v, tid, blk0, p.R, p.x, p.mean, p.rstd, p.gamma, p.dy, p.out, p.outc,
                         p.partials + (size_t)blockIdx.x * 2 * D);
  }
  if constexpr (VAR == 6) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[6] = __builtin_amdgcn_s_memtime() - t_loop_end;
    if (blockIdx.x == 80 && lane == 0)
      for (int i = 0; i < 8; ++i) g_mlp_stamps[wave * 8 + i] = st[i];
  }
}

template <bool BWD, int VAR = 0, int TG = 4, bool PROJ = false, bool NEXT = false, int TT = 2>
__global__ __launch_bounds__(128 * TG) void mlp_kernel(MlpParams p) {
  __shared__ __attribute__((aligned(256))) char smem[MLP_SMEM];
  mlp_body<BWD, VAR, TG, PROJ, NEXT, TT>(p, smem);
}

// d to_qkv + LayerNorm backward of layer l, then the MLP backward of layer l - 1, in ONE launch (round 4).  Both kernels give a
// workgroup the same 96 rows, and everything the second reads of the first's output (dx fp32 and its compute-dtype copy) are
// that workgroup's OWN rows: behind a workgroup barrier they come back from this XCD's L2 instead of crossing HBM, and the
// chain has one launch boundary per layer less.  12 waves x 16 tokens only (the geometry of the one-round shapes).
// The caller may ALIAS p1's dres (read by the first half) with p2's dx_mid output (written by the second): both halves own the
// same rows, and the second half starts behind the drain + barrier below (csrc/encoder.hip passes S.dxB for both).
__global__ __launch_bounds__(768) void ln_gemm_mlp_bwd_kernel(LnGemmParams p1, MlpParams p2) {
  static_assert(lg_bwd_smem<6, 1>() <= MLP_SMEM, "LDS plan");
  __shared__ __attribute__((aligned(256))) char smem[MLP_SMEM];
  ln_gemm_bwd_body<6, 1>(p1, smem, reinterpret_cast<const h16*>(g_zero_page_mlp));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's dx / dxc rows have left the CU ...
  __syncthreads();                                       // ... and nobody reads the first kernel's LDS image any more
  mlp_body<true, 0, 6, false, false, 1>(p2, smem);
}

// 96-row workgroups run as 12 waves of 16 tokens (3 per SIMD); SITK_MLP_TT1=0 selects the 6 x 32-token
// variant they replaced (2,2,1,1 waves per SIMD), kept for A/B measurements
static bool mlp_tt1() {
  static const int v = sitk_ab_switch("SITK_MLP_TT1", 1);
  return v != 0;
}

static int mlp_check(const char* what, int64_t rows, int D, int M, int dtype) {
  SITK_REQUIRE(dtype == SITK_H16 && D == MLP_D && M % 64 == 0 && M >= 64 && M <= MLP_MAX_M && rows > 0 && rows < (1ll << 31),
               "%s: the fused path is specialised for h16, dim 192, mlp_dim %% 64 == 0 and <= %d (got dtype %d dim %d mlp_dim %d)",
               what, MLP_MAX_M, dtype, D, M);
  return SITK_OK;
}

}  // namespace sitk

using namespace sitk;

SITK_F16_TWIN(sitk_mlp_fused_supported)
extern "C" int sitk_mlp_fused_supported(int D, int M, int dtype) {
  SITK_FORWARD_F16(dtype, sitk_mlp_fused_supported, D, M, dtype);
  return dtype == SITK_H16 && D == MLP_D && M % 64 == 0 && M >= 64 && M <= MLP_MAX_M;
}

SITK_F16_TWIN(sitk_mlp_fwd)
extern "C" int sitk_mlp_fwd(const float* x, const float* ln_w, const float* ln_b, const void* w1_c, const float* b1,
                            const void* w2_c, const float* b2, void* h, float* mean, float* rstd, void* u, void* g,
                            float* out, int64_t rows, int D, int M, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_mlp_fwd, x, ln_w, ln_b, w1_c, b1, w2_c, b2, h, mean, rstd, u, g, out, rows, D, M, dtype, stream);
  SITK_REQUIRE(x && ln_w && ln_b && w1_c && b1 && w2_c && b2 && out, "mlp_fwd: null pointer");
  SITK_REQUIRE((mean == nullptr) == (rstd == nullptr), "mlp_fwd: mean and rstd go together");
  SITK_TRY(mlp_check("mlp_fwd", rows, D, M, dtype));
  MlpParams p = {};
  p.x = x; p.gamma = ln_w; p.beta = ln_b;
  p.wa = reinterpret_cast<const h16*>(w1_c); p.b1 = b1; p.wb = reinterpret_cast<const h16*>(w2_c); p.b2 = b2;
  p.h = reinterpret_cast<h16*>(h); p.mean = mean; p.rstd = rstd;
  p.u = reinterpret_cast<h16*>(u); p.g = reinterpret_cast<h16*>(g); p.out = out;
  p.R = (int)rows; p.M = M;
  static const int var = sitk_ab_switch("SITK_MLP_VAR", 0);   // 6: stamped kernels (tools/mlp_stamps.py; -DSITK_AB builds only)
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
#ifdef SITK_AB
  if (var == 6 && mlp_tt1()) hipLaunchKernelGGL((mlp_kernel<false, 6, 6, false, false, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  else if (var == 6) hipLaunchKernelGGL((mlp_kernel<false, 6>), dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  else
#endif
  if (fused_block_rows(rows) == 96 && mlp_tt1()) hipLaunchKernelGGL((mlp_kernel<false, 0, 6, false, false, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  else if (fused_block_rows(rows) == 96) hipLaunchKernelGGL((mlp_kernel<false, 0, 3>), dim3(cdiv((int)rows, 96)), dim3(384), 0, hs, p);
  else hipLaunchKernelGGL((mlp_kernel<false, 0, 4>), dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  return check_launch("mlp_fwd");
}

SITK_F16_TWIN(sitk_attn_out_mlp_fused_supported)
extern "C" int sitk_attn_out_mlp_fused_supported(int64_t rows, int D, int I, int M, int dtype) {
  SITK_FORWARD_F16(dtype, sitk_attn_out_mlp_fused_supported, rows, D, I, M, dtype);
  return sitk_mlp_fused_supported(D, M, dtype) && I == MLP_D && rows > 0 && fused_block_rows(rows) == 96;
}

static int attn_out_mlp_launch(const void* o_c, const void* wo_c, const float* bo, const float* x, float* xmid,
                               const float* ln_w, const float* ln_b, const void* w1_c, const float* b1, const void* w2_c,
                               const float* b2, void* h, float* mean, float* rstd, void* u, void* g, float* out,
                               const float* n_ln_w, const float* n_ln_b, const void* n_wqkv_c, void* n_h, float* n_mean,
                               float* n_rstd, void* n_qkv, int N3, int64_t rows, int D, int I, int M, int dtype,
                               sitk_stream_t stream, const char* what) {
  SITK_REQUIRE(o_c && wo_c && bo && x && xmid && ln_w && ln_b && w1_c && b1 && w2_c && b2 && out, "%s: null pointer", what);
  SITK_REQUIRE((mean == nullptr) == (rstd == nullptr), "%s: mean and rstd go together", what);
  SITK_TRY(mlp_check(what, rows, D, M, dtype));
  SITK_REQUIRE(sitk_attn_out_mlp_fused_supported(rows, D, I, M, dtype),
               "%s: needs heads * 64 == 192 and a row count that takes 96-row workgroups (got I %d rows %lld)", what, I, (long long)rows);
  MlpParams p = {};
  p.o = reinterpret_cast<const h16*>(o_c); p.wo = reinterpret_cast<const h16*>(wo_c); p.bo = bo; p.xmid = xmid;
  p.x = x; p.gamma = ln_w; p.beta = ln_b;
  p.wa = reinterpret_cast<const h16*>(w1_c); p.b1 = b1; p.wb = reinterpret_cast<const h16*>(w2_c); p.b2 = b2;
  p.h = reinterpret_cast<h16*>(h); p.mean = mean; p.rstd = rstd;
  p.u = reinterpret_cast<h16*>(u); p.g = reinterpret_cast<h16*>(g); p.out = out;
  p.R = (int)rows; p.M = M;
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  if (n_qkv) {
    SITK_REQUIRE(n_ln_w && n_ln_b && n_wqkv_c && N3 % 64 == 0 && N3 >= 64 && (n_mean == nullptr) == (n_rstd == nullptr),
                 "%s: bad next-block arguments (N3 %d)", what, N3);
    p.n_gamma = n_ln_w; p.n_beta = n_ln_b; p.n_w = reinterpret_cast<const h16*>(n_wqkv_c);
    p.n_h = reinterpret_cast<h16*>(n_h); p.n_mean = n_mean; p.n_rstd = n_rstd; p.n_y = reinterpret_cast<h16*>(n_qkv); p.N3 = N3;
#ifdef SITK_AB
    static const int var = sitk_ab_switch("SITK_MLP_VAR", 0);   // 6: stamped kernel (tools/mlp_stamps.py tail)
    if (var == 6) hipLaunchKernelGGL((mlp_kernel<false, 6, 6, true, true, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
    else
#endif
    hipLaunchKernelGGL((mlp_kernel<false, 0, 6, true, true, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  } else {
    hipLaunchKernelGGL((mlp_kernel<false, 0, 6, true, false, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  }
  return check_launch(what);
}

SITK_F16_TWIN(sitk_attn_out_mlp_fwd)
extern "C" int sitk_attn_out_mlp_fwd(const void* o_c, const void* wo_c, const float* bo, const float* x, float* xmid,
                                     const float* ln_w, const float* ln_b, const void* w1_c, const float* b1, const void* w2_c,
                                     const float* b2, void* h, float* mean, float* rstd, void* u, void* g, float* out,
                                     int64_t rows, int D, int I, int M, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_attn_out_mlp_fwd, o_c, wo_c, bo, x, xmid, ln_w, ln_b, w1_c, b1, w2_c, b2, h, mean, rstd, u, g, out, rows, D, I, M, dtype, stream);
  return attn_out_mlp_launch(o_c, wo_c, bo, x, xmid, ln_w, ln_b, w1_c, b1, w2_c, b2, h, mean, rstd, u, g, out, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, 0, rows, D, I, M, dtype, stream, "attn_out_mlp_fwd");
}

SITK_F16_TWIN(sitk_attn_out_mlp_next_fwd)
extern "C" int sitk_attn_out_mlp_next_fwd(const void* o_c, const void* wo_c, const float* bo, const float* x, float* xmid,
                                          const float* ln_w, const float* ln_b, const void* w1_c, const float* b1,
                                          const void* w2_c, const float* b2, void* h, float* mean, float* rstd, void* u, void* g,
                                          float* out, const float* n_ln_w, const float* n_ln_b, const void* n_wqkv_c, void* n_h,
                                          float* n_mean, float* n_rstd, void* n_qkv, int N3, int64_t rows, int D, int I, int M,
                                          int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_attn_out_mlp_next_fwd, o_c, wo_c, bo, x, xmid, ln_w, ln_b, w1_c, b1, w2_c, b2, h, mean, rstd, u, g, out, n_ln_w, n_ln_b, n_wqkv_c, n_h, n_mean, n_rstd, n_qkv, N3, rows, D, I, M, dtype, stream);
  SITK_REQUIRE(n_qkv != nullptr, "attn_out_mlp_next_fwd: null qkv output");
  return attn_out_mlp_launch(o_c, wo_c, bo, x, xmid, ln_w, ln_b, w1_c, b1, w2_c, b2, h, mean, rstd, u, g, out, n_ln_w, n_ln_b,
                             n_wqkv_c, n_h, n_mean, n_rstd, n_qkv, N3, rows, D, I, M, dtype, stream, "attn_out_mlp_next_fwd");
}

// diagnostic: per-phase cycle sums of workgroup 0 written by the SITK_MLP_VAR=6 build (not part of the ABI header)
extern "C" int sitk_mlp_debug_stamps(unsigned long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_mlp_stamps), sizeof(unsigned long long) * (256 + 64)) == hipSuccess ? 0 : -1;
}

extern "C" size_t sitk_mlp_bwd_partial_floats(int64_t rows) {
  return rows > 0 ? (size_t)cdiv64(rows, fused_bwd_block_rows(rows)) * 2 * MLP_D : 0;
}

static int mlp_bwd_launch(const float* dy, const void* dy_c, void* dy_c_make, const float* x, const float* mean, const float* rstd,
                          const float* ln_w, const void* w2t_c, const void* w1t_c, const void* gd, void* du, float* dx,
                          void* dx_c, float* partials, int64_t rows, int D, int M, int dtype, sitk_stream_t stream) {
  SITK_REQUIRE(dy && (dy_c || dy_c_make) && x && mean && rstd && ln_w && w2t_c && w1t_c && gd && du && dx && dx_c && partials,
               "mlp_bwd: null pointer");
  SITK_TRY(mlp_check("mlp_bwd", rows, D, M, dtype));
  MlpParams p = {};
  p.x = x; p.gamma = ln_w; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
  p.wa = reinterpret_cast<const h16*>(w2t_c); p.wb = reinterpret_cast<const h16*>(w1t_c);
  p.u = const_cast<h16*>(reinterpret_cast<const h16*>(gd));
  p.du = reinterpret_cast<h16*>(du); p.dy = dy; p.dyc = reinterpret_cast<const h16*>(dy_c);
  p.dyc_make = reinterpret_cast<h16*>(dy_c_make);
  p.out = dx; p.outc = reinterpret_cast<h16*>(dx_c); p.partials = partials;
  p.R = (int)rows; p.M = M;
  static const int var = sitk_ab_switch("SITK_MLP_VAR", 0);
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
#ifdef SITK_AB
  if (var == 6 && mlp_tt1()) hipLaunchKernelGGL((mlp_kernel<true, 6, 6, false, false, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  else if (var == 6) hipLaunchKernelGGL((mlp_kernel<true, 6>), dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  else
#endif
  if (fused_bwd_block_rows(rows) == 96 && mlp_tt1()) hipLaunchKernelGGL((mlp_kernel<true, 0, 6, false, false, 1>), dim3(cdiv((int)rows, 96)), dim3(768), 0, hs, p);
  else if (fused_bwd_block_rows(rows) == 96) hipLaunchKernelGGL((mlp_kernel<true, 0, 3>), dim3(cdiv((int)rows, 96)), dim3(384), 0, hs, p);
  else hipLaunchKernelGGL((mlp_kernel<true, 0, 4>), dim3(cdiv((int)rows, 128)), dim3(512), 0, hs, p);
  return check_launch("mlp_bwd");
}

SITK_F16_TWIN(sitk_mlp_bwd)
extern "C" int sitk_mlp_bwd(const float* dy, const void* dy_c, const float* x, const float* mean, const float* rstd,
                            const float* ln_w, const void* w2t_c, const void* w1t_c, const void* gd, void* du,
                            float* dx, void* dx_c, float* partials, int64_t rows, int D, int M, int dtype,
                            sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_mlp_bwd, dy, dy_c, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, du, dx, dx_c, partials, rows, D, M, dtype, stream);
  SITK_REQUIRE(dy_c, "mlp_bwd: null pointer");
  return mlp_bwd_launch(dy, dy_c, nullptr, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, du, dx, dx_c, partials, rows, D, M, dtype, stream);
}

// sitk_mlp_bwd for the FIRST backward kernel of a chain, where dy exists in fp32 only: the kernel rounds its operand fragments
// from dy and writes the compute-dtype copy to dy_c (an OUTPUT here; the weight gradient of net.3 reads it) -- sitk_cast_rows +
// sitk_mlp_bwd in one launch, same bits.
SITK_F16_TWIN(sitk_mlp_bwd_cast)
extern "C" int sitk_mlp_bwd_cast(const float* dy, void* dy_c, const float* x, const float* mean, const float* rstd,
                                 const float* ln_w, const void* w2t_c, const void* w1t_c, const void* gd, void* du,
                                 float* dx, void* dx_c, float* partials, int64_t rows, int D, int M, int dtype,
                                 sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_mlp_bwd_cast, dy, dy_c, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, du, dx, dx_c, partials, rows, D, M, dtype, stream);
  SITK_REQUIRE(dy_c, "mlp_bwd_cast: null pointer");
  return mlp_bwd_launch(dy, nullptr, dy_c, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, du, dx, dx_c, partials, rows, D, M, dtype, stream);
}

SITK_F16_TWIN(sitk_ln_gemm_mlp_bwd_supported)
extern "C" int sitk_ln_gemm_mlp_bwd_supported(int64_t rows, int D, int N, int M, int dtype) {
  SITK_FORWARD_F16(dtype, sitk_ln_gemm_mlp_bwd_supported, rows, D, N, M, dtype);
  return sitk_mlp_fused_supported(D, M, dtype) && N % 64 == 0 && N >= 64 && rows > 0 && rows * (int64_t)N < (1ll << 30) &&
         fused_bwd_block_rows(rows) == 96 && mlp_tt1() && sitk_ab_switch("SITK_LG_TT1", 1) && sitk_ab_switch("SITK_BWD_PAIR", 1);
}

SITK_F16_TWIN(sitk_ln_gemm_mlp_bwd)
extern "C" int sitk_ln_gemm_mlp_bwd(const void* dqkv, const void* wqkv_t_c, const float* x, const float* mean1, const float* rstd1,
                                    const float* ln1_w, const float* dres, float* dx, void* dx_c, float* partials1, int N,
                                    const float* xmid, const float* mean2, const float* rstd2, const float* ln2_w,
                                    const void* w2t_c, const void* w1t_c, const void* gd, void* du, float* dx_mid, void* dx_mid_c,
                                    float* partials2, int64_t rows, int D, int M, int dtype, sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_ln_gemm_mlp_bwd, dqkv, wqkv_t_c, x, mean1, rstd1, ln1_w, dres, dx, dx_c, partials1, N, xmid, mean2, rstd2, ln2_w, w2t_c, w1t_c, gd, du, dx_mid, dx_mid_c, partials2, rows, D, M, dtype, stream);
  SITK_REQUIRE(dqkv && wqkv_t_c && x && mean1 && rstd1 && ln1_w && dx && dx_c && partials1 && xmid && mean2 && rstd2 && ln2_w &&
               w2t_c && w1t_c && gd && du && dx_mid && dx_mid_c && partials2, "ln_gemm_mlp_bwd: null pointer");
  SITK_REQUIRE(sitk_ln_gemm_mlp_bwd_supported(rows, D, N, M, dtype),
               "ln_gemm_mlp_bwd: needs h16, dim 192 and a row count that takes 96-row workgroups (got rows %lld dim %d N %d mlp_dim %d)",
               (long long)rows, D, N, M);
  LnGemmParams p1 = {};
  p1.x = x; p1.gamma = ln1_w; p1.w = reinterpret_cast<const h16*>(wqkv_t_c);
  p1.mean = const_cast<float*>(mean1); p1.rstd = const_cast<float*>(rstd1);
  p1.y = const_cast<h16*>(reinterpret_cast<const h16*>(dqkv));
  p1.dres = dres; p1.dx = dx; p1.dxc = reinterpret_cast<h16*>(dx_c); p1.partials = partials1;
  p1.R = (int)rows; p1.N = N;
  MlpParams p2 = {};
  p2.x = xmid; p2.gamma = ln2_w; p2.mean = const_cast<float*>(mean2); p2.rstd = const_cast<float*>(rstd2);
  p2.wa = reinterpret_cast<const h16*>(w2t_c); p2.wb = reinterpret_cast<const h16*>(w1t_c);
  p2.u = const_cast<h16*>(reinterpret_cast<const h16*>(gd));
  p2.du = reinterpret_cast<h16*>(du); p2.dy = dx; p2.dyc = reinterpret_cast<const h16*>(dx_c);
  p2.out = dx_mid; p2.outc = reinterpret_cast<h16*>(dx_mid_c); p2.partials = partials2;
  p2.R = (int)rows; p2.M = M;
  hipLaunchKernelGGL(ln_gemm_mlp_bwd_kernel, dim3(cdiv((int)rows, 96)), dim3(768), 0, reinterpret_cast<hipStream_t>(stream), p1, p2);
  return check_launch("ln_gemm_mlp_bwd");
}
