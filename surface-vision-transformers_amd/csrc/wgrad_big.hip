// sitk weight gradients, large-tile kernel with a slab reduction (16-bit compute types, dims that are multiples of 192).
//
//   dW[n][k] += sum_m dY[m][n] X[m][k]        (and db[n] += sum_m dY[m][n])
//
// The 64x64-tile kernel of gemm.hip stages 32 flop per byte and is bound by L2 -> LDS traffic
// (profiles/r01_pmc_wgrad_group.txt).  Rounds 1 - 2 ran 128 x 192 output tiles (77 flop/B, 64-token stages, two token
// halves per workgroup): still 9 - 12 TB/s between L2 and the CUs at every model size with the matrix pipe a quarter to a
// third busy.  Now a workgroup owns TWO such blocks (wgrad_x2_kernel below: 128 x 384 or 256 x 192):
//   * the two operands are the "P" and the "Q" side; whichever of dY / X has the dimension that is a multiple of 192
//     (384) goes on the Q side (the tile is written back transposed when that is dY);
//   * 4 waves, each with a 64 x 192 fp32 accumulator (48 MFMA tiles, 192 AGPRs: one wave per SIMD owns the register file);
//   * 32-token stages (8 or 7 panels of 32 rows x 128 B) move global -> LDS by LDS-DMA through a 5-deep ring, four stages in
//     flight, one raw barrier per stage; transposed fragments are read inside asm blocks (a compiler-visible LDS read would
//     make hipcc drain the DMA ring with s_waitcnt vmcnt(0));
//   * bookkeeping stays in units of 128 x 192 BLOCKS (WbProblem.tiles / tiles_q / block_begin, the slab, the reduce kernel);
//   * token-split partial blocks are written to a slab with plain stores and summed into dW by a second kernel: bitwise
//     reproducible, and none of the partials goes through float atomics (1.3 TB/s chip-wide, MI355X_MICROARCH.md).
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace sitk {

// One launch takes up to 52 problems (the 4 Linears of up to 12 encoder layers + the patch embedding: 5.4 KB of
// kernel arguments): a whole backward slice brings enough tiles to give each workgroup a long token run (see sitk_encoder_bwd).
constexpr int WB_MAX_PROBLEMS = 52;   // 12 layers x 4 + the patch embedding (+ spare)
constexpr int WB_TILE_ELEMS = 128 * 192;
struct WbProblem {
  const h16* P;   // P side operand (M, ldp): 128-column block rows
  const h16* Q;   // Q side operand (M, ldq): 192-column block columns
  int ldp, ldq, cp, cq;  // leading dims and total columns of each side
  float* dW;
  int lddw;
  int swapped;     // 0: P = dY (rows n), Q = X (cols k); 1: P = X (cols k), Q = dY (rows n)
  float* db;       // bias gradient of the dY side or null
  int M, tiles_q, tiles, block_begin, splits, chunk;
  // row map of each side (sitk_rowmap; group 0 = identity).  group % 64 == 0, so a 32-row stage never straddles two
  // groups and the running DMA pointer only takes (stride - group) extra rows when a stage starts a new group.
  int pgroup, pstride, poffset, qgroup, qstride, qoffset;
};
struct WbGroup {
  WbProblem p[WB_MAX_PROBLEMS];
  int count;
};

__device__ u32x4 g_zero_page_wb[4];

#define SITK_WB_TR2(dlo, dhi, areg, off, offhi) \
  "ds_read_b64_tr_b16 " dlo ", " areg " offset:" #off "\n\tds_read_b64_tr_b16 " dhi ", " areg " offset:" #offhi "\n\t"

// MFMA with the accumulator tile pinned to the AGPR half of the register file ("+a").  The kernel holds 60
// accumulator tiles (240 registers); left to the allocator they are parked in AGPRs and copied through VGPRs
// around every MFMA (441 v_accvgpr moves per 64-MFMA stage, more issue slots than the MFMAs themselves).
// Every tile is touched once per stage, so no two of these instructions are dependent within a stage; the
// first compiler-visible reader comes after the barrier that ends the loop.
SITK_DEV void mma_acc(f32x4& acc, u32x4 a, u32x4 b) {
  asm("v_mfma_f32_16x16x32_" SITK_H16_MNEMONIC " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the same, pinned in program order (the stage loop interleaves it with asm LDS reads and DMA issues by hand)
SITK_DEV void mma_acc_v(f32x4& acc, u32x4 a, u32x4 b) {
  asm volatile("v_mfma_f32_16x16x32_" SITK_H16_MNEMONIC " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// ------------------------------------------------------------------------------------------
// Two forms of the double tile:
//   WIDE  128 x 384 (0.8 x the operand bytes per flop of one block): dims 384 / 768, where every Linear has one side that
//         is a multiple of 384 and the other a multiple of 128; the four waves are 2 (P halves) x 2 (Q halves);
//   TALL  256 x 192 (0.7 x): dim 192 (576 = 2.25 x 256 and 192 = 0.75 x 256 are padded: 11 % more MFMAs, which were idle);
//         the four waves are the four P quarters and share the Q fragments.
// ALL four waves take every token of a stage (no token-half reduction at the end).  Software pipeline (one wave per SIMD, so
// nothing else hides a stall): while the 48 MFMAs of stage s run, the wave reads the fragments of stage s + 1 into the
// other register set and issues its DMA pieces of stage s + 4 -- one transposed read pair after every third MFMA, one DMA
// piece after every sixth.  A stage is ALWAYS issued and always read (past the end: zero page into a slot nobody reads,
// stale LDS into registers nobody uses), so the loop has no conditions and the wait in front of every stage is constant.
// The stage loop exists in three instantiations (no bias / dY on the P side / dY on the Q side), chosen once per wave: with
// the bias MFMAs under a run-time condition inside ONE loop, hipcc carried the bias tiles through VGPR copies of their
// AGPRs in every stage.  The finished tile leaves through LDS in two 128 x 192 halves = two blocks of the block numbering
// (side by side / one above the other; TALL rounds the P side up to an even number of block rows).
// Measured (MI355X): config 5's 12-layer launch 7.82 -> 7.04 ms (matrix pipe 52 % busy at the 1.85 GHz the chip holds
// under it), config 3's 1.85 -> 1.71 ms; SiT-tiny's 12-layer launch 337 -> 302 us, its two-layer side-stream launches 340 -> 260 us.
// ------------------------------------------------------------------------------------------
template <bool TALL>
__global__ __launch_bounds__(256) void wgrad_x2_kernel(WbGroup grp, float* __restrict__ slab) {
  constexpr int PNL = 4096;                       // one panel: 32 rows x 128 B
  constexpr int NPP = TALL ? 4 : 2, NQP = TALL ? 3 : 6;   // panels of the P / Q side = DMA pieces per wave of each side
  constexpr int PCS = NPP + NQP;
  constexpr int STG = PCS * PNL;
  constexpr int NSTG = 5;
  __shared__ __attribute__((aligned(256))) char smem[NSTG * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = TALL ? 0 : (wave & 1), wp = TALL ? wave : (wave >> 1);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);            // in units of double tiles = two 128 x 192 blocks
  int pi = 0;
  for (int i = 1; i < grp.count; ++i)
    if (2 * bid >= grp.p[i].block_begin) pi = i;
  const WbProblem P = grp.p[pi];
  const int tiles_r = P.tiles >> 1, tq_r = TALL ? P.tiles_q : (P.tiles_q >> 1);
  const int local = bid - (P.block_begin >> 1);
  const int split = local / tiles_r, tile = local % tiles_r;
  const int pr = tile / tq_r, qr = tile % tq_r;
  const int p0 = pr * (TALL ? 256 : 128), q0 = qr * (TALL ? 192 : 384);
  const int mbeg = split * P.chunk, mend = min(P.M, mbeg + P.chunk);
  const char* zero = reinterpret_cast<const char*>(g_zero_page_wb);

  const int r8 = lane >> 3;
  const h16* zerop = reinterpret_cast<const h16*>(zero);
  const h16* pbase[PCS];
  int prow[PCS], pdst[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const bool isP = i < NPP;
    const int q = isP ? wave * NPP + i : wave * NQP + (i - NPP);
    const int panel = q >> 2, row = (q & 3) * 8 + r8;
    const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int col = (isP ? p0 : q0) + panel * 64 + ((lane & 7) ^ (key << 1)) * 8;
    const bool colok = col < (isP ? P.cp : P.cq);
    const int ld = isP ? P.ldp : P.ldq;
    const int grp_i = isP ? P.pgroup : P.qgroup, lrow = mbeg + row;
    const int frow = grp_i ? (lrow / grp_i) * (isP ? P.pstride : P.qstride) + (isP ? P.poffset : P.qoffset) + lrow % grp_i : lrow;
    pbase[i] = colok ? (isP ? P.P : P.Q) + (size_t)frow * ld + col : zerop;
    prow[i] = colok ? row : (1 << 30);
    pdst[i] = (isP ? 0 : NPP * PNL) + q * 1024;
  }
  const char* zp = zero;
  asm volatile("" : "+v"(zp));
  const char* pcur[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) pcur[i] = reinterpret_cast<const char*>(pbase[i]);
  const int pincP = __builtin_amdgcn_readfirstlane(32 * P.ldp * (int)sizeof(h16));
  const int pincQ = __builtin_amdgcn_readfirstlane(32 * P.ldq * (int)sizeof(h16));
  int left = mend - mbeg;
  int gposP = P.pgroup ? mbeg % P.pgroup : 0, gposQ = P.qgroup ? mbeg % P.qgroup : 0;
  const int extraP = P.pgroup ? (P.pstride - P.pgroup) * P.ldp * (int)sizeof(h16) : 0;
  const int extraQ = P.qgroup ? (P.qstride - P.qgroup) * P.ldq * (int)sizeof(h16) : 0;
  int incP = pincP, incQ = pincQ;
  auto next_stage_steps = [&]() __attribute__((always_inline)) {
    gposP += 32;
    gposQ += 32;
    incP = pincP;
    incQ = pincQ;
    if (P.pgroup && gposP >= P.pgroup) { gposP = 0; incP += extraP; }
    if (P.qgroup && gposQ >= P.qgroup) { gposQ = 0; incQ += extraQ; }
  };
  auto issue_piece = [&](int i, char* sb) __attribute__((always_inline)) {
    const char* src = prow[i] < left ? pcur[i] : zp;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sb + pdst[i]), 16, 0, 0);
    pcur[i] += i < NPP ? incP : incQ;
    asm volatile("" : "+v"(pcur[i]));
  };
  auto issue = [&](int stage) __attribute__((always_inline)) {
    char* sb = smem + stage * STG;
    next_stage_steps();
#pragma unroll
    for (int i = 0; i < PCS; ++i) issue_piece(i, sb);
    left -= 32;
  };

  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int rowl = 8 * (lane >> 4) + ((lane >> 2) & 3);
  const int keyl = ((rowl >> 1) & 1) | (((rowl >> 3) & 1) << 1);
  uint32_t toffp[4], toffq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t t = lbase + rowl * 128 + 32 * (i ^ keyl) + 8 * (lane & 3);
    toffp[i] = t + wp * PNL;
    toffq[i] = t + (NPP + 3 * wq) * PNL;
  }

  f32x4 acc[4][12], accb[12];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 12; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias: column sums of the dY side through one extra MFMA per dY block against a ones fragment, by the waves that own
  // the columns once (P side: every wave its 64 columns, on the first Q tile; Q side: the waves of the first P quarter / half)
  const bool biasP = P.db != nullptr && !P.swapped && q0 == 0 && wq == 0;
  const bool biasQ = P.db != nullptr && P.swapped && p0 == 0 && wp == 0;
  u32x4 ones;
  {
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)1.0f;
    ones = __builtin_bit_cast(u32x4, o);
  }

  const int nstage = (mend - mbeg + 31) / 32;
  struct Frags {
    u32x2 pl[4], ph[4], ql[12], qh[12];
  };
  Frags fa, fb;
  auto read_pair = [&](Frags& f, int r, uint32_t so) __attribute__((always_inline)) {
    if (r < 4) {
      const uint32_t a = toffp[r] + so;
      asm volatile(SITK_WB_TR2("%0", "%1", "%2", 0, 512) : "=&v"(f.pl[r]), "=&v"(f.ph[r]) : "v"(a));
    } else {
      const int j = r - 4;
      const uint32_t a = toffq[j & 3] + so;
      if (j < 4) asm volatile(SITK_WB_TR2("%0", "%1", "%2", 0, 512) : "=&v"(f.ql[j]), "=&v"(f.qh[j]) : "v"(a));
      else if (j < 8) asm volatile(SITK_WB_TR2("%0", "%1", "%2", 4096, 4608) : "=&v"(f.ql[j]), "=&v"(f.qh[j]) : "v"(a));
      else asm volatile(SITK_WB_TR2("%0", "%1", "%2", 8192, 8704) : "=&v"(f.ql[j]), "=&v"(f.qh[j]) : "v"(a));
    }
  };
  for (int i = 0; i < NSTG - 1; ++i) issue(i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PCS) : "memory");   // stage 0 has landed (1..3 in flight)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) read_pair(fa, r, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  auto stage_body = [&](Frags& cur, Frags& nxt, int s, auto bp, auto bq) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PCS) : "memory");   // stage s + 1 has landed (s + 2, s + 3 may be in flight)
    __builtin_amdgcn_s_barrier();
    char* sb = smem + ((s + NSTG - 1) % NSTG) * STG;           // slot of stage s - 1: every wave has its fragments
    const uint32_t so = ((s + 1) % NSTG) * STG;
    next_stage_steps();
    u32x4 fp[4], fqv[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) fp[i] = u32x4{cur.pl[i][0], cur.pl[i][1], cur.ph[i][0], cur.ph[i][1]};
#pragma unroll
    for (int j = 0; j < 12; ++j) fqv[j] = u32x4{cur.ql[j][0], cur.ql[j][1], cur.qh[j][0], cur.qh[j][1]};
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      mma_acc_v(acc[m / 12][m % 12], fp[m / 12], fqv[m % 12]);
      if (m % 3 == 0) read_pair(nxt, m / 3, so);
      if (m % 6 == 5 && m / 6 < PCS) issue_piece(m / 6, sb);
    }
    left -= 32;
    if constexpr (decltype(bp)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mma_acc_v(accb[i], fp[i], ones);
    }
    if constexpr (decltype(bq)::value) {
#pragma unroll
      for (int j = 0; j < 12; ++j) mma_acc_v(accb[j], ones, fqv[j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto stages = [&](auto bp, auto bq) __attribute__((always_inline)) {
    for (int s = 0; s < nstage; s += 2) {
      stage_body(fa, fb, s, bp, bq);
      stage_body(fb, fa, s + 1, bp, bq);
    }
  };
  if (biasP) stages(std::true_type{}, std::false_type{});
  else if (biasQ) stages(std::false_type{}, std::true_type{});
  else stages(std::false_type{}, std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // acc[i][j][jj] <-> P column wp*64 + 16i + 4fq + jj, Q column wq*192 + 16j + fr of the double tile.  Half h (WIDE: the
  // waves with wq == h; TALL: those with wp / 2 == h) = one block of the 128 x 192 numbering:
  const int fr = lane & 15, fq = lane >> 4;
  const size_t blk_base = (size_t)P.block_begin + (size_t)split * P.tiles;
  auto blk_of = [&](int h) -> size_t { return blk_base + (TALL ? (size_t)(2 * pr + h) * P.tiles_q + qr : (size_t)tile * 2 + h); };
  const size_t total_blocks = (size_t)gridDim.x * 2;
  const bool vec_ok = (P.lddw & 3) == 0 && (reinterpret_cast<uintptr_t>(P.dW) & 15) == 0;
  float* til = reinterpret_cast<float*>(smem);
  const int myh = TALL ? (wp >> 1) : wq, myrow = (TALL ? (wp & 1) : wp) * 64;
  for (int h = 0; h < 2; ++h) {
    if (myh == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) til[(myrow + 16 * i + 4 * fq + jj) * 192 + 16 * j + fr] = acc[i][j][jj];
    }
    __syncthreads();
    const int ph0 = p0 + (TALL ? 128 * h : 0), qh0 = q0 + (TALL ? 0 : 192 * h);
    if (P.splits > 1) {
      float* out = slab + blk_of(h) * WB_TILE_ELEMS;
      for (int e4 = tid * 4; e4 < WB_TILE_ELEMS; e4 += 1024) *reinterpret_cast<f32x4*>(out + e4) = *reinterpret_cast<const f32x4*>(til + e4);
    } else if (!P.swapped) {
      for (int e4 = tid * 4; e4 < WB_TILE_ELEMS; e4 += 1024) {
        const int r = e4 / 192, c = e4 % 192, pc = ph0 + r, qc = qh0 + c;
        if (pc >= P.cp || qc >= P.cq) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(til + e4);
        float* dst = P.dW + (size_t)pc * P.lddw + qc;
        if (qc + 3 < P.cq && vec_ok) {
          *reinterpret_cast<f32x4*>(dst) += v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (qc + e < P.cq) dst[e] += v[e];
        }
      }
    } else {
      for (int t = tid; t < WB_TILE_ELEMS / 4; t += 256) {
        const int c = t % 192, r0 = (t / 192) * 4, qc = qh0 + c, pc = ph0 + r0;
        if (qc >= P.cq || pc >= P.cp) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = til[(r0 + e) * 192 + c];
        float* dst = P.dW + (size_t)qc * P.lddw + pc;
        if (pc + 3 < P.cp && vec_ok) {
          *reinterpret_cast<f32x4*>(dst) += v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (pc + e < P.cp) dst[e] += v[e];
        }
      }
    }
    __syncthreads();
  }
  // Bias gradient: this kernel has no token halves -- the whole sum goes to slot 0 of the block's [2][192] pair, zeros to slot 1
  float* bs = slab + total_blocks * WB_TILE_ELEMS;
  if (biasP && fr == 0) {
    const size_t blk = blk_of(myh);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int c = myrow + 16 * i + 4 * fq + jj, n = p0 + (TALL ? 128 * myh : 0) + c;
        if (P.splits > 1) { bs[(blk * 2 + 0) * 192 + c] = accb[i][jj]; bs[(blk * 2 + 1) * 192 + c] = 0.f; }
        else if (n < P.cp) unsafeAtomicAdd(P.db + n, accb[i][jj]);
      }
  }
  if (biasQ && fq == 0) {
    const size_t blk = blk_of(TALL ? 0 : wq);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int c = 16 * j + fr, n = q0 + 192 * wq + c;
      if (P.splits > 1) { bs[(blk * 2 + 0) * 192 + c] = accb[j][0]; bs[(blk * 2 + 1) * 192 + c] = 0.f; }
      else if (n < P.cq) unsafeAtomicAdd(P.db + n, accb[j][0]);
    }
  }
}

// dW (+)= sum over the splits of a tile's slabs, honouring the orientation; one thread per 4 Q columns
__global__ __launch_bounds__(256) void wgrad_big_reduce_kernel(WbGroup grp, const float* __restrict__ slab, int total_tiles,
                                                                 int total_blocks) {
  const int gt = blockIdx.y;  // global tile index over all problems
  int pi = 0, tbase = 0;
  for (int i = 0; i < grp.count; ++i) {
    if (gt < tbase + grp.p[i].tiles) { pi = i; break; }
    tbase += grp.p[i].tiles;
  }
  const WbProblem P = grp.p[pi];
  if (P.splits == 1) return;                              // written by the tile's own workgroup
  const int tile = gt - tbase;
  const int p0 = (tile / P.tiles_q) * 128, q0 = (tile % P.tiles_q) * 192;
  const int t = blockIdx.x * 256 + threadIdx.x;          // 6144 threads per 128 x 192 tile, 4 elements each
  if (t * 4 >= WB_TILE_ELEMS) return;
  const size_t sbase = (size_t)(P.block_begin + tile) * WB_TILE_ELEMS, sstep = (size_t)P.tiles * WB_TILE_ELEMS;
  // bias partials of the tile's splits ([block][token half][192], behind the tile slabs), summed in split order by the first
  // 192 threads of the tiles that carry the bias (the same predicates as in the tile kernel)
  if (P.db && t < 192 && ((!P.swapped && q0 == 0) || (P.swapped && p0 == 0))) {
    const float* bs = slab + (size_t)total_blocks * WB_TILE_ELEMS;
    const bool pside = !P.swapped;
    const int n = (pside ? p0 : q0) + t;
    if (t < (pside ? 128 : 192) && n < (pside ? P.cp : P.cq)) {
      float b = 0.f;
      // P side: both token halves of all four waves hold sums; Q side: only the waves with wh == 0 (token halves wt = 0, 1)
      for (int sp = 0; sp < P.splits; ++sp) {
        const size_t blk = (size_t)(P.block_begin + tile) + (size_t)sp * P.tiles;
        b += bs[(blk * 2 + 0) * 192 + t];
        b += bs[(blk * 2 + 1) * 192 + t];
      }
      P.db[n] += b;
    }
  }
  if (!P.swapped) {                                       // dW rows = P columns: 4 consecutive Q columns per thread
    const int e4 = t * 4, r = e4 / 192, c = e4 % 192;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < P.splits; ++sp) s += *reinterpret_cast<const f32x4*>(slab + sbase + sp * sstep + e4);
    const int pc = p0 + r;
    if (pc >= P.cp) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qc = q0 + c + e;
      if (qc < P.cq) P.dW[(size_t)pc * P.lddw + qc] += s[e];
    }
  } else {                                                // dW rows = Q columns: 4 consecutive P columns per thread, so
    const int c = t % 192, r0 = (t / 192) * 4;            // that the transposed write is 16 bytes per thread too
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < P.splits; ++sp)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += slab[sbase + sp * sstep + (size_t)(r0 + e) * 192 + c];
    const int qc = q0 + c, pc = p0 + r0;
    if (qc >= P.cq) return;
    float* dst = P.dW + (size_t)qc * P.lddw + pc;
    if (pc + 3 < P.cp && (P.lddw & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {   // (a C-ABI caller's dW may be 4-byte aligned only)
      *reinterpret_cast<f32x4*>(dst) += s;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pc + e < P.cp) dst[e] += s[e];
    }
  }
}

static bool wb_eligible(const sitk_wgrad_desc& d) {
  // 16-byte operand vectors: a column count that is not a multiple of 8 needs its zero-filled tail inside the row pitch
  const bool align = d.N % 4 == 0 && d.K % 4 == 0 && d.lddy % 8 == 0 && d.ldx % 8 == 0 && d.lddy >= ((d.N + 7) & ~7) &&
                     d.ldx >= ((d.K + 7) & ~7);
  const bool plain = (d.dymap.group == 0 || d.dymap.group % 64 == 0) && d.xmap.group == 0 && !d.dy_is_f32;
  return align && plain && d.M >= 2048 && (d.K % 192 == 0 || d.N % 192 == 0);
}

// the 128 x 384 kernel takes a launch whose every problem has a side that is a multiple of 384 (dims 384 / 768; not dim 192:
// to_qkv and to_out of SiT-tiny have 192 / 576 on both sides, and a tiny launch is sized for the CUs beside the chain)
static bool wb_wide(const sitk_wgrad_desc* d, int count) {
  for (int i = 0; i < count; ++i)
    if (d[i].K % 384 != 0 && d[i].N % 384 != 0) return false;
  return true;
}

// Plans a launch in units of 128 x 192 blocks; a workgroup takes a PAIR of them (WIDE: side by side, the Q side is then
// a multiple of 384; TALL: one above the other, the P side rounded up to an even number of block rows) and rounds are
// counted in pairs.
enum { WB_WIDE = 1, WB_TALL = 2 };
static int wb_mode(const sitk_wgrad_desc* d, int count) { return wb_wide(d, count) ? WB_WIDE : WB_TALL; }

static int wb_plan(const sitk_wgrad_desc* d, int count, WbGroup& g, int& blocks, int& tiles_total, int cus, int mode) {
  tiles_total = 0;
  const bool wide = mode == WB_WIDE, tall = mode == WB_TALL;
  const int qm = wide ? 384 : 192;
  auto prows = [&](int cp) { return tall ? 2 * cdiv(cp, 256) : cdiv(cp, 128); };   // block rows of the P side (TALL: an even number)
  for (int i = 0; i < count; ++i) {
    WbProblem& p = g.p[i];
    // Q side (192-column tiles) = X columns (k) or dY columns (n): whichever orientation needs fewer 128 x 192 tiles
    // (net.3 of dim 192: dY 192 x X 768 is 2 x 4 tiles with dY on the 128 side, half of them half empty, but 6 x 1
    // the other way round)
    const int tiles_normal = d[i].K % qm == 0 ? prows(d[i].N) * (d[i].K / 192) : (1 << 30);
    const int tiles_swapped = d[i].N % qm == 0 ? prows(d[i].K) * (d[i].N / 192) : (1 << 30);
    const bool normal = tiles_normal <= tiles_swapped;
    p.swapped = normal ? 0 : 1;
    p.P = reinterpret_cast<const h16*>(normal ? d[i].dY : d[i].X);
    p.Q = reinterpret_cast<const h16*>(normal ? d[i].X : d[i].dY);
    p.ldp = normal ? d[i].lddy : d[i].ldx;
    p.ldq = normal ? d[i].ldx : d[i].lddy;
    p.cp = normal ? d[i].N : d[i].K;
    p.cq = normal ? d[i].K : d[i].N;
    p.dW = d[i].dW; p.lddw = d[i].lddw; p.db = d[i].db; p.M = d[i].M;
    p.pgroup = p.pstride = p.poffset = p.qgroup = p.qstride = p.qoffset = 0;
    if (d[i].dymap.group) {                                   // the row-mapped operand is dY: P side when normal
      (normal ? p.pgroup : p.qgroup) = d[i].dymap.group;
      (normal ? p.pstride : p.qstride) = d[i].dymap.stride;
      (normal ? p.poffset : p.qoffset) = d[i].dymap.offset;
    }
    p.tiles_q = p.cq / 192;
    p.tiles = prows(p.cp) * p.tiles_q;
    tiles_total += p.tiles;
  }
  g.count = count;
  blocks = 0;
  // Token splits.  Up to one round of tiles (one workgroup per CU): split until the 256 CUs are covered.  More than one
  // round: every tile is the same amount of work, so the launch takes ceil(blocks / 256) rounds; 2 - 4 token splits
  // shorten the rounds and fill the last one (3.4 rounds run as 4; split in two, 6.75 as 7:
  // -12 %), at the price of one slab write + read per block (~1.5 % of a block's operand bytes per split).
  // cus < 256: the launch is meant to run BESIDE another kernel chain on the CUs that chain leaves idle (encoder.hip's side
  // streams): it may occupy `cus` CUs, not the chip
  int many = 1;
  const int units = tiles_total / 2;                            // workgroups per split (two blocks each)
  if (units > cus) {
    double best = 1e30;
    for (int sp = 1; sp <= 4; ++sp) {
      const double rounds = (double)cdiv(units * sp, cus);
      const double cost = rounds / sp * (1.0 + (sp > 1 ? 0.015 * sp : 0.0));
      if (cost < best - 1e-9) { best = cost; many = sp; }
    }
  }
  for (int i = 0; i < count; ++i) {
    WbProblem& p = g.p[i];
    int splits = units > cus ? many : std::max(1, cus / units);
    splits = std::min(splits, std::max(1, p.M / 256));
    p.chunk = cdiv(cdiv(p.M, splits), 64) * 64;
    p.splits = cdiv(p.M, p.chunk);
    p.block_begin = blocks;
    blocks += p.tiles * p.splits;
  }
  return SITK_OK;
}

}  // namespace sitk

using namespace sitk;

SITK_F16_TWIN(sitk_gemm_wgrad_group_ws_bytes)
extern "C" size_t sitk_gemm_wgrad_group_ws_bytes(const sitk_wgrad_desc* d, int count, int dtype) {
  SITK_FORWARD_F16(dtype, sitk_gemm_wgrad_group_ws_bytes, d, count, dtype);
  if (!d || count < 1 || count > WB_MAX_PROBLEMS || dtype != SITK_H16) return 0;
  for (int i = 0; i < count; ++i)
    if (!wb_eligible(d[i])) return 0;
  WbGroup g;
  int blocks, tiles;
  wb_plan(d, count, g, blocks, tiles, 256, wb_mode(d, count));             // (a launch for fewer CUs checks its own plan against the size it is given)
  return (size_t)blocks * (WB_TILE_ELEMS + 2 * 192) * sizeof(float);      // tile slabs + bias partials of split tiles
}

SITK_F16_TWIN(sitk_gemm_wgrad_group_ws)
extern "C" int sitk_gemm_wgrad_group_ws(const sitk_wgrad_desc* d, int count, int dtype, void* ws, size_t ws_bytes,
                                        sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gemm_wgrad_group_ws, d, count, dtype, ws, ws_bytes, stream);
  return sitk_gemm_wgrad_group_ws_cus(d, count, dtype, ws, ws_bytes, 256, stream);
}

SITK_F16_TWIN(sitk_gemm_wgrad_group_ws_cus)
extern "C" int sitk_gemm_wgrad_group_ws_cus(const sitk_wgrad_desc* d, int count, int dtype, void* ws, size_t ws_bytes, int cus,
                                            sitk_stream_t stream) {
  SITK_FORWARD_F16(dtype, sitk_gemm_wgrad_group_ws_cus, d, count, dtype, ws, ws_bytes, cus, stream);
  SITK_REQUIRE(d != nullptr && count >= 1 && count <= WB_MAX_PROBLEMS, "gemm_wgrad_group_ws: 1..%d problems", WB_MAX_PROBLEMS);
  SITK_REQUIRE(cus >= 1 && cus <= 256, "gemm_wgrad_group_ws: cus = %d (1..256)", cus);
  const size_t need = sitk_gemm_wgrad_group_ws_bytes(d, count, dtype);
  if (need == 0 || ws == nullptr || ws_bytes < need) {       // generic tiles, 4 problems per launch
    for (int i0 = 0; i0 < count; i0 += 4) SITK_TRY(sitk_gemm_wgrad_group(d + i0, std::min(4, count - i0), dtype, stream));
    return SITK_OK;
  }
  for (int i = 0; i < count; ++i)
    SITK_REQUIRE(d[i].dY && d[i].X && d[i].dW, "gemm_wgrad_group_ws: null operand in problem %d", i);
  WbGroup g;
  int blocks, tiles;
  const int mode = wb_mode(d, count);
  wb_plan(d, count, g, blocks, tiles, cus, mode);
  SITK_REQUIRE((size_t)blocks * (WB_TILE_ELEMS + 2 * 192) * sizeof(float) <= ws_bytes, "gemm_wgrad_group_ws: workspace of %zu bytes, "
               "%d blocks planned for %d CUs", ws_bytes, blocks, cus);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (mode == WB_WIDE) hipLaunchKernelGGL(wgrad_x2_kernel<false>, dim3(blocks / 2), dim3(256), 0, s, g, reinterpret_cast<float*>(ws));
  else hipLaunchKernelGGL(wgrad_x2_kernel<true>, dim3(blocks / 2), dim3(256), 0, s, g, reinterpret_cast<float*>(ws));
  SITK_LAUNCH_CHECK("wgrad_big");
  bool any_split = false;
  for (int i = 0; i < count; ++i) any_split |= g.p[i].splits > 1;
  if (!any_split) return SITK_OK;                          // every tile covered all its tokens and went straight to dW
  hipLaunchKernelGGL(wgrad_big_reduce_kernel, dim3(WB_TILE_ELEMS / 4 / 256, tiles), dim3(256), 0, s, g,
                     reinterpret_cast<const float*>(ws), tiles, blocks);
  return check_launch("wgrad_big_reduce");
}
