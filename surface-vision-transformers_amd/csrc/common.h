// sitk -- common device/host helpers for the gfx950 (MI355X, CDNA4) SiT kernels.
// Wave = 64 lanes everywhere.  No CUDA compatibility paths: this code targets gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- the 16-bit compute type of THIS translation unit ------------------------------------------------------------
// Every kernel source except core.hip is compiled twice: once with h16 = bf16 (SITK_BF16, the default objects) and once
// with -DSITK_TU_F16, h16 = IEEE half (SITK_F16, *.f16.o).  Same MFMA rate (v_mfma_f32_16x16x32_{bf16,f16}), same bytes,
// three more mantissa bits -- the mode that meets north_star's 1e-3 (bf16: one 2^-9 rounding per operand) -- at the
// price of a narrow exponent: backward runs on a loss-scaled gradient stream (engine.py, functional.py).
// In the f16 objects every exported name carries the suffix __f16 (f16_names.h) and `namespace sitk` is sitk_f16; the
// public entry points (the bf16 objects) forward there when called with dtype == SITK_F16 (SITK_FORWARD_F16).
#ifdef SITK_TU_F16
#include "f16_names.h"
#define sitk sitk_f16
#endif
#include "../../include/sitk.h"

#ifdef SITK_TU_F16
typedef _Float16 h16;
#define SITK_H16 SITK_F16
#define SITK_H16_MNEMONIC "f16"
#else
typedef __bf16 h16;
#define SITK_H16 SITK_BF16
#define SITK_H16_MNEMONIC "bf16"
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SITK_WAVE 64
#define SITK_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------
// host-side error plumbing (thread-local message, negative return codes)
// ------------------------------------------------------------------------------------------
namespace sitk_rt {   // core.hip (compiled once): one thread-local message for both sets of objects
void set_error(const char* fmt, ...);
int check_launch(const char* what);
}  // namespace sitk_rt
namespace sitk {
using sitk_rt::check_launch;
using sitk_rt::set_error;
}  // namespace sitk

// Public entry point `fn` called with dtype == SITK_F16: hand over to its twin in the f16 objects (same signature).
// SITK_F16_TWIN(fn) declares the twin at file scope; in the f16 objects both expand to nothing.
#ifdef SITK_TU_F16
#define SITK_F16_TWIN(fn)
#define SITK_FORWARD_F16(dt, fn, ...)
#else
#define SITK_F16_TWIN(fn) extern "C" decltype(fn) fn##__f16;
#define SITK_FORWARD_F16(dt, fn, ...) \
  do {                                \
    if ((dt) == SITK_F16) return fn##__f16(__VA_ARGS__); \
  } while (0)
#endif

#define SITK_REQUIRE(cond, ...)                      \
  do {                                               \
    if (!(cond)) {                                   \
      sitk::set_error(__VA_ARGS__);                  \
      return SITK_ERR_INVALID;                       \
    }                                                \
  } while (0)

#define SITK_LAUNCH_CHECK(what)                      \
  do {                                               \
    int _e = sitk::check_launch(what);               \
    if (_e) return _e;                               \
  } while (0)

#define SITK_TRY(expr)                               \
  do {                                               \
    int _e = (expr);                                 \
    if (_e) return _e;                               \
  } while (0)

// internal (cross-translation-unit) helpers of the LayerNorm backward used by the encoder
namespace sitk {
constexpr int LN_FINALIZE_MAX = 32;
struct LnFinalizeEntry {
  const float* partials;
  float* dgamma;
  float* dbeta;
  int nblocks;  // workgroups that wrote `partials` (0: the LayerNorm backward kernel's own grid)
};
struct LnFinalizeBatch {
  LnFinalizeEntry e[LN_FINALIZE_MAX];
};
// row/parameter-gradient kernel only: per-workgroup dgamma/dbeta sums go to `partials`
int layernorm_bwd_deferred(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* dres, float* dx_out, void* dx_out_c, float* partials, int64_t rows, int D, int dtype,
                           hipStream_t s);
// one launch reducing every deferred entry into its dgamma/dbeta
int layernorm_finalize_multi(const LnFinalizeEntry* entries, int count, int64_t rows, int D, hipStream_t s);
}  // namespace sitk

// A/B switches of the measurement tools (tools/gpu_tt1.sh, tools/mlp_stamps.py): environment variables select kernel
// variants ONLY in a diagnostic build (make AB=1 -> -DSITK_AB, a separate libsitk_ab.so picked through SITK_LIB); the
// shipped library reads no environment and always returns the default.
#ifdef SITK_AB
#include <cstdlib>
static inline int sitk_ab_switch(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static inline int sitk_ab_switch(const char*, int dflt) { return dflt; }
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
SITK_DEV int lane_id() { return threadIdx.x & 63; }

SITK_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
SITK_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in every lane: four v_add_f32 with a
// row_ror DPP operand.  (__shfl_xor compiles to ds_bpermute: one LDS round trip per step, and a LayerNorm
// needs 8 dependent ones per row group.)
SITK_DEV float row16_sum(float v) {
#define SITK_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
  v += SITK_ROR(8);
  v += SITK_ROR(4);
  v += SITK_ROR(2);
  v += SITK_ROR(1);
#undef SITK_ROR
  return v;
}

SITK_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
SITK_DEV float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Bijective XCD-aware block remap (8 XCDs, blocks dealt round-robin): logical ids that are
// consecutive land on the same XCD so that neighbouring tiles share that XCD's L2.
SITK_DEV int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Row remapping of a 2-D operand: physical_row(m) = (m / group) * stride + offset + m % group
// (group == 0: identity).  Lets GEMMs read/write "tokens 1..P of every sample" in place.
namespace sitk {
struct RowMap {
  int group, stride, offset;
};
SITK_DEV int map_row(const RowMap& r, int m) {
  return r.group ? (m / r.group) * r.stride + r.offset + (m % r.group) : m;
}
static inline RowMap to_rowmap(const sitk_rowmap& r) { return RowMap{r.group, r.stride, r.offset}; }
}  // namespace sitk

// ------------------------------------------------------------------------------------------
// LDS tile image: rows of 128 bytes, 32-byte windows XOR-swizzled by a key of the row.
// With key(row) = bit1(row) | bit3(row)<<1 the image is conflict-free BOTH for ds_read_b128 row
// reads by 16 consecutive rows (MFMA 16x16 operand, lane -> row l&15, 16-byte chunk l>>4) AND
// for ds_read_b64_tr_b16 transposed reads of 4-row blocks taken 8 rows apart by the two 16-lane
// groups of a half wave (bank = (addr/4) % 64 for both instructions).  The base must be
// 256-byte aligned.  Chunks of <= 32 bytes aligned to their size stay contiguous.
// ------------------------------------------------------------------------------------------
SITK_DEV int lds_off(int row, int byte_in_row) {
  const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
  return row * 128 + (byte_in_row ^ (key << 5));
}

// ------------------------------------------------------------------------------------------
// MFMA traits.  An "mma step" consumes one 16-byte vector per lane from each operand:
//   h16 : v_mfma_f32_16x16x32_{bf16,f16}, K = 32; lane l holds k = 8*(l>>4) + j, j = 0..7
//   f32 : 4 x v_mfma_f32_16x16x4_f32, K = 16; lane l holds k = 4*(l>>4) + j and the j-th
//         instruction contracts element j of every lane (same k on both operands, so the
//         permuted k order is consistent).  Exact f32 (fma chain), 1/16 of the bf16 rate:
//         this is the verification / parity mode.
// In both cases a step spans 64 BYTES of an operand row, so tile code is written in bytes.
// D = A * B with A rows on the output's register axis: acc[jj] <-> (row = 4*(l>>4)+jj, col = l&15).
// ------------------------------------------------------------------------------------------
template <typename T>
struct Mma;

template <>
struct Mma<h16> {
  static constexpr int EPV = 8;     // elements per 16-byte vector
  static constexpr int KSTEP = 32;  // contraction elements per step
  static SITK_DEV f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
#ifdef SITK_TU_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
#endif
  }
};

template <>
struct Mma<float> {
  static constexpr int EPV = 4;
  static constexpr int KSTEP = 16;
  static SITK_DEV f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
    return c;
  }
};

// ---- element conversion / packing ---------------------------------------------------------
template <typename T>
SITK_DEV T from_f32(float v);
template <>
SITK_DEV float from_f32<float>(float v) { return v; }
template <>
SITK_DEV h16 from_f32<h16>(float v) { return (h16)v; }

SITK_DEV float to_f32(float v) { return v; }
SITK_DEV float to_f32(h16 v) { return (float)v; }

// store 4 consecutive elements (8 or 16 bytes)
SITK_DEV void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
SITK_DEV void store4(h16* p, f32x4 v) {
  h16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (h16)v[i];
  *reinterpret_cast<h16x4*>(p) = o;
}
SITK_DEV f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
SITK_DEV f32x4 load4(const h16* p) {
  const h16x4 v = *reinterpret_cast<const h16x4*>(p);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
  return o;
}

// Load one 16-byte operand vector (EPV elements of T) from memory holding TS elements
// (TS == T: plain 16-byte load; TS == float, T == bf16: two 16-byte loads + convert).
template <typename T, typename TS>
struct VecLoad;
template <typename T>
struct VecLoad<T, T> {
  static SITK_DEV u32x4 load(const T* p) { return *reinterpret_cast<const u32x4*>(p); }
};
template <>
struct VecLoad<h16, float> {
  static SITK_DEV u32x4 load(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    h16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = (h16)a[i]; o[i + 4] = (h16)b[i]; }
    return __builtin_bit_cast(u32x4, o);
  }
};

// Pack two accumulator tiles (8 floats) into one operand vector for the NEXT mma whose
// contraction runs over the accumulators' register axis (see attention kernels):
//   bf16: elements j=0..3 from `lo`, j=4..7 from `hi`
template <typename T>
struct Dtype;
template <>
struct Dtype<float> { static constexpr int code = SITK_F32; };
template <>
struct Dtype<h16> { static constexpr int code = SITK_H16; };
