// Micro-benchmark: can ONE wave keep the matrix pipe and the vector ALU busy at the same time?
//   mode 0: 12 MFMAs, then NV independent VALU ops (v_fma_f32), per iteration (phases back to back, as a barrier-locked loop runs them)
//   mode 1: the same instructions interleaved: one MFMA, then NV / 12 VALU ops
//   mode 2: MFMAs only      mode 3: VALU only
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/coexec.hip -o build/coexec ; run: build/coexec <waves per SIMD>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c0 = 1.0001f, c1 = 0.0003f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 8]) : "v"(c0), "v"(c1));
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV / 12; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * (NV / 12) + j) % 8]) : "v"(c0), "v"(c1));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);    // slowest wave of workgroup 0
}

template <int MODE, int NV>
static void run(const char* name, int wps, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s NV=%3d  %6.1f cycles / iteration (slowest wave)\n", name, NV, (double)h / iters);
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  printf("waves per SIMD: %d (12 MFMA 16x16x32 bf16 = 192 matrix-pipe cycles per wave and iteration)\n", wps);
  run<2, 0>("MFMA only", wps, out, cyc);
  run<3, 48>("VALU only", wps, out, cyc);
  run<3, 96>("VALU only", wps, out, cyc);
  run<0, 48>("MFMA phase then VALU phase", wps, out, cyc);
  run<1, 48>("interleaved", wps, out, cyc);
  run<0, 96>("MFMA phase then VALU phase", wps, out, cyc);
  run<1, 96>("interleaved", wps, out, cyc);
  return 0;
}
