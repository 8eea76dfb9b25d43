// HBM read / write / mixed streaming rates of one MI355X with the grid shapes of the fused forward kernels
// (hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_rw.hip -o gpurun_out/hbm_rw && gpurun_out/hbm_rw).
// Buffers rotate through 1.5 GB so that neither L2 nor the 256 MB Infinity Cache holds a launch's bytes.
// mode: bytes read per 16 B written = R : W.  Every thread streams 16-B pieces, a wave 1 KB contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, int W>
__global__ void stream_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16, int nt) {
    // n16 = 16-B pieces of the smaller side's unit; a "unit" = R pieces read + W pieces written
    const long stride = (long)gridDim.x * blockDim.x;
    uint4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint4 v = src[i + (long)r * n16];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint4 o = acc; o.x += w;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            v4u ov = {o.x, o.y, o.z, o.w};
            if (nt) __builtin_nontemporal_store(ov, (v4u*)&dst[i + (long)w * n16]);
            else *(v4u*)&dst[i + (long)w * n16] = ov;
        }
    }
    if (W == 0 && acc.x == 0x12345678u) dst[0] = acc;
}

template <int R, int W>
static void run(const char* name, int wgs, int threads, int nt, char* pool, size_t pool_bytes, size_t unit_bytes) {
    // unit_bytes per side-piece: reads R * unit_bytes, writes W * unit_bytes per launch
    const long n16 = unit_bytes / 16;
    const size_t per = (size_t)(R + W) * unit_bytes;
    const int nrot = (int)(pool_bytes / per);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 40;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) {
            char* base = pool + (size_t)(i % nrot) * per;
            stream_kernel<R, W><<<wgs, threads>>>((const uint4*)base, (uint4*)(base + (size_t)R * unit_bytes), n16, nt);
        }
        if (pass) CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-28s grid %5d x %4d%s  read %6.1f MB  write %6.1f MB  %7.1f us  %5.2f TB/s\n", name, wgs, threads, nt ? " nt" : "   ",
           R * unit_bytes / 1e6, W * unit_bytes / 1e6, us, per / us / 1e6);
}

int main() {
    const size_t pool_bytes = (size_t)3 << 29;   // 1.5 GB
    char* pool;
    CHECK(hipMalloc(&pool, pool_bytes));
    CHECK(hipMemset(pool, 1, pool_bytes));
    const size_t MB = 1 << 20;
    int grids[][2] = {{214, 768}, {256, 1024}, {2048, 256}, {8192, 256}};
    for (auto& g : grids) {
        run<1, 0>("read only", g[0], g[1], 0, pool, pool_bytes, 160 * MB);
        run<0, 1>("write only", g[0], g[1], 0, pool, pool_bytes, 160 * MB);
        run<0, 1>("write only", g[0], g[1], 1, pool, pool_bytes, 160 * MB);
        run<1, 1>("copy 1 : 1", g[0], g[1], 0, pool, pool_bytes, 80 * MB);
        run<1, 5>("read 1 : write 5 (fwd tail)", g[0], g[1], 0, pool, pool_bytes, 26 * MB);
        run<1, 5>("read 1 : write 5 (fwd tail)", g[0], g[1], 1, pool, pool_bytes, 26 * MB);
        run<3, 1>("read 3 : write 1 (backward)", g[0], g[1], 0, pool, pool_bytes, 40 * MB);
    }
    return 0;
}
