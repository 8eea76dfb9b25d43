// What does a fork / join cost the stream that records (or waits for) the event?  A chain of 24 dependent ~20-us kernels on one
// stream; every 2nd boundary carries one of:
//   none      nothing
//   record    hipEventRecord(ev, main) + hipStreamWaitEvent(side, ev) + a 60-us kernel on the side stream     (the step's fork)
//   flagk     a one-thread kernel on main that stores a sequence number + hipStreamWaitValue32(side, >= seq)   (no event on main)
//   inkernel  the chain kernel's last workgroup stores the sequence number + hipStreamWaitValue32(side)         (nothing on main)
//   join      hipEventRecord(ev, side) + hipStreamWaitEvent(main, ev), side idle                                  (the step's join)
// (hipcc --offload-arch=gfx950 -O3 tools/micro/event_gap.hip -o build/event_gap)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void busy_kernel(float* p, int iters, unsigned* counter, unsigned* flag, unsigned seq) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    if (v == 123.f) p[0] = v;
    if (flag) {                                    // last workgroup out publishes the sequence number
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned n = atomicAdd(counter, 1u);
            if (n == gridDim.x - 1) {
                *counter = 0;
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
__global__ void flag_kernel(unsigned* flag, unsigned seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int main() {
    float* p; unsigned *counter, *flag;
    CHECK(hipMalloc(&p, 4096));
    CHECK(hipMemset(p, 0, 4096));
    CHECK(hipMalloc(&counter, 4));
    CHECK(hipMemset(counter, 0, 4));
    // hipStreamWaitValue32 wants memory the command processor can poll: fine-grained / signal memory
    if (hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) != hipSuccess) {
        printf("(no signal memory: plain device memory for the flag)\n");
        CHECK(hipMalloc(&flag, 8));
    }
    CHECK(hipMemset(flag, 0, 8));
    int lo, hi;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t mainS, side;
    CHECK(hipStreamCreateWithPriority(&mainS, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
    hipEvent_t e0, e1, ev[64];
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int iters = 1100, NK = 24;
    const char* modes[] = {"none", "record_only", "record", "flagk", "inkernel", "join"};
    unsigned seq = 0;
    for (const char* mode : modes) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, mainS));
            for (int k = 0; k < NK; ++k) {
                const bool edge = (k & 1) == 1 && k < NK - 1;
                if (edge && !strcmp(mode, "inkernel")) {
                    ++seq;
                    busy_kernel<<<214, 768, 0, mainS>>>(p, iters, counter, flag, seq);
                    CHECK(hipStreamWaitValue32(side, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
                    busy_kernel<<<4, 256, 0, side>>>(p, 3 * iters, nullptr, nullptr, 0);
                    continue;
                }
                busy_kernel<<<214, 768, 0, mainS>>>(p, iters, nullptr, nullptr, 0);
                if (!edge) continue;
                if (!strcmp(mode, "record_only")) {
                    CHECK(hipEventRecord(ev[k], mainS));
                    CHECK(hipStreamWaitEvent(side, ev[k], 0));
                } else if (!strcmp(mode, "record")) {
                    CHECK(hipEventRecord(ev[k], mainS));
                    CHECK(hipStreamWaitEvent(side, ev[k], 0));
                    busy_kernel<<<4, 256, 0, side>>>(p, 3 * iters, nullptr, nullptr, 0);
                } else if (!strcmp(mode, "flagk")) {
                    ++seq;
                    flag_kernel<<<1, 1, 0, mainS>>>(flag, seq);
                    CHECK(hipStreamWaitValue32(side, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
                    busy_kernel<<<4, 256, 0, side>>>(p, 3 * iters, nullptr, nullptr, 0);
                } else if (!strcmp(mode, "join")) {
                    CHECK(hipEventRecord(ev[k], side));
                    CHECK(hipStreamWaitEvent(mainS, ev[k], 0));
                }
            }
            CHECK(hipEventRecord(e1, mainS));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-9s chain of %d kernels: %8.1f us  (%.2f us per kernel)\n", mode, NK, best * 1e3, best * 1e3 / NK);
    }
    return 0;
}
