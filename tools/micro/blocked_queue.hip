// What does a stream that sits BLOCKED behind an event (a barrier packet in its hardware queue) cost a chain of dependent
// kernels on another stream?  (round 6: the data-parallel step's "+0.7 ms whatever the stand-in does" with the stand-in on a
// stream of its own, +2.2 ms with GPU_MAX_HW_QUEUES=8, +0 with the same operations on the bucket stream: profiles/r05_dp_*.)
//
// The shape of the tiny step: a chain of NK dependent ~20-us kernels of 214 one-per-CU workgroups on the main stream (the null
// stream, as under torch), four ~260-us kernels of 40 workgroups on a lowest-priority side stream, each forked from the chain by
// an event and followed by a "done" event.  Behind the host's enqueue of all of that, per done event i, one of:
//   none       nothing
//   wait       X[j] waits for done[i], records an event; main waits for it at the end                  (events alone)
//   kernel     X[j] waits for done[i], runs a 16-workgroup 1-us kernel, records; main waits at the end (the bucket stream)
//   hop        X[j] waits for done[i], records e0; X[j2] waits for e0, runs the kernel, records; ...   (a stream of its own)
// for every extra stream j (and j2 = j + 1) of NX normal-priority streams created up front -- ROCm maps streams onto few
// hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order, so WHICH stream blocks matters if queue sharing is the cause.
// Prints the chain's duration (event to event on the main stream) and the whole step's.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/blocked_queue.hip -o build/blocked_queue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void busy_kernel(float* p, int iters) {
    extern __shared__ float lds[];
    float v = p[threadIdx.x & 255];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    if (v == 123.f) { lds[threadIdx.x] = v; p[0] = lds[0]; }
}

int main(int argc, char** argv) {
    const int NX = argc > 1 ? atoi(argv[1]) : 8;           // extra normal-priority streams
    const int use_null = argc > 2 ? atoi(argv[2]) : 1;     // 1: the chain runs on the null stream (torch's default stream)
    const int ndummy = argc > 3 ? atoi(argv[3]) : 0;       // lowest-priority streams created (and kept) in front of the X's: each
                                                           // takes a hardware queue of its own, so the X's queues are created LATER
    float* p;
    CHECK(hipMalloc(&p, 4096));
    CHECK(hipMemset(p, 0, 4096));
    CHECK(hipFuncSetAttribute((const void*)busy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    int lo, hi;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t mainS = nullptr, side;
    if (!use_null) CHECK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
    std::vector<hipStream_t> dummy(ndummy);
    for (auto& s : dummy) { CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo)); busy_kernel<<<1, 64, 0, s>>>(p, 10); }
    std::vector<hipStream_t> X(NX);
    for (auto& s : X) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1, t2, fork[4], done[4], e0[4], e1[4];
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1)); CHECK(hipEventCreate(&t2));
    for (int i = 0; i < 4; ++i) {
        CHECK(hipEventCreateWithFlags(&fork[i], hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&e0[i], hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&e1[i], hipEventDisableTiming));
    }
    const int NK = 110, chain_iters = 1100, side_iters = 45000;
    // warm every stream's queue once
    for (auto s : X) busy_kernel<<<1, 64, 0, s>>>(p, 10);
    busy_kernel<<<1, 64, 0, side>>>(p, 10);
    CHECK(hipDeviceSynchronize());
    // "alone": the chain with NO side stream at all; "sidefree": the four side kernels run, but unforked (enqueued up front, nothing
    // parked behind an event of the chain) -- what the side stream's own parked waits cost the chain
    const char* all_modes[] = {"none", "wait", "kernel", "hop", "alone", "sidefree"};
    std::vector<const char*> modes(all_modes, all_modes + (argc > 4 ? atoi(argv[4]) : 4));     // argv[4] = 2: "none" and "wait" only
    if (argc > 5 && atoi(argv[5])) modes = {"none", "alone", "sidefree"};
    printf("# %d extra streams, %d dummy low-priority streams in front of them, chain on the %s stream, GPU_MAX_HW_QUEUES=%s\n", NX, ndummy, use_null ? "null" : "created",
           getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
    for (const char* mode : modes) {
        const bool alone = !strcmp(mode, "alone"), sidefree = !strcmp(mode, "sidefree");
        const bool none = !strcmp(mode, "none") || alone || sidefree;
        for (int j = 0; j < (none ? 1 : NX); ++j) {
            const int j2 = (j + 1) % NX;
            float best_chain = 1e9f, best_all = 1e9f, sum_chain = 0.f;
            const int reps = 8;
            for (int rep = 0; rep < reps; ++rep) {
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(t0, mainS));
                int nfork = 0;
                for (int k = 0; k < NK; ++k) {
                    busy_kernel<<<214, 768, 150 * 1024, mainS>>>(p, chain_iters);
                    if (!alone && k >= 50 && (k - 50) % 12 == 11 && nfork < 4) {      // four forks through the second half of the chain
                        if (!sidefree) {
                            CHECK(hipEventRecord(fork[nfork], mainS));
                            CHECK(hipStreamWaitEvent(side, fork[nfork], 0));
                        }
                        busy_kernel<<<40, 256, 150 * 1024, side>>>(p, side_iters);
                        CHECK(hipEventRecord(done[nfork], side));
                        ++nfork;
                    }
                }
                CHECK(hipEventRecord(t1, mainS));
                // the collectives' side of the step: issued behind the host's enqueue of the whole chain
                if (!none) {
                    for (int i = 0; i < nfork; ++i) {
                        CHECK(hipStreamWaitEvent(X[j], done[i], 0));
                        if (!strcmp(mode, "wait")) {
                            CHECK(hipEventRecord(e1[i], X[j]));
                        } else if (!strcmp(mode, "kernel")) {
                            busy_kernel<<<16, 256, 19744, X[j]>>>(p, 60);
                            CHECK(hipEventRecord(e1[i], X[j]));
                        } else {
                            CHECK(hipEventRecord(e0[i], X[j]));
                            CHECK(hipStreamWaitEvent(X[j2], e0[i], 0));
                            busy_kernel<<<16, 256, 19744, X[j2]>>>(p, 60);
                            CHECK(hipEventRecord(e1[i], X[j2]));
                        }
                    }
                    for (int i = 0; i < nfork; ++i) CHECK(hipStreamWaitEvent(mainS, e1[i], 0));
                } else {
                    for (int i = 0; i < nfork; ++i) CHECK(hipStreamWaitEvent(mainS, done[i], 0));
                }
                busy_kernel<<<214, 768, 150 * 1024, mainS>>>(p, chain_iters);     // the optimizer pass
                CHECK(hipEventRecord(t2, mainS));
                CHECK(hipEventSynchronize(t2));
                CHECK(hipDeviceSynchronize());
                float c, a;
                CHECK(hipEventElapsedTime(&c, t0, t1));
                CHECK(hipEventElapsedTime(&a, t0, t2));
                if (rep >= 2) sum_chain += c;
                if (c < best_chain) best_chain = c;
                if (a < best_all) best_all = a;
            }
            printf("%-7s blocked stream X[%d]%s: chain %8.1f us (mean %8.1f), step %8.1f us\n", mode, j,
                   !strcmp(mode, "hop") ? " -> X[j+1]" : "          ", best_chain * 1e3, sum_chain / (reps - 2) * 1e3, best_all * 1e3);
        }
    }
    return 0;
}
