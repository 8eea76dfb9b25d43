// sitk core: error reporting and ABI bookkeeping.
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "common.h"

namespace sitk_rt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Launch-time errors only (bad configuration, missing code object): never synchronises.
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SITK_ERR_LAUNCH;
  }
  return SITK_OK;
}

}  // namespace sitk_rt

// ---- timeline: HIP events between the launches of a chain (profiling aid, see sitk.h) ----
struct sitk_timeline {
  std::vector<hipEvent_t> ev;
  std::vector<const char*> label;
  int n = 0;
};
extern "C" sitk_timeline* sitk_timeline_create(int capacity) {
  if (capacity < 2) return nullptr;
  sitk_timeline* t = new sitk_timeline;
  t->ev.resize(capacity);
  t->label.assign(capacity, "");
  for (int i = 0; i < capacity; ++i)
    if (hipEventCreate(&t->ev[i]) != hipSuccess) { sitk::set_error("timeline: hipEventCreate failed"); t->ev.resize(i); sitk_timeline_destroy(t); return nullptr; }
  return t;
}
extern "C" void sitk_timeline_destroy(sitk_timeline* t) {
  if (!t) return;
  for (hipEvent_t e : t->ev) (void)hipEventDestroy(e);
  delete t;
}
extern "C" void sitk_timeline_reset(sitk_timeline* t) { if (t) t->n = 0; }
extern "C" int sitk_timeline_mark(sitk_timeline* t, const char* label, sitk_stream_t stream) {
  if (!t) return SITK_OK;
  if (t->n >= (int)t->ev.size()) return SITK_OK;                 // full: further marks are dropped
  if (hipEventRecord(t->ev[t->n], reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    sitk::set_error("timeline: hipEventRecord failed");
    return SITK_ERR_LAUNCH;
  }
  t->label[t->n++] = label;
  return SITK_OK;
}
extern "C" int sitk_timeline_read(sitk_timeline* t, float* us, const char** labels, int max) {
  if (!t || !us || !labels) return SITK_ERR_INVALID;
  if (t->n < 2) return 0;
  if (hipEventSynchronize(t->ev[t->n - 1]) != hipSuccess) { sitk::set_error("timeline: hipEventSynchronize failed"); return SITK_ERR_LAUNCH; }
  int k = 0;
  for (int i = 0; i + 1 < t->n && k < max; ++i, ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]) != hipSuccess) { sitk::set_error("timeline: hipEventElapsedTime failed"); return SITK_ERR_LAUNCH; }
    us[k] = ms * 1e3f;
    labels[k] = t->label[i + 1];
  }
  return k;
}

// ---- overlap: a side stream and the events that tie it to the caller's stream (sitk.h) ----
struct sitk_overlap {
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ev;     // [0, max_layers): forks of sitk_encoder_bwd_overlap; max_layers: its chain / join event;
                                  // max_layers + 1: sitk_overlap_fork / _join
  std::vector<hipEvent_t> done;   // [i]: recorded on the side stream behind side launch i (+ its slab reduction) of the last
                                  // sitk_encoder_bwd_overlap call: the gradients that launch wrote are final there
  int max_layers = 0, layers = 0, cus = 0, caller_joins = 0, n_done = 0, tail_cus = 256, group = 2;
};
extern "C" sitk_overlap* sitk_overlap_create(int max_layers, int cus, int caller_joins) {
  if (max_layers < 1 || max_layers > 64 || cus < 1 || cus > 128) { sitk_rt::set_error("overlap: bad arguments"); return nullptr; }
  sitk_overlap* o = new sitk_overlap;
  o->max_layers = o->layers = max_layers; o->cus = cus; o->caller_joins = caller_joins != 0;
  // LOWEST stream priority: (a) the runtime keeps streams of different priorities on different hardware queues -- a side stream of
  // the default priority can land on the queue of the caller's stream (seen with RCCL's streams in the process: every kernel of
  // the step on one queue, the "side" launches in line with the chain) --, (b) work beside the chain should never win a CU from it
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (hipStreamCreateWithPriority(&o->side, hipStreamNonBlocking, prio_least) != hipSuccess &&
      hipStreamCreateWithFlags(&o->side, hipStreamNonBlocking) != hipSuccess) {
    sitk_rt::set_error("overlap: hipStreamCreate failed");
    delete o;
    return nullptr;
  }
  o->ev.resize(max_layers + 2);
  for (size_t i = 0; i < o->ev.size(); ++i)
    if (hipEventCreateWithFlags(&o->ev[i], hipEventDisableTiming) != hipSuccess) {
      sitk_rt::set_error("overlap: hipEventCreate failed");
      o->ev.resize(i);
      sitk_overlap_destroy(o);
      return nullptr;
    }
  o->done.resize(max_layers);
  for (size_t i = 0; i < o->done.size(); ++i)
    if (hipEventCreateWithFlags(&o->done[i], hipEventDisableTiming) != hipSuccess) {
      sitk_rt::set_error("overlap: hipEventCreate failed");
      o->done.resize(i);
      sitk_overlap_destroy(o);
      return nullptr;
    }
  return o;
}
extern "C" void sitk_overlap_destroy(sitk_overlap* o) {
  if (!o) return;
  for (hipEvent_t e : o->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : o->done) (void)hipEventDestroy(e);
  if (o->side) (void)hipStreamDestroy(o->side);
  delete o;
}
extern "C" sitk_stream_t sitk_overlap_stream(sitk_overlap* o) { return o ? (sitk_stream_t)o->side : nullptr; }
extern "C" int sitk_overlap_set_layers(sitk_overlap* o, int layers) {
  if (!o || layers < 0 || layers > o->max_layers) { sitk_rt::set_error("overlap_set_layers: 0..max_layers"); return SITK_ERR_INVALID; }
  o->layers = layers;
  return SITK_OK;
}
extern "C" int sitk_overlap_set_tail_cus(sitk_overlap* o, int cus) {
  if (!o || cus < 1 || cus > 256) { sitk_rt::set_error("overlap_set_tail_cus: 1..256"); return SITK_ERR_INVALID; }
  o->tail_cus = cus;
  return SITK_OK;
}
extern "C" int sitk_overlap_set_group(sitk_overlap* o, int layers_per_launch) {
  if (!o || layers_per_launch < 1 || layers_per_launch > 3) { sitk_rt::set_error("overlap_set_group: 1..3 layers per side launch"); return SITK_ERR_INVALID; }
  o->group = layers_per_launch;
  return SITK_OK;
}
extern "C" int sitk_overlap_side_launches(const sitk_overlap* o) { return o ? o->n_done : 0; }
extern "C" int sitk_overlap_wait_side_launch(sitk_overlap* o, int i, sitk_stream_t stream) {
  if (!o || i < 0 || i >= o->n_done) { sitk_rt::set_error("overlap_wait_side_launch: launch %d of %d", i, o ? o->n_done : 0); return SITK_ERR_INVALID; }
  if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), o->done[i], 0) != hipSuccess) {
    sitk_rt::set_error("overlap: wait for side launch %d failed", i);
    return SITK_ERR_LAUNCH;
  }
  return SITK_OK;
}
extern "C" int sitk_overlap_fork(sitk_overlap* o, sitk_stream_t stream) {
  if (!o) return SITK_ERR_INVALID;
  hipEvent_t e = o->ev[o->max_layers + 1];
  if (hipEventRecord(e, reinterpret_cast<hipStream_t>(stream)) != hipSuccess || hipStreamWaitEvent(o->side, e, 0) != hipSuccess) {
    sitk_rt::set_error("overlap: fork failed");
    return SITK_ERR_LAUNCH;
  }
  return SITK_OK;
}
extern "C" int sitk_overlap_join(sitk_overlap* o, sitk_stream_t stream) {
  if (!o) return SITK_ERR_INVALID;
  hipEvent_t e = o->ev[o->max_layers + 1];
  if (hipEventRecord(e, o->side) != hipSuccess || hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), e, 0) != hipSuccess) {
    sitk_rt::set_error("overlap: join failed");
    return SITK_ERR_LAUNCH;
  }
  return SITK_OK;
}
// internal accessors for encoder.hip (both sets of objects)
extern "C" __attribute__((visibility("hidden"))) int sitk_overlap_caller_joins_(const sitk_overlap* o) { return o ? o->caller_joins : 0; }
extern "C" __attribute__((visibility("hidden"))) int sitk_overlap_max_layers_(const sitk_overlap* o) { return o ? o->max_layers : 0; }
extern "C" __attribute__((visibility("hidden"))) void* sitk_overlap_stream_(sitk_overlap* o) { return o ? (void*)o->side : nullptr; }
extern "C" __attribute__((visibility("hidden"))) void* sitk_overlap_event_(sitk_overlap* o, int i) { return (o && i >= 0 && i < (int)o->ev.size()) ? (void*)o->ev[i] : nullptr; }
extern "C" __attribute__((visibility("hidden"))) int sitk_overlap_layers_(const sitk_overlap* o) { return o ? o->layers : 0; }
extern "C" __attribute__((visibility("hidden"))) int sitk_overlap_cus_(const sitk_overlap* o) { return o ? o->cus : 0; }
extern "C" __attribute__((visibility("hidden"))) int sitk_overlap_tail_cus_(const sitk_overlap* o) { return o ? o->tail_cus : 256; }
extern "C" __attribute__((visibility("hidden"))) int sitk_overlap_group_(const sitk_overlap* o) { return o ? o->group : 2; }
extern "C" __attribute__((visibility("hidden"))) void sitk_overlap_reset_done_(sitk_overlap* o) { if (o) o->n_done = 0; }
// the event behind the next side launch (null when the object has no room left: cannot happen, one launch holds >= 1 layer)
extern "C" __attribute__((visibility("hidden"))) void* sitk_overlap_next_done_(sitk_overlap* o) {
  return (o && o->n_done < (int)o->done.size()) ? (void*)o->done[o->n_done++] : nullptr;
}


// ---- stream placement probe (ABI 11; sitk.h) ----
// `workgroups` workgroups of 256 threads that hold their CUs for `ticks` of the 100 MHz wall clock (bounded spin).
__global__ __launch_bounds__(256) void sitk_spin_kernel(unsigned long long ticks, int* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int spins = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && spins < (1 << 20)) {
    __builtin_amdgcn_s_sleep(8);
    ++spins;
  }
  if (sink && spins < 0) sink[0] = spins;
}
extern "C" int sitk_stream_probe(sitk_stream_t main_stream, sitk_stream_t candidate, float* chain_free_us, float* chain_blocked_us,
                                 float* candidate_done_us, float* release_us) {
  if (!chain_free_us || !chain_blocked_us || !candidate_done_us || !release_us) { sitk_rt::set_error("stream_probe: null pointer"); return SITK_ERR_INVALID; }
  hipStream_t ms = reinterpret_cast<hipStream_t>(main_stream), cs = reinterpret_cast<hipStream_t>(candidate);
  hipStream_t helper = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr, tc = nullptr, late = nullptr;
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  int rc = SITK_OK;
  auto fail = [&](const char* what) { sitk_rt::set_error("stream_probe: %s failed", what); rc = SITK_ERR_LAUNCH; };
  if (hipStreamCreateWithPriority(&helper, hipStreamNonBlocking, prio_least) != hipSuccess) { fail("hipStreamCreate"); return rc; }
  if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess || hipEventCreate(&tc) != hipSuccess ||
      hipEventCreate(&late) != hipSuccess)
    fail("hipEventCreate");
  const int chain = 128;                                  // dependent launches, one workgroup per CU of 214 CUs, ~11 us each:
                                                          // ~1.4 ms, so the release at 0.9 ms falls INSIDE the chain
  const unsigned long long link_ticks = 1000, late_ticks = 90000;     // 10 us; the helper releases the candidate at 900 us
  // pass 0: warm-up (the first launch of the kernel pays for its code object), discarded; pass 1: the reference -- the chain beside
  // the helper's kernel, the candidate idle; pass 2: the same with the candidate blocked behind the helper's release event
  for (int pass = 0; pass < 3 && rc == SITK_OK; ++pass) {
    if (hipDeviceSynchronize() != hipSuccess) { fail("hipDeviceSynchronize"); break; }
    if (hipEventRecord(t0, ms) != hipSuccess) { fail("hipEventRecord"); break; }
    // the helper holds the release event back for 900 us; it starts with the chain (behind t0)
    if (hipStreamWaitEvent(helper, t0, 0) != hipSuccess) { fail("hipStreamWaitEvent"); break; }
    hipLaunchKernelGGL(sitk_spin_kernel, dim3(1), dim3(256), 0, helper, late_ticks, (int*)nullptr);
    if (hipEventRecord(late, helper) != hipSuccess) { fail("hipEventRecord"); break; }
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(sitk_spin_kernel, dim3(214), dim3(256), 0, ms, link_ticks, (int*)nullptr);
    if (hipEventRecord(t1, ms) != hipSuccess) { fail("hipEventRecord"); break; }
    if (pass == 2) {
      // issued BEHIND the host's enqueue of the chain, as the engine issues its collectives: the candidate sits blocked behind
      // the release event, then runs one small kernel
      if (hipStreamWaitEvent(cs, late, 0) != hipSuccess) { fail("hipStreamWaitEvent"); break; }
      hipLaunchKernelGGL(sitk_spin_kernel, dim3(1), dim3(256), 0, cs, 100ull, (int*)nullptr);
      if (hipEventRecord(tc, cs) != hipSuccess) { fail("hipEventRecord"); break; }
      if (hipStreamWaitEvent(ms, tc, 0) != hipSuccess) { fail("hipStreamWaitEvent"); break; }      // (the caller's stream joins)
    }
    if (hipStreamWaitEvent(ms, late, 0) != hipSuccess) { fail("hipStreamWaitEvent"); break; }      // (... and the helper)
    if (sitk_rt::check_launch("stream_probe") != SITK_OK) { rc = SITK_ERR_LAUNCH; break; }
    if (hipDeviceSynchronize() != hipSuccess) { fail("hipDeviceSynchronize"); break; }
    float ms_chain = 0.f, ms_c = 0.f;
    if (hipEventElapsedTime(&ms_chain, t0, t1) != hipSuccess) { fail("hipEventElapsedTime"); break; }
    if (pass == 1) *chain_free_us = ms_chain * 1e3f;
    else if (pass == 2) {
      *chain_blocked_us = ms_chain * 1e3f;
      if (hipEventElapsedTime(&ms_c, t0, tc) != hipSuccess) { fail("hipEventElapsedTime"); break; }
      *candidate_done_us = ms_c * 1e3f;
    }
  }
  *release_us = (float)late_ticks / 100.f;
  if (t0) (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  if (tc) (void)hipEventDestroy(tc);
  if (late) (void)hipEventDestroy(late);
  (void)hipStreamDestroy(helper);
  return rc;
}

extern "C" int sitk_abi_version(void) { return SITK_ABI_VERSION; }
extern "C" const char* sitk_last_error(void) { return sitk_rt::g_err; }
extern "C" int sitk_dtype_size(int dtype) { return (dtype == SITK_BF16 || dtype == SITK_F16) ? 2 : (dtype == SITK_F32 ? 4 : 0); }

#ifdef SITK_AB
// Diagnostic build only (tools/dp_cu_budget.py): `workgroups` workgroups that hold their CUs for `microseconds` (wall clock,
// s_memrealtime at 100 MHz) -- a stand-in for the channels of a gradient all-reduce whose wire time a one-GPU box cannot
// produce.  The footprint is the one of RCCL's own kernel on this chip (the gfx950 code object inside torch's librccl.so,
// `rcclGenericKernel<*>`: 256 threads, 19 744 B of LDS, 261 - 280 registers per lane, 352 B of scratch), so what can and
// cannot share a CU with it is what can and cannot share a CU with a real channel: nothing of this library's chain (its
// workgroups take 146 KB of LDS or the whole register file).  Bounded: at most 5 ms whatever is asked.
__global__ __launch_bounds__(256) void sitk_debug_occupy_kernel(unsigned long long ticks, int* sink) {
  extern __shared__ int pad[];              // 19 744 B, given at launch (a static array the kernel never reads is dropped)
  pad[threadIdx.x] = (int)threadIdx.x;
  asm volatile("v_mov_b32 v247, 0\n\tv_accvgpr_write_b32 a31, 0" ::: "v247", "a31");     // 248 + 32 registers per lane
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int spins = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && spins < (1 << 22)) {
    __builtin_amdgcn_s_sleep(32);
    ++spins;
  }
  if (sink && spins < 0) sink[0] = pad[(threadIdx.x + 1) & 4095];
}
extern "C" int sitk_debug_occupy(int workgroups, int microseconds, sitk_stream_t stream) {
  if (workgroups < 1 || workgroups > 256 || microseconds < 1) { sitk_rt::set_error("debug_occupy: bad arguments"); return SITK_ERR_INVALID; }
  const unsigned long long ticks = 100ull * (unsigned long long)(microseconds > 5000 ? 5000 : microseconds);
  hipLaunchKernelGGL(sitk_debug_occupy_kernel, dim3(workgroups), dim3(256), 19744, reinterpret_cast<hipStream_t>(stream), ticks,
                     (int*)nullptr);
  return sitk_rt::check_launch("debug_occupy");
}
#endif
