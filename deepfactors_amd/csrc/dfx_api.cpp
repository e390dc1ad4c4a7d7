// dfx_api.cpp -- C-ABI of libdfx.so (include/dfx.h): argument validation, host-side pose algebra
// (RelativePose + Jacobians, reference common/algorithm/warping.h:98-137, done once per pair in double),
// scratch management and the launch/copy/sync protocol around the gfx950 kernels.
//
// This is HIP-only product code: there is no CPU fallback and nothing here touches oracle/.
#include "../../include/dfx.h"
#include "dfx_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define DFX_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(DFX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));   \
  } while (0)

constexpr int kStageSlots = 8;

// ---- images owned through the library (dfx_img_alloc) --------------------------------------------------------------------------
// The library knows every writer of such an image (fill / upload / the kernels it launches into it), which is what lets a valid0 map
// carry a SHADOW: one bit per pixel, set = "the pixel is known to hold 1.0" (dfx_sfm_step.hip).  The SfM step then reads 8 bytes per
// 64-pixel chunk instead of the map's 256 to learn that nothing has to be written -- the map is a write-only output of the path
// (dense_sfm.h:161), all ones from the keyframe build on (mapper.cpp:937).  Foreign memory (a caller's own hipMalloc, a torch tensor)
// has writers the library cannot see and always takes the reading variant.  Shadow allocation: [stamp (8 bytes)][bits: one u64 per
// 64 pixels]; the descriptor points at the bits, the stamp word sits right in front of them.
struct ImgRec {
  int device;
  size_t pitch;
  uint32_t w, h;
  size_t elem;
  unsigned long long* shadow;   // null until the image is first used as a valid0 map
  bool uniform;                 // every element holds `value` (fresh allocation, dfx_img_fill_f32): initial state of a shadow created later
  float value;
};
std::mutex g_img_mu;
std::unordered_map<const void*, ImgRec> g_imgs;
std::atomic<int> g_shadow_count{ 0 };
std::atomic<int> g_img_count{ 0 };   // registered images: img_note_write returns at once while there is none
std::atomic<unsigned> g_launch_id{ 0 };

unsigned next_launch_id() {
  unsigned id;
  do { id = g_launch_id.fetch_add(1u, std::memory_order_relaxed) + 1u; } while (id == 0u);
  return id;
}

}  // namespace

struct dfx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t copy_stream = nullptr;          // descriptor uploads of batched launches run beside the previous launch's kernels
  hipEvent_t slot_done[kStageSlots] = {};     // recorded on `stream` behind the kernels that read a slot's device copy of the SfM pair descriptors (pairs_dev)
  bool slot_busy[kStageSlots] = {};
  hipEvent_t sdesc_done[kStageSlots] = {};    // the same for the device copies of the batched SE3 / EvaluateError descriptors (sdesc_dev): own events -- sharing
  bool sdesc_busy[kStageSlots] = {};          // slot_done let a small-operator launch overwrite the event a deferred SfM tail's descriptor region was guarded by
  int cu_count = 0;
  int step_blocks = 0;   // 0 = auto
  int mfma_mode = DFX_MFMA_AUTO;
  int schedule = DFX_SCHEDULE_AUTO;
  int last_dynamic = 0;        // 1 when the last batched SfM step ran the dynamic schedule
  int last_mfma = DFX_MFMA_F32_CHAIN;   // evaluation mode the last SfM / DepthAligner step resolved to
  unsigned* qhead = nullptr;   // dynamic schedule: one item-queue head per pair of a batch (rewound by the finalize kernel)
  size_t qhead_cap = 0;
  bool qhead_dirty = false;    // a dynamic launch failed between its step and its finalize kernel: the heads were not rewound
  unsigned* node_cnt = nullptr;   // graph assembly inside the reduction tail: arrivals per node, zero between launches (k_sfm_tail_b3 rewinds)
  size_t node_cnt_cap = 0;
  bool node_cnt_dirty = false;    // a launch failed after the counters were handed out

  char* partials_base = nullptr;   // allocation (two halves in deferred-tail mode)
  float* partials = nullptr;   // device scratch for workgroup partials
  float* partials_alt = nullptr;   // deferred-tail mode: second half, used by every other batched step
  size_t partials_bytes = 0;   // per half

  // Deferred tail (dfx_set_tail_stream): the reduction tail of a batched SfM step (finalize kernel, then graph assembly) runs on
  // `tail_stream`, beside the step kernel of the NEXT batched launch on `stream`; the two halves of partials / queue heads alternate.
  hipStream_t tail_stream = nullptr;
  int tail_parity = 0;
  hipEvent_t ev_mid[2] = {};    // step kernel done (recorded on `stream`): the tail may start
  hipEvent_t ev_tail[2] = {};   // finalize done (recorded on `tail_stream`): the half may be written again
  bool tail_busy[2] = {};
  hipEvent_t ev_join = nullptr;
  char* items_dev = nullptr;   // device result items (sync API)
  size_t items_bytes = 0;
  dfx::SfmPairDev* pairs_dev = nullptr;   // per stage slot: the pair descriptors of a batch, then (mixed image sizes) its workgroup map
  size_t pairs_cap = 0;        // BYTES per stage slot
  float* code_dev = nullptr;   // 64 floats per stage slot
  float* depth_scratch = nullptr;
  size_t depth_scratch_bytes = 0;
  char* sdesc_dev = nullptr;   // descriptor arrays of the batched SE3 / error launches, one region per stage slot
  size_t sdesc_cap = 0;        // bytes per slot
  dfx::DepthJobDev* jobs_dev = nullptr;   // batched UpdateDepth descriptors, one region per stage slot
  size_t jobs_cap = 0;

  // pinned host staging ring (descriptors / codes going up) and a result area coming down
  char* stage_host = nullptr;
  size_t stage_slot_bytes = 0;
  hipEvent_t stage_ev[kStageSlots] = {};
  hipEvent_t stage_rel_ev[kStageSlots] = {};   // stage_release's events: they only tell the HOST that the kernels / copies in front of them have read a slot -- no
  std::vector<char> pyr_build, pyr_last;       // the descriptor block of the build being enqueued / of the one whose copy sits in pyr_dev
  bool pyr_mirror_valid = false;
  uint32_t pyr_seq = 0;                        // pyramid builds enqueued so far; build q's first kernel stores q into done_flag_host[kPyrStartWord] when it starts
  uint32_t stage_pyr[kStageSlots] = {};        // != 0: the slot was last read by pyramid build stage_pyr[s] -- free once a LATER build is running (no event recorded)
  bool stage_rel[kStageSlots] = {};            // device-written data travels behind them, so they carry no system-scope fence (DFX_STAGE_EVENT_FLAGS); true = the slot's last guard
  bool stage_used[kStageSlots] = {};
  int stage_next = 0;
  char* result_host = nullptr;
  size_t result_bytes = 0;
  // blocking single-result calls: the call's last kernel stores done_seq behind the result and the host polls the word (dfx_kernels.hpp, DoneFlag)
  uint32_t* done_flag_host = nullptr;   // pinned, mapped, coherent
  uint32_t* done_flag_dev = nullptr;
  unsigned* done_counter = nullptr;     // device: arrivals of a multi-workgroup finalize kernel, zero between calls
  uint32_t done_seq = 0;
  bool poll = true;                     // dfx_set_result_wait
  bool simple_zerocopy = true;          // dfx_ctx_configure(DFX_OPT_SIMPLE_DESC_ZEROCOPY)
  bool step_zerocopy = false;           // dfx_ctx_configure(DFX_OPT_STEP_DESC_ZEROCOPY)
  double* fin_scratch = nullptr;        // a single pair's four-workgroups-per-tile finalize kernel (k_sfm_finalize_b3_split): dfx::kSplitScratchBytes
  unsigned* fin_cnt = nullptr;          // + 64 arrival counters, zero between calls
  dfx::DoneFlag* done_armed = nullptr;        // set by a blocking entry around the one impl call whose finalize kernel is to signal
  void* track_state_dev = nullptr;   // [TrackState x n][SimplePairDev x levels x n]
  size_t track_bytes = 0;
  char* sg_dev = nullptr;      // sparse geometric: codes + points + rows
  size_t sg_bytes = 0;
  char* pyr_dev = nullptr;     // pyramid build: per-level descriptors of a large batch
  size_t pyr_bytes = 0;

  // per-camera ray tables of the SfM step kernel: [W] (x - u0) / fx, [H + kRayTabSlack] (y - v0) / fy
  struct RayTab { float fx, fy, u0, v0; uint32_t W, H; float* dev; };
  std::vector<RayTab> ray_tabs;

  // measurement hook (dfx_set_profiling): event pairs around the step kernel
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;   // all created pairs
  size_t prof_used = 0;                                       // pairs recorded since the last read
};

namespace {

int ensure_device(dfx_ctx* c) {
  DFX_HIP(hipSetDevice(c->device));
  return DFX_OK;
}

// Grows a device scratch buffer; callers drain the stream first when the old buffer may still be in use.
// The clear is enqueued on the context's stream so that it is ordered before the kernels that use the buffer.
#ifndef DFX_PYR_DESC_CACHE
#define DFX_PYR_DESC_CACHE 1
#endif
#ifndef DFX_STAGE_EVENT_FLAGS
#define DFX_STAGE_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
int grow_dev(void** p, size_t* cap, size_t need, hipStream_t stream) {
  if (*cap >= need) return DFX_OK;
  if (*p) DFX_HIP(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  size_t n = need + need / 2;
  DFX_HIP(hipMalloc(p, n));
  DFX_HIP(hipMemsetAsync(*p, 0, n, stream));
  *cap = n;
  return DFX_OK;
}

// Measurement hook (dfx_set_profiling): the next pair of events bracketing a launch's main kernel, or nulls when profiling is off
int prof_events(dfx_ctx* c, hipEvent_t* eb, hipEvent_t* ee) {
  *eb = *ee = nullptr;
  if (!c->profiling) return DFX_OK;
  if (c->prof_used == c->prof_pool.size()) {
    hipEvent_t a, b;
    DFX_HIP(hipEventCreate(&a));
    DFX_HIP(hipEventCreate(&b));
    c->prof_pool.emplace_back(a, b);
  }
  *eb = c->prof_pool[c->prof_used].first;
  *ee = c->prof_pool[c->prof_used].second;
  c->prof_used++;
  return DFX_OK;
}

// Workgroup-partials scratch.  Growing drains the stream first (the old
// buffer may be in use); the clear is ordered on the context's stream in front of the kernels that use the buffer.
// `for_step` = false (every user but the batched SfM step): the caller is about to write half 0 from the context's stream, so that stream
// first waits for deferred tails still reading it.
int grow_partials(dfx_ctx* c, size_t need, bool for_step = false) {
  if (c->partials_bytes < need) {
    DFX_HIP(hipStreamSynchronize(c->stream));
    if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
    c->tail_busy[0] = c->tail_busy[1] = false;
    if (c->partials_base) DFX_HIP(hipFree(c->partials_base));
    c->partials_base = nullptr; c->partials = c->partials_alt = nullptr; c->partials_bytes = 0;
    const size_t n = (need + need / 2 + 255) & ~(size_t)255, half = n;
    const int halves = c->tail_stream ? 2 : 1;
    DFX_HIP(hipMalloc((void**)&c->partials_base, half * halves));
    DFX_HIP(hipMemsetAsync(c->partials_base, 0, half * halves, c->stream));
    c->partials = reinterpret_cast<float*>(c->partials_base);
    if (halves == 2) c->partials_alt = reinterpret_cast<float*>(c->partials_base + half);
    c->partials_bytes = n;
  }
  if (!for_step && c->tail_busy[0]) { DFX_HIP(hipStreamWaitEvent(c->stream, c->ev_tail[0], 0)); c->tail_busy[0] = false; }
  return DFX_OK;
}

// Zero-copy descriptors: the kernels of a batched launch read their descriptor array straight out of the pinned staging slot (host memory,
// mapped) instead of a device copy -- no copy, no event between the copy stream and the launch stream; the slot is free again behind the
// kernels.  Measured (profiles/r04_launch_gaps.txt): the batched SE3 step / EvaluateError gain 7 / 3.5 us per call (165.8 -> 158.7,
// 93.0 -> 89.7 us per 128 pairs) -> default for them; the batched SfM step gains 8 us outside its kernel and loses 4 inside it (3840 long-lived
// workgroups read 1.5 MB over PCIe) -> off by default.  Both are per-context options (dfx_ctx_configure: DFX_OPT_SIMPLE_DESC_ZEROCOPY,
// DFX_OPT_STEP_DESC_ZEROCOPY); the two descriptor paths give the same bytes (tests/test_gpu_desc_paths.py).
bool simple_zerocopy(const dfx_ctx* c) { return c->simple_zerocopy; }
bool desc_zerocopy(const dfx_ctx* c) { return c->step_zerocopy; }

// Pinned staging ring: returns a host slot whose previous upload has completed.
// Slots of at least `bytes`.  Growing frees the old ring: never while a slot is handed out and not yet released -- a caller that acquires a second slot before
// releasing the first (dfx_sfm_linearize_batch: the decoder's job list inside the step's preparation) reserves the second one's size up front.
constexpr size_t kPyrTailMaxPixels = 160 * 120;   // a pyramid level up to this size starts the one-launch tail of a build (k_pyr_tail); 320 x 240 -- level 1 of a 640 x 480 build folded in -- measured 28.3 us against 12.5 + 9.2 + a boundary
constexpr size_t kStageSlotMaxBytes = size_t(64) << 20;   // x kStageSlots = 512 MiB of pinned memory at the very most
int stage_reserve(dfx_ctx* c, size_t bytes) {
  if (bytes <= c->stage_slot_bytes) return DFX_OK;
  // drain and regrow
  DFX_HIP(hipStreamSynchronize(c->stream));
  if (c->copy_stream) DFX_HIP(hipStreamSynchronize(c->copy_stream));
  if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
  if (c->stage_host) DFX_HIP(hipHostFree(c->stage_host));
  c->stage_host = nullptr;
  c->stage_slot_bytes = 0;
  // growth headroom: double small slots (descriptor arrays grow with the batch), at most 1 MiB of slack on large ones; and a ceiling -- the ring is
  // kStageSlots slots of PINNED host memory, so one oversized call (65535 sparse-geometric factors with host-resident points ~ 50 MB) must not pin
  // gigabytes for the life of the context: split such a batch
  if (bytes > kStageSlotMaxBytes)
    return fail(DFX_E_INVALID, "a single call stages %zu bytes of descriptors / host payload; the pinned ring takes at most %zu per call: split the batch", bytes, kStageSlotMaxBytes);
  size_t n = bytes + (bytes < (size_t(1) << 20) ? bytes : (size_t(1) << 20));
  if (n < 4096) n = 4096;
  if (n > kStageSlotMaxBytes) n = kStageSlotMaxBytes;
  DFX_HIP(hipHostMalloc((void**)&c->stage_host, n * kStageSlots, hipHostMallocDefault));
  c->stage_slot_bytes = n;
  for (int i = 0; i < kStageSlots; ++i) c->stage_used[i] = false;
  return DFX_OK;
}
constexpr int kPyrStartWord = 16;   // (its own cache line of the 128-byte flag area)
int wait_pyr_started(dfx_ctx* c, uint32_t need);
int stage_acquire(dfx_ctx* c, size_t bytes, int* slot, char** host) {
  int rc;
  if ((rc = stage_reserve(c, bytes))) return rc;
  const int s = c->stage_next;
  c->stage_next = (s + 1) % kStageSlots;
  if (c->stage_used[s]) {
    if (c->stage_pyr[s]) { if ((rc = wait_pyr_started(c, c->stage_pyr[s] + 1))) return rc; }
    else DFX_HIP(hipEventSynchronize(c->stage_rel[s] ? c->stage_rel_ev[s] : c->stage_ev[s]));
  }
  c->stage_pyr[s] = 0;
  *slot = s;
  *host = c->stage_host + (size_t)s * c->stage_slot_bytes;
  return DFX_OK;
}

int stage_release(dfx_ctx* c, int slot) {
  DFX_HIP(hipEventRecord(c->stage_rel_ev[slot], c->stream));
  c->stage_used[slot] = true;
  c->stage_rel[slot] = true;
  c->stage_pyr[slot] = 0;
  return DFX_OK;
}

int ensure_result_host(dfx_ctx* c, size_t bytes) {
  if (c->result_bytes >= bytes) return DFX_OK;
  if (c->result_host) DFX_HIP(hipHostFree(c->result_host));
  c->result_host = nullptr;
  size_t n = bytes * 2;
  if (n < 8192) n = 8192;
  DFX_HIP(hipHostMalloc((void**)&c->result_host, n, hipHostMallocMapped | hipHostMallocCoherent));
  c->result_bytes = n;
  return DFX_OK;
}

// Blocking calls with small results: the finalize kernel stores straight into the pinned, device-mapped result area and
// the call only waits for the stream -- no device->host copy operation behind the kernels.
constexpr size_t kDirectResultMax = 256 * 1024;
int result_target(dfx_ctx* c, size_t bytes, void** dev_ptr) {
  int rc;
  if ((rc = ensure_result_host(c, bytes))) return rc;
  DFX_HIP(hipHostGetDevicePointer(dev_ptr, c->result_host, 0));
  return DFX_OK;
}
int wait_stream(dfx_ctx* c);
int finish_result(dfx_ctx* c, void* host_out, size_t bytes) {
  int rc;
  if ((rc = wait_stream(c))) return rc;
  std::memcpy(host_out, c->result_host, bytes);
  return DFX_OK;
}
// The polled form (DoneFlag in dfx_kernels.hpp) is the default; dfx_set_result_wait(ctx, DFX_WAIT_STREAM) makes every blocking call wait for the stream as above.
// a flag for one call: *d stays empty when polling is off (the launchers then signal nothing and finish_result_polled waits for the stream)
int ensure_done_flag(dfx_ctx* c) {
  if (c->done_flag_dev) return DFX_OK;   // (set last: a half-made set is completed by the next call)
  if (!c->done_counter) {
    DFX_HIP(hipMalloc((void**)&c->done_counter, 64));
    DFX_HIP(hipMemsetAsync(c->done_counter, 0, 64, c->stream));
  }
  if (!c->done_flag_host) {
    DFX_HIP(hipHostMalloc((void**)&c->done_flag_host, 128, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->done_flag_host, 0, 128);
  }
  void* dp = nullptr;
  DFX_HIP(hipHostGetDevicePointer(&dp, c->done_flag_host, 0));
  c->done_flag_dev = static_cast<uint32_t*>(dp);
  return DFX_OK;
}
int ensure_fin_scratch(dfx_ctx* c) {
  if (c->fin_scratch) return DFX_OK;
  if (!c->fin_cnt) {
    DFX_HIP(hipMalloc((void**)&c->fin_cnt, 64 * sizeof(unsigned)));
    DFX_HIP(hipMemsetAsync(c->fin_cnt, 0, 64 * sizeof(unsigned), c->stream));
  }
  DFX_HIP(hipMalloc((void**)&c->fin_scratch, dfx::kSplitScratchBytes));
  return DFX_OK;
}
int new_done_flag(dfx_ctx* c, dfx::DoneFlag* d) {
  *d = dfx::DoneFlag{};
  if (!c->poll) return DFX_OK;
  int rc;
  if ((rc = ensure_done_flag(c))) return rc;
  if (++c->done_seq == 0) c->done_seq = 1;
  d->flag = c->done_flag_dev; d->seq = c->done_seq; d->counter = c->done_counter;
  return DFX_OK;
}
// Spins until *f == seq -- politely: a `pause` per poll (the SMT sibling keeps its issue slots), and for at most kPollSpinUs.  A call that lands behind a queue
// of *_async work (millisecond step kernels) would otherwise hold a core at 100 % for the whole queue time; past the bound the wait turns into
// hipStreamSynchronize, which also surfaces a failed launch or a faulted queue (they never write the word).  A drained stream without the word is an error:
// the kernel that was to write it did not run to its end.
constexpr long long kPollSpinUs = 200;
inline void cpu_relax() {
#if defined(__x86_64__)
  __asm__ __volatile__("pause" ::: "memory");
#elif defined(__aarch64__)
  __asm__ __volatile__("yield" ::: "memory");
#endif
}
int poll_word(dfx_ctx* c, const uint32_t* f, uint32_t seq) {
  using clk = std::chrono::steady_clock;
  clk::time_point t0{};
  bool timed = false;
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == seq) return DFX_OK;
    cpu_relax();
    if ((spins & 0xff) != 0) continue;
    const clk::time_point now = clk::now();
    if (!timed) { t0 = now; timed = true; continue; }
    if (std::chrono::duration_cast<std::chrono::microseconds>(now - t0).count() < kPollSpinUs) continue;
    const hipError_t q = hipStreamSynchronize(c->stream);
    if (q != hipSuccess) return fail(DFX_E_HIP, "stream failed while waiting for a result: %s", hipGetErrorString(q));
    if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == seq) return DFX_OK;
    return fail(DFX_E_HIP, "the stream drained without the result flag (sequence %u)", seq);
  }
}
// A staging slot last read by pyramid build `need - 1` is free once build `need` (or a later one) has STARTED: its first workgroup stores the build's number into
// the flag area (pyr_mirror_descs).  If no such build was enqueued, or it does not start within the polling bound, the stream is drained instead.  (The bound is
// that of a host running a full ring -- kStageSlots - 1 builds -- ahead of the device: draining the stream there would idle the device once per ring.)
constexpr long long kPyrStartSpinUs = 2000;
int wait_pyr_started(dfx_ctx* c, uint32_t need) {
  const uint32_t* f = c->done_flag_host + kPyrStartWord;
  auto started = [&] { return (int32_t)(__atomic_load_n(f, __ATOMIC_ACQUIRE) - need) >= 0; };
  if (started()) return DFX_OK;
  if ((int32_t)(c->pyr_seq - need) >= 0) {
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    for (unsigned spins = 1;; ++spins) {
      if (started()) return DFX_OK;
      cpu_relax();
      if ((spins & 0xff) == 0 && std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count() >= kPyrStartSpinUs) break;
    }
  }
  const hipError_t q = hipStreamSynchronize(c->stream);
  if (q != hipSuccess) return fail(DFX_E_HIP, "stream failed while waiting for a staging slot: %s", hipGetErrorString(q));
  return DFX_OK;
}
// The end of a blocking call that has no kernel of its own to signal (no result, or many last writers).  A word written by the command processor behind the
// stream's work (hipStreamWriteValue32) and polled by the host was measured here too, interleaved with this in one process: the best case is 2.5 us sooner,
// the mean is not (UpdateDepth 22.5-24.0 against 22.5-24.4 us, SobelGradients 14.2 against 12.5-13.4, a 16-pair RunStepBatch 156 against 149;
// profiles/r05_poll_result.txt) -- so these calls wait for the stream.
int wait_stream(dfx_ctx* c) {
  DFX_HIP(hipStreamSynchronize(c->stream));
  return DFX_OK;
}
int finish_result_polled(dfx_ctx* c, void* host_out, size_t bytes, const dfx::DoneFlag& d) {
  if (!d.flag) return finish_result(c, host_out, bytes);
  int rc;
  if ((rc = poll_word(c, c->done_flag_host, d.seq))) return rc;
  std::memcpy(host_out, c->result_host, bytes);
  return DFX_OK;
}

// ---- pose algebra in double ---------------------------------------------------------------------------------
void quat_to_R(const float* q, double* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  if (n > 0) { x /= n; y /= n; z /= n; w /= n; }
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// pose_10 = pose1^-1 * pose0 and the two blocks of its Jacobians (warping.h:105-137, called as
// RelativePose(pose1, pose0, J1, J0) at cu_sfmaligner.cpp:166):  M = R1^T,  HM = hat(R1^T (t1 - t0)) R1^T
void relative_pose(const dfx_se3& p0, const dfx_se3& p1, float* R10, float* t10, float* M, float* HM) {
  double R0[9], R1[9];
  quat_to_R(p0.q, R0);
  quat_to_R(p1.q, R1);
  double Mt[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Mt[i * 3 + j] = R1[j * 3 + i];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += Mt[i * 3 + k] * R0[k * 3 + j];
    R10[i * 3 + j] = (float)s;
  }
  const double d01[3] = { (double)p0.t[0] - p1.t[0], (double)p0.t[1] - p1.t[1], (double)p0.t[2] - p1.t[2] };
  double v[3];
  for (int i = 0; i < 3; ++i) {
    const double s = Mt[i * 3] * d01[0] + Mt[i * 3 + 1] * d01[1] + Mt[i * 3 + 2] * d01[2];
    t10[i] = (float)s;
    v[i] = -s;   // R1^T (t1 - t0)
  }
  const double Hh[9] = { 0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0 };
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += Hh[i * 3 + k] * Mt[k * 3 + j];
    HM[i * 3 + j] = (float)s;
    if (M) M[i * 3 + j] = (float)Mt[i * 3 + j];
  }
}

// K^-1 (x, y, 1) per column / row, evaluated on the host with the IEEE float expressions of ReprojectDepth
// (pinhole_camera_impl.h:77-86) -- bit-identical to the per-pixel evaluation it replaces.  One table per camera
// (pyramid level), created on first use; the upload is synchronous, so it is ordered before any later launch.
int ray_table(dfx_ctx* c, const dfx_cam* cam, uint32_t W, uint32_t H, const float** out) {
  for (const auto& t : c->ray_tabs)
    if (t.fx == cam->fx && t.fy == cam->fy && t.u0 == cam->u0 && t.v0 == cam->v0 && t.W == W && t.H == H) { *out = t.dev; return DFX_OK; }
  const size_t n = (size_t)W + H + dfx::kRayTabSlack;
  std::vector<float> h(n, 0.0f);
  volatile float fx = cam->fx, fy = cam->fy, u0 = cam->u0, v0 = cam->v0;   // volatile: no reciprocal / contraction rewrites
  for (uint32_t x = 0; x < W; ++x) { const float d = (float)x - u0; h[x] = d / fx; }
  for (uint32_t y = 0; y < H; ++y) { const float d = (float)y - v0; h[W + y] = d / fy; }
  float* dev = nullptr;
  DFX_HIP(hipMalloc((void**)&dev, n * sizeof(float)));
  DFX_HIP(hipMemcpy(dev, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  c->ray_tabs.push_back({ cam->fx, cam->fy, cam->u0, cam->v0, W, H, dev });
  *out = dev;
  return DFX_OK;
}

// A caller cycling through cameras: start over once nothing is in flight.  Called BEFORE the descriptors of a batch are filled,
// never in between (a descriptor filled earlier in the same batch would keep a pointer to a freed table).
int ray_table_gc(dfx_ctx* c) {
  if (c->ray_tabs.size() < 256) return DFX_OK;
  DFX_HIP(hipStreamSynchronize(c->stream));
  for (auto& t : c->ray_tabs) (void)hipFree(t.dev);
  c->ray_tabs.clear();
  return DFX_OK;
}

bool img_ok(const dfx_img* im) { return im && im->ptr && im->w > 0 && im->h > 0 && im->pitch_bytes > 0; }

int check_img(const dfx_img* im, const char* name, uint32_t w, uint32_t h, size_t elem_bytes) {
  if (!img_ok(im)) return fail(DFX_E_INVALID, "%s: null or empty image view", name);
  if (im->w != w || im->h != h) return fail(DFX_E_INVALID, "%s: size %ux%u, expected %ux%u", name, im->w, im->h, w, h);
  if (im->pitch_bytes < (size_t)w * elem_bytes) return fail(DFX_E_INVALID, "%s: pitch %zu < row bytes %zu", name, im->pitch_bytes, (size_t)w * elem_bytes);
  if (im->pitch_bytes * (size_t)h >= 0x70000000ull) return fail(DFX_E_INVALID, "%s: image of %zu bytes exceeds the 0x70000000-byte buffer-resource limit", name, im->pitch_bytes * (size_t)h);
  if (((uintptr_t)im->ptr | im->pitch_bytes) & 3) return fail(DFX_E_INVALID, "%s: pointer/pitch not 4-byte aligned", name);
  return DFX_OK;
}

int valid0_shadow(dfx_ctx* c, const dfx_img* v, uint32_t W, uint32_t H, unsigned long long** out);   // below

int fill_sfm_pair(dfx_ctx* c, int cs, const dfx_se3* pose0, const dfx_se3* pose1, const dfx_cam* cam, const dfx_img* img0,
                  const dfx_img* img1, const dfx_img* dpt0, const dfx_img* valid0, const dfx_img* jac, const dfx_img* grad1,
                  uint32_t W, uint32_t H, dfx::SfmPairDev* d) {
  int rc;
  if ((rc = check_img(img0, "img0", W, H, 4))) return rc;
  if ((rc = check_img(img1, "img1", W, H, 4))) return rc;
  if ((rc = check_img(dpt0, "dpt0", W, H, 4))) return rc;
  if ((rc = check_img(grad1, "grad1", W, H, 8))) return rc;
  if ((rc = check_img(jac, "prx0_jac", W * (uint32_t)cs, H, 4))) return rc;
  const size_t jalign = (size_t)(cs / 16) * 4;   // one MFMA-layout vector load per lane
  if (((uintptr_t)jac->ptr | jac->pitch_bytes) & (jalign - 1))
    return fail(DFX_E_INVALID, "prx0_jac: pointer/pitch must be %zu-byte aligned for cs=%d", jalign, cs);
  if (((uintptr_t)grad1->ptr | grad1->pitch_bytes) & 7) return fail(DFX_E_INVALID, "grad1: pointer/pitch must be 8-byte aligned");
  if (valid0 && valid0->ptr) {
    if ((rc = check_img(valid0, "valid0", W, H, 4))) return rc;
    d->valid0 = (float*)valid0->ptr;
    d->pitch_valid0 = (uint32_t)valid0->pitch_bytes;
    if ((rc = valid0_shadow(c, valid0, W, H, &d->valid0_shadow))) return rc;
  } else {
    d->valid0 = nullptr;
    d->pitch_valid0 = 0;
    d->valid0_shadow = nullptr;
  }
  relative_pose(*pose0, *pose1, d->R, d->t, d->M, d->HM);
  if ((rc = ray_table(c, cam, W, H, &d->ray_tab))) return rc;
  d->fx = cam->fx; d->fy = cam->fy; d->u0 = cam->u0; d->v0 = cam->v0; d->w = cam->w; d->h = cam->h;
  d->img0 = (const float*)img0->ptr; d->img1 = (const float*)img1->ptr; d->dpt0 = (const float*)dpt0->ptr;
  d->jac = (const float*)jac->ptr; d->grad1 = (const float*)grad1->ptr;
  d->pitch_img0 = (uint32_t)img0->pitch_bytes; d->pitch_img1 = (uint32_t)img1->pitch_bytes;
  d->pitch_dpt0 = (uint32_t)dpt0->pitch_bytes; d->pitch_jac = (uint32_t)jac->pitch_bytes;
  d->pitch_grad1 = (uint32_t)grad1->pitch_bytes;
  return DFX_OK;
}

// Workgroups per pair for the step kernel, from launch-shape sweeps on MI355X at steady clocks (tools/ab_bench.py --blocks, 3000
// launches of preroll; round 1's sweeps were taken inside the clock ramp and favoured twice as many, shorter workgroups).  A wave's
// prologue + epilogue (descriptor and ray-table loads, cold first loads, the barrier in front of the cross-wave fold) cost about as
// much as two chunks, so waves should be long: 5 chunks per wave while the batch is small (1 / 2 / 4 pairs of 640x480: 240
// workgroups per pair is best), then as many as leave ~1.25 rounds of the 16 * CUs resident waves, up to 30:
//   8 pairs 78 -> 74 us, 16 pairs 148 -> 138 us, 32 pairs 294 -> 258 us, 64 pairs 550 -> 533 us against the old five-round rule.
// Longest wave (chunks) the launch shape hands out.  30 for the fp32 chain (four workgroups per CU: 40 workgroups per 640x480 pair x 128 pairs =
// 5.0 rounds of 1024).  The bf16 split runs three workgroups per CU (147 registers), and with it 40 chunks per wave -- 30 workgroups per pair
// -- measured the same or better at every batch size tried when every pair streams its OWN Jacobian (round 3, profiles/r03_ab_launch_shape.txt:
// 48 pairs -2.3 %, 128 pairs -1.0 % incl. 3 us less reduction tail, 256 pairs -0.5 %, three pyramid levels in one launch -0.5 %, 64 pairs and
// CS = 16 +-0).  Batches whose pairs SHARE a keyframe's Jacobian (a relinearisation round: 120 pairs of 16 keyframes) live on the pairs of a
// keyframe running side by side through the L2 / Infinity Cache, and there the shorter waves are 8 % faster (917 vs 990 us): they keep 30,
// and so does CS = 64.
int chunks_per_wave_max(int cs, bool b3, bool distinct_jacobians) {
  return (b3 && cs < 64 && distinct_jacobians) ? 40 : 30;
}

int auto_step_blocks(const dfx_ctx* c, uint32_t W, uint32_t H, int npairs, int cs, int requested = 0, bool b3 = false, bool distinct_jacobians = false) {
  const int nchunks = (int)(((size_t)W * H + 63) / 64);
  int maxb = (nchunks + 3) / 4;   // one chunk per wave at most
  if (maxb < 1) maxb = 1;
  int b = requested > 0 ? requested : c->step_blocks;
  if (b <= 0) {
    const long long total_chunks = (long long)nchunks * npairs;
    // CS 64 runs two waves per SIMD (215 VGPRs) and its chunks carry 3x the matrix work: 2.5 rounds of its 8 * CUs resident waves
    // measured best (4 pairs 1280x960: 309 us at 15 chunks per wave vs 330 at 5; 16 pairs: 1204 us at 30 vs 1245 at 10)
    const long long resident_waves = (cs >= 64 ? 8LL : 16LL) * c->cu_count;
    int cpw = cs >= 64 ? (int)(2 * total_chunks / (5 * resident_waves)) : (int)(4 * total_chunks / (5 * resident_waves));
    const int cpw_max = chunks_per_wave_max(cs, b3, distinct_jacobians);
    if (cpw < 5) cpw = 5;
    if (cpw > cpw_max) cpw = cpw_max;
    b = (nchunks + 4 * cpw - 1) / (4 * cpw);
    const int cap = (64 * c->cu_count + npairs - 1) / npairs;   // at most 64 workgroups per CU over the whole batch
    if (b > cap) b = cap;
    if (b < 1) b = 1;
  }
  return b < maxb ? b : maxb;
}

int simple_blocks(uint32_t W, uint32_t H) {
  int b = (int)(((size_t)W * H + 255) / 256);
  if (b > dfx::kMaxSimpleBlocks) b = dfx::kMaxSimpleBlocks;
  return b < 1 ? 1 : b;
}

// R10 (optional): the rotation of pose_10 as nine floats -- the callers that get it from relative_pose() (the quaternion of pose_10 is then
// ignored); the constants of the kernels' fast geometry are derived from the rotation the kernels will see.
int fill_simple(dfx_ctx* c, const dfx_se3* pose_10, const dfx_cam* cam, const dfx_img* img0, const dfx_img* img1, const dfx_img* dpt0,
                const dfx_img* grad1, const dfx_img* img2, dfx::SimplePairDev* d, const float* R10 = nullptr) {
  if (!pose_10 || !cam) return fail(DFX_E_INVALID, "null pose/camera");
  if (!img_ok(img0)) return fail(DFX_E_INVALID, "img0: null or empty image view");
  const uint32_t W = img0->w, H = img0->h;
  int rc;
  if ((rc = check_img(img0, "img0", W, H, 4))) return rc;
  if ((rc = check_img(img1, "img1", W, H, 4))) return rc;
  if ((rc = check_img(dpt0, "dpt0", W, H, 4))) return rc;
  // The row walk decides validity in the CAMERA's coordinates (PixelValid against cam.width / cam.height, pinhole_camera_impl.h:105-108) and addresses its
  // taps in the IMAGE: a camera larger than the image would let an inlier tap beyond the image (the buffer load then returns 0 silently).  The reference
  // always pairs a level's camera with that level's image (camera_pyramid.h:41-46); anything else is a caller error, refused here.
  if (cam->w != (float)W || cam->h != (float)H)
    return fail(DFX_E_INVALID, "camera is %gx%g, images are %ux%u: the camera of a pyramid level has the size of that level's images", (double)cam->w, (double)cam->h, W, H);
  double R[9];
  quat_to_R(pose_10->q, R);
  for (int i = 0; i < 9; ++i) d->R[i] = R10 ? R10[i] : (float)R[i];
  d->t[0] = pose_10->t[0]; d->t[1] = pose_10->t[1]; d->t[2] = pose_10->t[2];
  d->fx = cam->fx; d->fy = cam->fy; d->u0 = cam->u0; d->v0 = cam->v0; d->w = cam->w; d->h = cam->h;
  {
    // constants of the fast geometry of the pixel reductions (dfx_kernels.hpp FastGeo), of the fp32 pose the kernels see
    double Rf[9], tf[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (double)d->R[i];
    for (int i = 0; i < 3; ++i) tf[i] = (double)d->t[i];
    dfx::derive_fast_cam(cam->fx, cam->fy, cam->u0, cam->v0, cam->w, cam->h, (int)W, (int)H, &d->fc);
    dfx::derive_fast_geo(Rf, tf, cam->fx, cam->fy, cam->u0, cam->v0, cam->w, cam->h, d->fc, &d->fg);
  }
  d->img0 = (const float*)img0->ptr; d->img1 = (const float*)img1->ptr; d->dpt0 = (const float*)dpt0->ptr;
  d->pitch_img0 = (uint32_t)img0->pitch_bytes; d->pitch_img1 = (uint32_t)img1->pitch_bytes;
  d->pitch_dpt0 = (uint32_t)dpt0->pitch_bytes;
  d->grad1 = nullptr; d->pitch_grad1 = 0; d->img2 = nullptr; d->pitch_img2 = 0;
  if ((rc = ray_table(c, cam, W, H, &d->ray_tab))) return rc;   // K^-1 (x, y, 1) per column / row, staged in LDS by the reduction kernels
  if (grad1) {
    if ((rc = check_img(grad1, "grad1", W, H, 8))) return rc;
    if (((uintptr_t)grad1->ptr | grad1->pitch_bytes) & 7) return fail(DFX_E_INVALID, "grad1: pointer/pitch must be 8-byte aligned");
    d->grad1 = (const float*)grad1->ptr; d->pitch_grad1 = (uint32_t)grad1->pitch_bytes;
  }
  if (img2) {
    if ((rc = check_img(img2, "img2", W, H, 4))) return rc;
    d->img2 = (float*)img2->ptr; d->pitch_img2 = (uint32_t)img2->pitch_bytes;
  }
  return DFX_OK;
}

// copy `bytes` of device results to the host and wait (the reference's blocking copyFrom)
int fetch_result(dfx_ctx* c, const void* dev, void* host_out, size_t bytes) {
  int rc;
  if ((rc = ensure_result_host(c, bytes))) return rc;
  DFX_HIP(hipMemcpyAsync(c->result_host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
  if ((rc = wait_stream(c))) return rc;
  std::memcpy(host_out, c->result_host, bytes);
  return DFX_OK;
}

bool cs_supported(int cs) { return cs == 16 || cs == 32 || cs == 64; }

// DFX_MFMA_AUTO: the evaluation mode the library picks per code size (include/dfx.h; measurements in DESIGN.md section 5)
int resolve_mfma(dfx_ctx* c, int cs) {
  int m = c->mfma_mode;
  if (m == DFX_MFMA_AUTO) m = cs >= DFX_AUTO_BF16X3_MIN_CS ? DFX_MFMA_BF16X3 : DFX_MFMA_F32_CHAIN;
  c->last_mfma = m;
  return m;
}

size_t shadow_words(uint32_t w, uint32_t h) { return ((size_t)w * h + 63) / 64; }

// A writer the library launches (or a fill / upload) is about to change a library-owned image: its shadow, if it has one, forgets
// everything (value != 1) or learns that every pixel is 1.0 (a fill with 1.0).  Ordered on the context's stream like the write itself.
// With a tail stream set, the finalize / tail kernel of an earlier step may still be reading the image (a valid0 map) and REBUILDING its
// shadow on that stream: the writer -- and the shadow's memset -- are ordered behind every tail first (round-3 advisor finding: a refill
// right after an async step could race the rebuild and leave "known 1.0" bits over other data).
int img_note_write(dfx_ctx* c, const dfx_img* im, bool uniform, float value) {
  if (!im || !im->ptr) return DFX_OK;
  if (g_img_count.load(std::memory_order_relaxed) == 0) return DFX_OK;   // no library-owned image exists: nothing can have a record (callers with their own memory pay no lock)
  std::lock_guard<std::mutex> lk(g_img_mu);
  auto it = g_imgs.find(im->ptr);
  if (it == g_imgs.end()) return DFX_OK;
  ImgRec& r = it->second;
  // the record ALWAYS learns about the write (a shadow created later starts from it: an early return here once left `uniform, 1.0` behind
  // an upload, and the first step then took every pixel for "known to hold 1.0"); only the memset needs an existing shadow
  r.uniform = uniform; r.value = value;
  if (r.shadow) {
    if (c->tail_stream && (c->tail_busy[0] || c->tail_busy[1])) {
      DFX_HIP(hipEventRecord(c->ev_join, c->tail_stream));
      DFX_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    DFX_HIP(hipMemsetAsync(r.shadow, (uniform && value == 1.0f) ? 0xFF : 0x00, 8 + 8 * shadow_words(r.w, r.h), c->stream));
  }
  return DFX_OK;
}
int img_note_write(dfx_ctx* c, const dfx_img* im) { return img_note_write(c, im, false, 0.f); }
// many images written by one call (a pyramid build): skipped outright while the library owns no image, else one registry lookup (one lock) per image
int img_note_writes(dfx_ctx* c, const std::vector<const void*>& ptrs) {
  if (g_img_count.load(std::memory_order_relaxed) == 0) return DFX_OK;
  for (const void* p : ptrs) {
    const dfx_img im{ const_cast<void*>(p), 0, 0, 0 };
    int rc;
    if ((rc = img_note_write(c, &im))) return rc;
  }
  return DFX_OK;
}

// The shadow of a valid0 map, created on first use; null for memory the library does not own (or a view that is not the whole image).
int valid0_shadow(dfx_ctx* c, const dfx_img* v, uint32_t W, uint32_t H, unsigned long long** out) {
  *out = nullptr;
  std::lock_guard<std::mutex> lk(g_img_mu);
  auto it = g_imgs.find(v->ptr);
  if (it == g_imgs.end()) return DFX_OK;
  ImgRec& r = it->second;
  if (r.device != c->device || r.elem != 4 || r.w != W || r.h != H || r.pitch != v->pitch_bytes) return DFX_OK;
  if (!r.shadow) {
    const size_t bytes = 8 + 8 * shadow_words(W, H);
    DFX_HIP(hipMalloc((void**)&r.shadow, bytes));
    // the stamp word must never equal a launch id by accident: 0 and 0xFFFFFFFF are not ids (next_launch_id skips 0; 2^32 - 1 launches away)
    DFX_HIP(hipMemsetAsync(r.shadow, (r.uniform && r.value == 1.0f) ? 0xFF : 0x00, bytes, c->stream));
    g_shadow_count.fetch_add(1, std::memory_order_relaxed);
  }
  *out = r.shadow + 1;
  return DFX_OK;
}

}  // namespace

// ---- hooks for the other translation units of the library (dfx_comm.cpp); not exported --------------------------
extern "C" int dfx_internal_fail(int code, const char* msg) { g_last_error = msg ? msg : ""; return code; }
// the stream the results of the batched step become complete on: where an exchange of them has to be enqueued
extern "C" void* dfx_internal_exchange_stream(dfx_ctx* c) { return c->tail_stream ? (void*)c->tail_stream : (void*)c->stream; }
extern "C" int dfx_internal_ctx_device(dfx_ctx* c) { (void)hipSetDevice(c->device); return c->device; }
// the stream the INPUTS of the library's launches are ordered on (a broadcast that rewrites keyframe buffers goes there)
extern "C" void* dfx_internal_main_stream(dfx_ctx* c) { return (void*)c->stream; }
// a collective is about to overwrite device memory: if it is a library-owned image, its record / valid0 shadow learn about the write
extern "C" int dfx_internal_note_write(dfx_ctx* c, void* ptr) {
  const dfx_img im{ ptr, 0, 0, 0 };
  return img_note_write(c, &im);
}

// -------------------------------------------------------------------------------------------------------------
extern "C" {

DFX_API const char* dfx_last_error(void) { return g_last_error.c_str(); }
DFX_API const char* dfx_version(void) { return "dfx 0.1 (gfx950, HIP)"; }

DFX_API int dfx_ctx_create(int device, void* stream, dfx_ctx** out) {
  if (!out) return fail(DFX_E_INVALID, "dfx_ctx_create: out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(DFX_E_NOGPU, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 && (e = hipGetDevice(&device)) != hipSuccess) return fail(DFX_E_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(DFX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
  DFX_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  DFX_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(DFX_E_NOGPU, "libdfx is built for gfx950 only; device %d is %s", device, prop.gcnArchName);
  dfx_ctx* c = new dfx_ctx();
  c->device = device;
  c->cu_count = prop.multiProcessorCount;
  // (no environment variable steers a context: evaluation mode, schedule, wait mode and descriptor paths are set through dfx_set_* / dfx_ctx_configure only)
  // NULL = the device's default stream, on which the reference runs everything (cuda/launch_utils.h:26-32): work is
  // then ordered with any other default-stream producer of the images (e.g. PyTorch ops on its default stream).
  c->stream = (hipStream_t)stream;
  for (int i = 0; i < kStageSlots; ++i) {
    e = hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->stage_rel_ev[i], DFX_STAGE_EVENT_FLAGS);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sdesc_done[i], hipEventDisableTiming);
    if (e != hipSuccess) { dfx_ctx_destroy(c); return fail(DFX_E_HIP, "hipEventCreate failed: %s", hipGetErrorString(e)); }   // destroys what exists so far
  }
  e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { dfx_ctx_destroy(c); return fail(DFX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
  *out = c;
  return DFX_OK;
}

DFX_API void dfx_ctx_destroy(dfx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->tail_stream) (void)hipStreamSynchronize(c->tail_stream);
  for (int i = 0; i < 2; ++i) { if (c->ev_mid[i]) (void)hipEventDestroy(c->ev_mid[i]); if (c->ev_tail[i]) (void)hipEventDestroy(c->ev_tail[i]); }
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->partials_base) (void)hipFree(c->partials_base);
  if (c->items_dev) (void)hipFree(c->items_dev);
  if (c->pairs_dev) (void)hipFree(c->pairs_dev);
  if (c->code_dev) (void)hipFree(c->code_dev);
  if (c->depth_scratch) (void)hipFree(c->depth_scratch);
  if (c->jobs_dev) (void)hipFree(c->jobs_dev);
  if (c->sdesc_dev) (void)hipFree(c->sdesc_dev);
  if (c->qhead) (void)hipFree(c->qhead);
  if (c->node_cnt) (void)hipFree(c->node_cnt);
  if (c->track_state_dev) (void)hipFree(c->track_state_dev);
  if (c->done_flag_host) (void)hipHostFree(c->done_flag_host);
  if (c->done_counter) (void)hipFree(c->done_counter);
  if (c->fin_scratch) (void)hipFree(c->fin_scratch);
  if (c->fin_cnt) (void)hipFree(c->fin_cnt);
  if (c->sg_dev) (void)hipFree(c->sg_dev);
  if (c->pyr_dev) (void)hipFree(c->pyr_dev);
  for (auto& t : c->ray_tabs) (void)hipFree(t.dev);
  if (c->stage_host) (void)hipHostFree(c->stage_host);
  if (c->result_host) (void)hipHostFree(c->result_host);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (int i = 0; i < kStageSlots; ++i) if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
  for (int i = 0; i < kStageSlots; ++i) if (c->stage_rel_ev[i]) (void)hipEventDestroy(c->stage_rel_ev[i]);
  for (int i = 0; i < kStageSlots; ++i) if (c->slot_done[i]) (void)hipEventDestroy(c->slot_done[i]);
  for (int i = 0; i < kStageSlots; ++i) if (c->sdesc_done[i]) (void)hipEventDestroy(c->sdesc_done[i]);
  for (auto& pr : c->prof_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

DFX_API int dfx_ctx_set_stream(dfx_ctx* c, void* stream) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((hipStream_t)stream == c->stream) return DFX_OK;
  DFX_HIP(hipStreamSynchronize(c->stream));   // staging slots, scratch and result area are ordered on the old stream
  DFX_HIP(hipStreamSynchronize(c->copy_stream));
  if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
  c->tail_busy[0] = c->tail_busy[1] = false;
  c->stream = (hipStream_t)stream;
  for (int i = 0; i < kStageSlots; ++i) { c->stage_used[i] = false; c->slot_busy[i] = false; c->sdesc_busy[i] = false; }
  return DFX_OK;
}

DFX_API int dfx_ctx_device(dfx_ctx* c) { return c ? c->device : -1; }

DFX_API int dfx_sync(dfx_ctx* c) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  { int rc; if ((rc = wait_stream(c))) return rc; }
  if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
  return DFX_OK;
}

DFX_API int dfx_set_tail_stream(dfx_ctx* c, void* tail_stream) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((hipStream_t)tail_stream == c->tail_stream) return DFX_OK;
  if (tail_stream && (hipStream_t)tail_stream == c->stream) return fail(DFX_E_INVALID, "the tail stream must differ from the context's stream");
  DFX_HIP(hipStreamSynchronize(c->stream));
  if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
  DFX_HIP(hipStreamSynchronize(c->copy_stream));
  for (int i = 0; i < 2 && tail_stream; ++i) {
    if (!c->ev_mid[i]) DFX_HIP(hipEventCreateWithFlags(&c->ev_mid[i], hipEventDisableTiming));
    if (!c->ev_tail[i]) DFX_HIP(hipEventCreateWithFlags(&c->ev_tail[i], hipEventDisableTiming));
  }
  if (tail_stream && !c->ev_join) DFX_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  c->tail_stream = (hipStream_t)tail_stream;
  c->tail_busy[0] = c->tail_busy[1] = false;
  c->tail_parity = 0;
  for (int i = 0; i < kStageSlots; ++i) { c->stage_used[i] = false; c->slot_busy[i] = false; c->sdesc_busy[i] = false; }
  // the scratch is re-created with (or without) its second half on next use
  if (c->partials_base) DFX_HIP(hipFree(c->partials_base));
  c->partials_base = nullptr; c->partials = c->partials_alt = nullptr; c->partials_bytes = 0;
  if (c->qhead) DFX_HIP(hipFree(c->qhead));
  c->qhead = nullptr; c->qhead_cap = 0; c->qhead_dirty = false;
  return DFX_OK;
}

DFX_API int dfx_tail_join(dfx_ctx* c) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  if (!c->tail_stream) return DFX_OK;
  DFX_HIP(hipEventRecord(c->ev_join, c->tail_stream));
  DFX_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  c->tail_busy[0] = c->tail_busy[1] = false;   // everything enqueued on the context's stream from here on is behind every tail
  return DFX_OK;
}

// The launch shape the library would pick for a batch of `npairs` pairs of one size -- what a rank of a sharded job pins (dfx_sfm_params.step_blocks)
// so that its pairs see the shape of the WHOLE job: per-pair results are then bit-identical however the pair list is split.
DFX_API int dfx_sfm_auto_step_blocks(dfx_ctx* c, int cs, uint32_t w, uint32_t h, int npairs, int distinct_jacobians, int* blocks_out) {
  if (!c || !blocks_out) return fail(DFX_E_INVALID, "dfx_sfm_auto_step_blocks: null argument");
  if (!cs_supported(cs) || w == 0 || h == 0 || npairs < 1) return fail(DFX_E_INVALID, "dfx_sfm_auto_step_blocks: bad argument");
  const int pinned = c->step_blocks, last = c->last_mfma;
  c->step_blocks = 0;            // the context's own pinned value must not answer the question
  *blocks_out = auto_step_blocks(c, w, h, npairs, cs, 0, resolve_mfma(c, cs) == DFX_MFMA_BF16X3, distinct_jacobians != 0);
  c->step_blocks = pinned; c->last_mfma = last;
  return DFX_OK;
}

DFX_API int dfx_sfm_set_step_blocks(dfx_ctx* c, int blocks_per_pair) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  if (blocks_per_pair < 0 || blocks_per_pair > 65535) return fail(DFX_E_INVALID, "blocks_per_pair out of range");
  c->step_blocks = blocks_per_pair;
  return DFX_OK;
}

DFX_API int dfx_device_cu_count(dfx_ctx* c) { return c ? c->cu_count : 0; }

DFX_API int dfx_set_mfma_mode(dfx_ctx* c, int mode) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  if (mode != DFX_MFMA_F32_CHAIN && mode != DFX_MFMA_BF16X3 && mode != DFX_MFMA_AUTO) return fail(DFX_E_INVALID, "unknown MFMA mode %d", mode);
  c->mfma_mode = mode;
  return DFX_OK;
}

DFX_API int dfx_last_mfma_mode(dfx_ctx* c, int* mode) {
  if (!c || !mode) return fail(DFX_E_INVALID, "null argument");
  *mode = c->last_mfma;
  return DFX_OK;
}

DFX_API int dfx_set_schedule(dfx_ctx* c, int mode) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  if (mode != DFX_SCHEDULE_AUTO && mode != DFX_SCHEDULE_STATIC && mode != DFX_SCHEDULE_DYNAMIC) return fail(DFX_E_INVALID, "unknown schedule %d", mode);
  c->schedule = mode;
  return DFX_OK;
}

DFX_API int dfx_set_result_wait(dfx_ctx* c, int mode) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  if (mode != DFX_WAIT_STREAM && mode != DFX_WAIT_POLL) return fail(DFX_E_INVALID, "unknown wait mode %d", mode);
  c->poll = mode == DFX_WAIT_POLL;
  return DFX_OK;
}

DFX_API int dfx_ctx_configure(dfx_ctx* c, int option, int value) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  if (value != 0 && value != 1) return fail(DFX_E_INVALID, "dfx_ctx_configure: option %d takes 0 or 1, got %d", option, value);
  switch (option) {
    case DFX_OPT_SIMPLE_DESC_ZEROCOPY: c->simple_zerocopy = value != 0; return DFX_OK;
    case DFX_OPT_STEP_DESC_ZEROCOPY: c->step_zerocopy = value != 0; return DFX_OK;
    default: return fail(DFX_E_INVALID, "dfx_ctx_configure: unknown option %d", option);
  }
}

DFX_API int dfx_last_schedule(dfx_ctx* c, int* dynamic) {
  if (!c || !dynamic) return fail(DFX_E_INVALID, "null argument");
  *dynamic = c->last_dynamic;
  return DFX_OK;
}

DFX_API int dfx_set_profiling(dfx_ctx* c, int enable) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  DFX_HIP(hipStreamSynchronize(c->stream));
  c->profiling = enable != 0;
  c->prof_used = 0;
  return DFX_OK;
}

DFX_API int dfx_debug_read_partials(dfx_ctx* c, void* host, size_t bytes) {
  if (!c || !host) return fail(DFX_E_INVALID, "null argument");
  if (bytes > c->partials_bytes) return fail(DFX_E_INVALID, "only %zu bytes of partials exist", c->partials_bytes);
  DFX_HIP(hipStreamSynchronize(c->stream));
  DFX_HIP(hipMemcpy(host, c->partials, bytes, hipMemcpyDeviceToHost));
  return DFX_OK;
}

DFX_API int dfx_profile_read_ex(dfx_ctx* c, int* n_launches, double* total_ms, double* min_ms, double* max_ms) {
  if (!c || !n_launches || !total_ms) return fail(DFX_E_INVALID, "null argument");
  DFX_HIP(hipStreamSynchronize(c->stream));
  double tot = 0, lo = 0, hi = 0;
  for (size_t i = 0; i < c->prof_used; ++i) {
    float ms = 0;
    DFX_HIP(hipEventElapsedTime(&ms, c->prof_pool[i].first, c->prof_pool[i].second));
    tot += ms;
    if (i == 0 || ms < lo) lo = ms;
    if (i == 0 || ms > hi) hi = ms;
  }
  *n_launches = (int)c->prof_used;
  *total_ms = tot;
  if (min_ms) *min_ms = lo;
  if (max_ms) *max_ms = hi;
  c->prof_used = 0;
  return DFX_OK;
}

DFX_API int dfx_profile_read(dfx_ctx* c, int* n_launches, double* total_ms) { return dfx_profile_read_ex(c, n_launches, total_ms, nullptr, nullptr); }

// ---- library-owned device images -------------------------------------------------------------------------------
DFX_API int dfx_img_alloc(dfx_ctx* c, uint32_t w, uint32_t h, size_t elem_bytes, dfx_img* out) {
  if (!c || !out) return fail(DFX_E_INVALID, "dfx_img_alloc: null argument");
  if (w == 0 || h == 0 || (elem_bytes != 4 && elem_bytes != 8)) return fail(DFX_E_INVALID, "dfx_img_alloc: %ux%u elements of %zu bytes", w, h, elem_bytes);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const size_t pitch = ((size_t)w * elem_bytes + 15) & ~(size_t)15;
  if (pitch * (size_t)h >= 0x70000000ull) return fail(DFX_E_INVALID, "image of %zu bytes exceeds the 0x70000000-byte buffer-resource limit", pitch * (size_t)h);
  void* p = nullptr;
  DFX_HIP(hipMalloc(&p, pitch * (size_t)h));
  hipError_t e = hipMemsetAsync(p, 0, pitch * (size_t)h, c->stream);
  if (e != hipSuccess) { (void)hipFree(p); return fail(DFX_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e)); }
  {
    std::lock_guard<std::mutex> lk(g_img_mu);
    g_imgs[p] = ImgRec{ c->device, pitch, w, h, elem_bytes, nullptr, true, 0.0f };
    g_img_count.fetch_add(1, std::memory_order_relaxed);
  }
  *out = dfx_img{ p, pitch, w, h };
  return DFX_OK;
}

DFX_API int dfx_host_alloc(dfx_ctx* c, size_t bytes, void** out) {
  if (!c || !out || bytes == 0) return fail(DFX_E_INVALID, "dfx_host_alloc: null argument / zero size");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  void* p = nullptr;
  DFX_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  *out = p;
  return DFX_OK;
}
DFX_API int dfx_host_free(dfx_ctx* c, void* ptr) {
  if (!c) return fail(DFX_E_INVALID, "dfx_host_free: null context");
  if (!ptr) return DFX_OK;
  int rc;
  if ((rc = ensure_device(c))) return rc;
  DFX_HIP(hipStreamSynchronize(c->stream));   // a copy into it may still be in flight
  DFX_HIP(hipHostFree(ptr));
  return DFX_OK;
}

DFX_API int dfx_img_free(dfx_ctx* c, dfx_img* img) {
  if (!c || !img) return fail(DFX_E_INVALID, "dfx_img_free: null argument");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (img->ptr) {
    DFX_HIP(hipStreamSynchronize(c->stream));
    if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));   // a deferred tail may still read the map and rewrite its shadow
    unsigned long long* shadow = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_img_mu);
      auto it = g_imgs.find(img->ptr);
      if (it != g_imgs.end()) { shadow = it->second.shadow; g_imgs.erase(it); g_img_count.fetch_sub(1, std::memory_order_relaxed); }
    }
    if (shadow) { (void)hipFree(shadow); g_shadow_count.fetch_sub(1, std::memory_order_relaxed); }
    DFX_HIP(hipFree(img->ptr));
  }
  *img = dfx_img{ nullptr, 0, 0, 0 };
  return DFX_OK;
}

DFX_API int dfx_img_upload(dfx_ctx* c, const dfx_img* dst, const void* host, size_t host_pitch, size_t elem_bytes) {
  if (!c || !img_ok(dst) || !host) return fail(DFX_E_INVALID, "dfx_img_upload: null argument");
  const size_t row = (size_t)dst->w * elem_bytes;
  if (host_pitch < row || dst->pitch_bytes < row) return fail(DFX_E_INVALID, "dfx_img_upload: pitch smaller than a row of %zu bytes", row);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = img_note_write(c, dst))) return rc;
  DFX_HIP(hipMemcpy2DAsync(dst->ptr, dst->pitch_bytes, host, host_pitch, row, dst->h, hipMemcpyHostToDevice, c->stream));
  DFX_HIP(hipStreamSynchronize(c->stream));
  return DFX_OK;
}

DFX_API int dfx_img_download(dfx_ctx* c, const dfx_img* src, void* host, size_t host_pitch, size_t elem_bytes) {
  if (!c || !img_ok(src) || !host) return fail(DFX_E_INVALID, "dfx_img_download: null argument");
  const size_t row = (size_t)src->w * elem_bytes;
  if (host_pitch < row || src->pitch_bytes < row) return fail(DFX_E_INVALID, "dfx_img_download: pitch smaller than a row of %zu bytes", row);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  DFX_HIP(hipMemcpy2DAsync(host, host_pitch, src->ptr, src->pitch_bytes, row, src->h, hipMemcpyDeviceToHost, c->stream));
  DFX_HIP(hipStreamSynchronize(c->stream));
  return DFX_OK;
}

DFX_API int dfx_img_fill_f32(dfx_ctx* c, const dfx_img* dst, float value) {
  if (!c || !img_ok(dst)) return fail(DFX_E_INVALID, "dfx_img_fill_f32: null argument");
  if (dst->pitch_bytes & 3) return fail(DFX_E_INVALID, "dfx_img_fill_f32: pitch not a multiple of 4");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  int bits;
  std::memcpy(&bits, &value, 4);
  if ((rc = img_note_write(c, dst, true, value))) return rc;
  DFX_HIP(hipMemsetD32Async((hipDeviceptr_t)dst->ptr, bits, dst->pitch_bytes / 4 * (size_t)dst->h, c->stream));   // row padding included
  return DFX_OK;
}

DFX_API int dfx_debug_read_valid0_shadow(dfx_ctx* c, const dfx_img* img, uint64_t* host_words, size_t cap_words, size_t* n_words) {
  if (!c || !img || !n_words) return fail(DFX_E_INVALID, "dfx_debug_read_valid0_shadow: null argument");
  *n_words = 0;
  unsigned long long* sh = nullptr;
  size_t nw = 0;
  {
    std::lock_guard<std::mutex> lk(g_img_mu);
    auto it = g_imgs.find(img->ptr);
    if (it != g_imgs.end() && it->second.shadow) { sh = it->second.shadow; nw = shadow_words(it->second.w, it->second.h); }
  }
  if (!sh) return DFX_OK;
  if (!host_words || cap_words < nw) return fail(DFX_E_INVALID, "shadow has %zu words, buffer holds %zu", nw, cap_words);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  DFX_HIP(hipStreamSynchronize(c->stream));
  DFX_HIP(hipMemcpy(host_words, sh + 1, nw * 8, hipMemcpyDeviceToHost));
  *n_words = nw;
  return DFX_OK;
}

// ---- SfmAligner ------------------------------------------------------------------------------------------------
struct dfx_graph;
// before_launch (optional): enqueued work the step kernel depends on (the decoder launches of dfx_sfm_linearize_batch), issued once the step's own host work --
// validation, descriptors, launch shape -- is done, right in front of the step kernel: the GPU then runs the two back to back instead of idling through the
// step's preparation (5.6 us between a single pair's decode and its step in the kernel trace).
static int sfm_step_batch_impl(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n, void* out_items_dev, bool allow_defer,
                               const dfx_graph* graph = nullptr, int first_pair = 0, float* sys_dev = nullptr, const std::function<int()>* before_launch = nullptr,
                               const std::function<int()>* after_launch = nullptr);   // after_launch: behind the step's launches (runs on every exit once before_launch has)

DFX_API int dfx_sfm_step_batch_async(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n,
                                     void* out_items_dev) {
  return sfm_step_batch_impl(c, cs, params, pairs, n, out_items_dev, true);
}

// allow_defer = false (the blocking entry points): the finalize kernel runs on the context's stream even in deferred-tail mode
static int graph_tail(dfx_ctx* c, int cs, const dfx_graph* g, int first_pair, int n, float* sys_dev, dfx::TailGraphDev* tg, int* node_wgs);

static int sfm_step_batch_impl(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n, void* out_items_dev, bool allow_defer,
                               const dfx_graph* graph, int first_pair, float* sys_dev, const std::function<int()>* before_launch, const std::function<int()>* after_launch) {
  if (!c || !params || !pairs || !out_items_dev) return fail(DFX_E_INVALID, "dfx_sfm_step_batch: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "batch size %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (params->step_blocks < 0 || params->step_blocks > 65535) return fail(DFX_E_INVALID, "step_blocks %d out of range [0,65535]", params->step_blocks);
  // graph assembly: inside the reduction tail where the launch has the one-workgroup-per-pair tail kernel (batched, bf16 split), as a
  // second kernel behind the finalize kernel otherwise (single pair, fp32 chain) -- the same sums in the same order either way.
  // Checked first: an inconsistent graph fails the call before anything is enqueued.
  dfx::TailGraphDev tg{};
  int node_wgs = 0;
  if (graph && (rc = graph_tail(c, cs, graph, first_pair, n, sys_dev, &tg, &node_wgs))) return rc;
  // Image sizes: a batch may mix pyramid levels (the reference walks all levels of a factor set per relinearisation,
  // core/mapping/df_work.cpp:118-136, tools/kernel_benchmark.cpp:192-203).  W, H = the largest width / height (ray-table LDS).
  uint32_t W = 0, H = 0;
  bool uniform = true;
  for (int p = 0; p < n; ++p) {
    if (!img_ok(&pairs[p].img0)) return fail(DFX_E_INVALID, "pair %d: img0 null or empty", p);
    const uint32_t w = pairs[p].img0.w, h = pairs[p].img0.h;
    if ((size_t)w * h >= (1ull << 31)) return fail(DFX_E_INVALID, "pair %d: image too large", p);
    uniform = uniform && w == pairs[0].img0.w && h == pairs[0].img0.h;
    W = w > W ? w : W; H = h > H ? h : H;
  }
  // does every pair stream its own Jacobian image?  (decides how long the waves may be: chunks_per_wave_max)
  bool distinct_jac = true;
  {
    std::vector<const void*> jp((size_t)n);
    for (int p = 0; p < n; ++p) jp[(size_t)p] = pairs[p].prx0_jac.ptr;
    std::sort(jp.begin(), jp.end());
    distinct_jac = std::adjacent_find(jp.begin(), jp.end()) == jp.end();
  }
  // Launch shape of a mixed batch: one chunks-per-wave figure for all pairs (so that a 160x120 pair gets 1/16 of the workgroups of a
  // 640x480 one instead of the same number of much shorter ones), a 1-D grid, the large pairs first: the small levels fill the tail.
  std::vector<uint32_t> nblk, blk0;
  std::vector<int> order;
  int total_blocks = 0;
  if (!uniform) {
    long long total_chunks = 0;
    for (int p = 0; p < n; ++p) total_chunks += ((long long)pairs[p].img0.w * pairs[p].img0.h + 63) / 64;
    const long long resident_waves = (cs >= 64 ? 8LL : 12LL) * c->cu_count;
    int cpw = (int)(4 * total_chunks / (5 * resident_waves));
    if (cpw < 5) cpw = 5;
    const int cpw_cap = chunks_per_wave_max(cs, resolve_mfma(c, cs) == DFX_MFMA_BF16X3, distinct_jac);
    if (cpw > cpw_cap) cpw = cpw_cap;
    if (params->step_blocks > 0 || c->step_blocks > 0) {   // an explicit request is read as "workgroups of a pair of the LARGEST size"
      const int req = params->step_blocks > 0 ? params->step_blocks : c->step_blocks;
      const long long big = ((long long)W * H + 63) / 64;
      cpw = (int)((big + 4LL * req - 1) / (4LL * req));
      if (cpw < 1) cpw = 1;
    }
    nblk.resize((size_t)n); blk0.resize((size_t)n); order.resize((size_t)n);
    for (int p = 0; p < n; ++p) {
      const long long nch = ((long long)pairs[p].img0.w * pairs[p].img0.h + 63) / 64;
      long long b = (nch + 4LL * cpw - 1) / (4LL * cpw);
      if (b < 1) b = 1;
      if (b > 65535) b = 65535;
      nblk[p] = (uint32_t)b;
      order[p] = p;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (size_t)pairs[a].img0.w * pairs[a].img0.h > (size_t)pairs[b].img0.w * pairs[b].img0.h; });
    // the workgroup map packs (pair << 16 | block): 65535 pairs x 65535 workgroups each, and the grid must stay a positive int
    if (n > 65535) return fail(DFX_E_INVALID, "a batch of several image sizes holds at most 65535 pairs (got %d)", n);
    long long tb = 0;
    for (int p : order) { blk0[p] = (uint32_t)tb; tb += (long long)nblk[p]; }
    if (tb > 0x7fffffffLL) return fail(DFX_E_INVALID, "launch shape of %lld workgroups exceeds the grid limit", tb);
    total_blocks = (int)tb;
  }

  if ((rc = ray_table_gc(c))) return rc;
  // n == 1 (the reference's call pattern: one pair per blocking call): the descriptor travels in the kernel arguments -- no
  // staging slot, no host-to-device copy in front of the launch.  n > 1: descriptor array (+ the workgroup map of a mixed batch) in
  // device memory.
  dfx::SfmPairDev one;
  dfx::SfmPairDev* hd = &one;
  dfx::SfmPairDev* dd = nullptr;
  unsigned* map_dev = nullptr;
  int slot = -1;
  const size_t desc_bytes = (sizeof(dfx::SfmPairDev) * (size_t)n + 15) & ~(size_t)15;
  const size_t map_bytes = sizeof(unsigned) * (size_t)total_blocks;
  if (n > 1) {
    char* host;
    if ((rc = stage_acquire(c, desc_bytes + map_bytes, &slot, &host))) return rc;
    hd = reinterpret_cast<dfx::SfmPairDev*>(host);
    unsigned* hm = reinterpret_cast<unsigned*>(host + desc_bytes);
    size_t g = 0;
    for (int p : order)
      for (uint32_t b = 0; b < nblk[p]; ++b) hm[g++] = ((unsigned)p << 16) | b;
  }
  for (int p = 0; p < n; ++p) {
    const dfx_sfm_pair& q = pairs[p];
    if ((rc = fill_sfm_pair(c, cs, &q.pose0, &q.pose1, &q.cam, &q.img0, &q.img1, &q.dpt0, &q.valid0, &q.prx0_jac, &q.grad1, q.img0.w, q.img0.h, &hd[p]))) {
      const std::string why = "pair " + std::to_string(p) + ": " + g_last_error;
      if (slot >= 0) (void)stage_release(c, slot);   // the slot was handed out: give it its event like every other path
      g_last_error = why;
      return rc;
    }
    hd[p].w_px = q.img0.w; hd[p].h_px = q.img0.h;
    hd[p].nblk = uniform ? 0u : nblk[p];
    hd[p].blk0 = uniform ? 0u : blk0[p];
  }
  if (n > 1) {
    // device copy: one region per stage slot so that in-flight launches keep their own.  The upload runs on the context's copy stream,
    // beside the kernels of the previous launches (it used to sit between them on the launch stream: ~10 us per step); the launch
    // stream waits for it, and the copy stream waits for the last kernels that read this slot.
    if (c->pairs_cap < desc_bytes + map_bytes) {
      DFX_HIP(hipStreamSynchronize(c->stream));
      DFX_HIP(hipStreamSynchronize(c->copy_stream));
      if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
      if (c->pairs_dev) DFX_HIP(hipFree(c->pairs_dev));
      c->pairs_dev = nullptr;
      const size_t cap = ((desc_bytes + map_bytes) * 2 + 255) & ~(size_t)255;
      DFX_HIP(hipMalloc((void**)&c->pairs_dev, cap * kStageSlots));
      c->pairs_cap = cap;
      for (int i = 0; i < kStageSlots; ++i) c->slot_busy[i] = false;
    }
    char* region = reinterpret_cast<char*>(c->pairs_dev) + (size_t)slot * c->pairs_cap;
    if (desc_zerocopy(c)) {
      void* hdev = nullptr;
      DFX_HIP(hipHostGetDevicePointer(&hdev, hd, 0));
      region = reinterpret_cast<char*>(hdev);   // the staging slot itself; stage_ev[slot] is recorded behind the kernels below
    }
    dd = reinterpret_cast<dfx::SfmPairDev*>(region);
    if (!uniform) map_dev = reinterpret_cast<unsigned*>(region + desc_bytes);
    if (!desc_zerocopy(c)) {
      if (c->slot_busy[slot]) DFX_HIP(hipStreamWaitEvent(c->copy_stream, c->slot_done[slot], 0));
      DFX_HIP(hipMemcpyAsync(region, hd, desc_bytes + map_bytes, hipMemcpyHostToDevice, c->copy_stream));
      DFX_HIP(hipEventRecord(c->stage_ev[slot], c->copy_stream));
      c->stage_used[slot] = true;
      c->stage_rel[slot] = false;
      DFX_HIP(hipStreamWaitEvent(c->stream, c->stage_ev[slot], 0));
    }
  }

  // the dense-stream variant needs every pair's Jacobian rows back to back (and, in a mixed batch, whole 64-pixel chunks: the launcher
  // checks that for a batch of one size); one pitched pair selects the general kernel
  bool jac_dense = true;
  bool vsh = true;   // the shadow-reading kernel variant needs a shadow behind EVERY valid0 map of the batch (library-owned images)
  for (int p = 0; p < n; ++p) {
    jac_dense = jac_dense && (hd[p].pitch_jac == hd[p].w_px * (uint32_t)cs * 4u) && (uniform || ((size_t)hd[p].w_px * hd[p].h_px) % 64 == 0);
    vsh = vsh && (hd[p].valid0 == nullptr || hd[p].valid0_shadow != nullptr);
  }
  // Dynamic schedule (k_sfm_step<..., DYN>, opt-in: DFX_SCHEDULE_DYNAMIC): resident wave-workers popping items from per-pair queues.
  // It needs 64-pixel columns (W % 64 == 0), dense Jacobian rows and the per-wave ray tables beside the P rows in LDS.  Measured
  // against the static launch on the same box, 128 pairs: -1.2 ... -1.5 % on four boxes, +-0 on one, +4.5 % on two -- its gain
  // depends on the box, so it is not the default.  The queue heads are device-scope atomics (served memory-side, across the 8 XCDs'
  // L2s: ~0.2 us each, serialised per word): teams of more than 32 waves (fewer than 128 pairs) spend their time queueing for pops
  // (16 pairs: 515 vs 151 us; 64 pairs: 647 vs 558 us).
  dfx::DynDev dyn{ nullptr, 0, 0, 0, 0, 0u };
  int dyn_grid = 0;
  {
    const int resident_wgs = 4 * c->cu_count;
    const int team = n > 0 ? (4 * resident_wgs) / n : 0;
    const size_t dyn_lds = sizeof(float) * 4 * ((size_t)W + H + dfx::kRayTabSlack + 16 * 68);
    const bool team_ok = team >= 1 && team <= 1024;
    if (n > 1 && uniform && c->schedule == DFX_SCHEDULE_DYNAMIC && params->step_blocks == 0 && c->step_blocks == 0 && team_ok && jac_dense && W % 64 == 0 &&
        W / 64 <= 64 && dyn_lds <= 40 * 1024 && (size_t)W * H < (1u << 26)) {
      const int vs = (int)(W / 64);
      int R = (int)(((long long)H * vs) / ((long long)team * 24));
      if (R < 2) R = 2;
      if (R > 32) R = 32;
      dyn.items_per_pair = vs * (int)((H + R - 1) / R);
      dyn.rows_per_item = R;
      dyn.npairs = n;
      dyn.team = team;
      dyn.vs_magic = (unsigned)((1ull << 32) / (unsigned)vs + 1ull);
      dyn_grid = resident_wgs;
      if (c->qhead_cap < (size_t)n) {
        DFX_HIP(hipStreamSynchronize(c->stream));
        if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
        if (c->qhead) DFX_HIP(hipFree(c->qhead));
        c->qhead = nullptr;
        DFX_HIP(hipMalloc((void**)&c->qhead, sizeof(unsigned) * (size_t)n * 4));   // two halves (deferred-tail mode alternates them)
        DFX_HIP(hipMemsetAsync(c->qhead, 0, sizeof(unsigned) * (size_t)n * 4, c->stream));
        c->qhead_cap = (size_t)n * 2;
      }
      dyn.qhead = c->qhead;   // + the half's offset below
    }
  }
  c->last_dynamic = dyn.qhead ? 1 : 0;
  int bpp = dyn.qhead ? dyn.team : (uniform ? auto_step_blocks(c, W, H, n, cs, params->step_blocks, resolve_mfma(c, cs) == DFX_MFMA_BF16X3, distinct_jac) : 0);
  if (!uniform) for (int p = 0; p < n; ++p) bpp = std::max(bpp, (int)nblk[(size_t)p]);   // mixed sizes: the partials of the largest pair (the launcher picks the tail kernel by it)
  const size_t pbytes = uniform ? dfx::sfm_step_partials_bytes(cs, n, bpp) : dfx::sfm_step_partials_bytes(cs, 1, total_blocks);
  if ((rc = grow_partials(c, pbytes, true))) return rc;
  // Deferred tail: this launch's finalize kernel goes to the tail stream and runs beside the NEXT launch's step kernel; the two halves of
  // the partials (and of the queue heads) alternate, and a half is written again only after the finalize that read it (ev_tail).
  const bool defer = allow_defer && c->tail_stream != nullptr;
  const int par = c->tail_stream ? c->tail_parity : 0;
  float* const partials = par ? c->partials_alt : c->partials;
  if (c->tail_busy[par]) { DFX_HIP(hipStreamWaitEvent(c->stream, c->ev_tail[par], 0)); c->tail_busy[par] = false; }
  if (dyn.qhead) dyn.qhead = c->qhead + (size_t)par * c->qhead_cap;

  dfx::SfmParamsDev prm{ params->huber_delta, params->avg_dpt, params->min_dpt, (float)params->valid_border, next_launch_id() };
  if (before_launch && (rc = (*before_launch)())) {
    const std::string why = g_last_error;
    if (after_launch) (void)(*after_launch)();
    if (slot >= 0) (void)stage_release(c, slot);
    g_last_error = why;
    return rc;
  }
  struct AfterGuard {   // the hook runs on every exit below
    const std::function<int()>* f;
    ~AfterGuard() { if (f) (void)(*f)(); }
  } after_guard{ before_launch ? after_launch : nullptr };
  hipEvent_t eb = nullptr, ee = nullptr;
  if ((rc = prof_events(c, &eb, &ee))) return rc;
  if (dyn.qhead) {
    if (c->qhead_dirty) {   // a failed launch left heads behind: nothing may be in flight on them when they are cleared
      if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
      DFX_HIP(hipMemsetAsync(c->qhead, 0, sizeof(unsigned) * c->qhead_cap * 2, c->stream));
    }
    c->qhead_dirty = true;
  }
  hipStream_t const fin_stream = defer ? c->tail_stream : c->stream;
  bool assembled = false;
  dfx::DoneFlag* dn = c->done_armed;   // (blocking single-pair entries; launch_sfm_step clears the flag when the launch has no signalling tail)
  if (dn && (graph || defer)) { dn->flag = nullptr; dn = nullptr; }
  const bool split_fin = n == 1 && uniform && !defer;   // a single pair on the context's stream: the finalize kernel with four workgroups per tile
  if (split_fin && (rc = ensure_fin_scratch(c))) return rc;
  DFX_HIP(dfx::launch_sfm_step(cs, dd, n, (int)W, (int)H, prm, bpp, partials, out_items_dev, dfx_item_size(12 + cs), c->stream,
                               jac_dense, resolve_mfma(c, cs), eb, ee, n == 1 ? &one : nullptr, dyn.qhead ? &dyn : nullptr, dyn_grid, vsh,
                               fin_stream, defer ? c->ev_mid[par] : nullptr, map_dev, total_blocks, graph ? &tg : nullptr, node_wgs, &assembled,
                               dn, split_fin ? c->fin_scratch : nullptr, split_fin ? c->fin_cnt : nullptr));
  c->qhead_dirty = false;   // both kernels are enqueued: the finalize kernel rewinds the heads
  if (graph) {
    c->node_cnt_dirty = false;   // the tail kernel rewinds the counters it used
    if (!assembled) DFX_HIP(dfx::launch_graph_assemble(cs, tg.G, out_items_dev, dfx_item_size(12 + cs), first_pair, n, sys_dev, fin_stream));
  }
  if (defer) {
    DFX_HIP(hipEventRecord(c->ev_tail[par], fin_stream));
    c->tail_busy[par] = true;
  }
  if (c->tail_stream) c->tail_parity ^= 1;
  if (n > 1) {   // the finalize kernel reads the descriptors too
    if (desc_zerocopy(c)) {   // the kernels read the staging slot itself: it is free again behind them
      DFX_HIP(hipEventRecord(c->stage_ev[slot], fin_stream));
      c->stage_used[slot] = true;
      c->stage_rel[slot] = false;
    } else {
      DFX_HIP(hipEventRecord(c->slot_done[slot], fin_stream));
      c->slot_busy[slot] = true;
    }
  }
  return DFX_OK;
}

// ---- keyframe graph: block-sparse Gauss-Newton normal equations ------------------------------------------------
struct dfx_graph {
  int device = 0;
  int cs = 0, n_nodes = 0, n_pairs = 0;
  int* dev = nullptr;   // [kf_begin (n_nodes + 1)][fr_begin (n_nodes + 1)][kf_pairs (n_pairs)][fr_pairs (n_pairs)][pair_nodes (2 n_pairs)]
  dfx::GraphDev view{};
  const int* pair_nodes_dev = nullptr;
  bool has_isolated = false;   // a node without any pair: its (zero) blocks need a writer even when every pair is local
};

DFX_API int dfx_graph_create(dfx_ctx* c, int cs, int n_nodes, int n_pairs, const int32_t* pair_nodes, dfx_graph** out) {
  if (!c || !pair_nodes || !out) return fail(DFX_E_INVALID, "dfx_graph_create: null argument");
  *out = nullptr;
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  if (n_nodes < 2 || n_pairs < 1 || n_nodes > (1 << 20) || n_pairs > (1 << 24)) return fail(DFX_E_INVALID, "graph of %d nodes / %d pairs out of range", n_nodes, n_pairs);
  for (int p = 0; p < n_pairs; ++p) {
    const int a = pair_nodes[2 * p], b = pair_nodes[2 * p + 1];
    if (a < 0 || a >= n_nodes || b < 0 || b >= n_nodes || a == b)
      return fail(DFX_E_INVALID, "pair %d links nodes (%d, %d): need two distinct nodes in [0, %d)", p, a, b, n_nodes);
  }
  int rc;
  if ((rc = ensure_device(c))) return rc;
  // CSR by counting sort: pair indices stay ascending inside every node's list (the fixed summation order)
  const size_t nb = (size_t)n_nodes + 1;
  std::vector<int> h(2 * nb + 4 * (size_t)n_pairs, 0);
  int* kf_begin = h.data();
  int* fr_begin = h.data() + nb;
  int* kf_pairs = h.data() + 2 * nb;
  int* fr_pairs = kf_pairs + n_pairs;
  for (int p = 0; p < n_pairs; ++p) { kf_begin[pair_nodes[2 * p] + 1]++; fr_begin[pair_nodes[2 * p + 1] + 1]++; }
  for (int n = 0; n < n_nodes; ++n) { kf_begin[n + 1] += kf_begin[n]; fr_begin[n + 1] += fr_begin[n]; }
  std::vector<int> kc(kf_begin, kf_begin + n_nodes), fc(fr_begin, fr_begin + n_nodes);
  for (int p = 0; p < n_pairs; ++p) { kf_pairs[kc[pair_nodes[2 * p]]++] = p; fr_pairs[fc[pair_nodes[2 * p + 1]]++] = p; }
  std::memcpy(fr_pairs + n_pairs, pair_nodes, sizeof(int) * 2 * (size_t)n_pairs);
  bool isolated = false;
  for (int n = 0; n < n_nodes; ++n) isolated = isolated || (kf_begin[n + 1] == kf_begin[n] && fr_begin[n + 1] == fr_begin[n]);
  dfx_graph* g = new dfx_graph();
  g->device = c->device; g->cs = cs; g->n_nodes = n_nodes; g->n_pairs = n_pairs;
  hipError_t e = hipMalloc((void**)&g->dev, h.size() * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(g->dev, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (g->dev) (void)hipFree(g->dev); delete g; return fail(DFX_E_HIP, "dfx_graph_create: %s", hipGetErrorString(e)); }
  g->view = dfx::GraphDev{ n_nodes, n_pairs, g->dev, g->dev + 2 * nb, g->dev + nb, g->dev + 2 * nb + n_pairs };
  g->pair_nodes_dev = g->dev + 2 * nb + 2 * (size_t)n_pairs;
  g->has_isolated = isolated;
  *out = g;
  return DFX_OK;
}

DFX_API void dfx_graph_destroy(dfx_graph* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->dev) (void)hipFree(g->dev);
  delete g;
}

DFX_API size_t dfx_graph_system_floats(const dfx_graph* g) {
  if (!g) return 0;
  const size_t D = 6 + (size_t)g->cs;
  return (size_t)g->n_nodes * D * D + (size_t)g->n_pairs * D * 6 + (size_t)g->n_nodes * D;
}

DFX_API int dfx_graph_assemble_async(dfx_ctx* c, const dfx_graph* g, const void* items_dev, int first_pair, int n_local, float* sys_dev) {
  if (!c || !g || !items_dev || !sys_dev) return fail(DFX_E_INVALID, "dfx_graph_assemble: null argument");
  if (g->device != c->device) return fail(DFX_E_INVALID, "graph lives on device %d, context on device %d", g->device, c->device);
  if (first_pair < 0 || n_local < 0 || first_pair + n_local > g->n_pairs)
    return fail(DFX_E_INVALID, "pairs [%d, %d) are not inside the graph's %d pairs", first_pair, first_pair + n_local, g->n_pairs);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  // deferred-tail mode: the items are produced on the tail stream, so the assembly runs there as well
  DFX_HIP(dfx::launch_graph_assemble(g->cs, g->view, items_dev, dfx_item_size(12 + g->cs), first_pair, n_local, sys_dev,
                                     c->tail_stream ? c->tail_stream : c->stream));
  return DFX_OK;
}

// Arguments of the graph assembly inside the reduction tail: the graph's device view, this rank's pair range, the arrival counters.
// Node workgroups (zero fill of what no local pair writes) are only launched when this rank does not hold every pair of the graph.
static int graph_tail(dfx_ctx* c, int cs, const dfx_graph* g, int first_pair, int n, float* sys_dev, dfx::TailGraphDev* tg, int* node_wgs) {
  if (!sys_dev) return fail(DFX_E_INVALID, "dfx_sfm_step_batch_assemble: null system buffer");
  if (g->device != c->device) return fail(DFX_E_INVALID, "graph lives on device %d, context on device %d", g->device, c->device);
  if (g->cs != cs) return fail(DFX_E_INVALID, "graph of code size %d, step of code size %d", g->cs, cs);
  if (first_pair < 0 || first_pair + n > g->n_pairs)
    return fail(DFX_E_INVALID, "pairs [%d, %d) are not inside the graph's %d pairs", first_pair, first_pair + n, g->n_pairs);
  if (c->node_cnt_cap < (size_t)g->n_nodes || c->node_cnt_dirty) {
    DFX_HIP(hipStreamSynchronize(c->stream));
    if (c->tail_stream) DFX_HIP(hipStreamSynchronize(c->tail_stream));
    if (c->node_cnt_cap < (size_t)g->n_nodes) {
      if (c->node_cnt) DFX_HIP(hipFree(c->node_cnt));
      c->node_cnt = nullptr; c->node_cnt_cap = 0;
      DFX_HIP(hipMalloc((void**)&c->node_cnt, sizeof(unsigned) * (size_t)g->n_nodes * 2));
      c->node_cnt_cap = (size_t)g->n_nodes * 2;
    }
    DFX_HIP(hipMemsetAsync(c->node_cnt, 0, sizeof(unsigned) * c->node_cnt_cap, c->stream));
  }
  c->node_cnt_dirty = true;   // cleared once the kernel that rewinds them is enqueued
  *tg = dfx::TailGraphDev{ g->view, g->pair_nodes_dev, first_pair, n, sys_dev, c->node_cnt };
  *node_wgs = (first_pair == 0 && n == g->n_pairs && !g->has_isolated) ? 0 : g->n_nodes;
  return DFX_OK;
}

DFX_API int dfx_sfm_step_batch_assemble_async(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n,
                                              void* out_items_dev, const dfx_graph* graph, int first_pair, float* sys_dev) {
  if (!graph) return fail(DFX_E_INVALID, "dfx_sfm_step_batch_assemble: null graph");
  return sfm_step_batch_impl(c, cs, params, pairs, n, out_items_dev, true, graph, first_pair, sys_dev);
}

DFX_API int dfx_sfm_step_batch(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n,
                               void* out_items_host) {
  if (!c || !out_items_host) return fail(DFX_E_INVALID, "dfx_sfm_step_batch: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  if (n <= 0) return fail(DFX_E_INVALID, "batch size %d", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const size_t bytes = dfx_item_size(12 + cs) * (size_t)n;
  if (bytes <= kDirectResultMax) {
    void* tgt;
    if ((rc = result_target(c, bytes, &tgt))) return rc;
    dfx::DoneFlag done;
    if (n == 1 && (rc = new_done_flag(c, &done))) return rc;   // a single pair (the reference's per-factor call): the finalize kernel signals, the host polls
    c->done_armed = done.flag ? &done : nullptr;
    rc = sfm_step_batch_impl(c, cs, params, pairs, n, tgt, false);
    c->done_armed = nullptr;
    if (rc) return rc;
    return finish_result_polled(c, out_items_host, bytes, done);
  }
  if (c->items_bytes < bytes) DFX_HIP(hipStreamSynchronize(c->stream));
  if ((rc = grow_dev((void**)&c->items_dev, &c->items_bytes, bytes, c->stream))) return rc;
  if ((rc = sfm_step_batch_impl(c, cs, params, pairs, n, c->items_dev, false))) return rc;
  return fetch_result(c, c->items_dev, out_items_host, bytes);
}

DFX_API int dfx_sfm_step(dfx_ctx* c, int cs, const dfx_se3* pose0, const dfx_se3* pose1, const dfx_cam* cam,
                         const dfx_sfm_params* params, const dfx_img* img0, const dfx_img* img1, const dfx_img* dpt0,
                         const dfx_img* std0, const dfx_img* valid0, const dfx_img* prx0_jac, const dfx_img* grad1,
                         void* out_item) {
  (void)std0;   // dead input in the reference: the uncertainty weight is computed and discarded (dense_sfm.h:58-67)
  if (!pose0 || !pose1 || !cam || !img0 || !img1 || !dpt0 || !prx0_jac || !grad1) return fail(DFX_E_INVALID, "dfx_sfm_step: null argument");
  dfx_sfm_pair p;
  std::memset(&p, 0, sizeof(p));
  p.pose0 = *pose0; p.pose1 = *pose1; p.cam = *cam;
  p.img0 = *img0; p.img1 = *img1; p.dpt0 = *dpt0; p.prx0_jac = *prx0_jac; p.grad1 = *grad1;
  if (valid0) p.valid0 = *valid0;
  return dfx_sfm_step_batch(c, cs, params, &p, 1, out_item);
}

DFX_API int dfx_sfm_error(dfx_ctx* c, const dfx_se3* pose0, const dfx_se3* pose1, const dfx_cam* cam,
                          const dfx_sfm_params* params, const dfx_img* img0, const dfx_img* img1, const dfx_img* dpt0,
                          const dfx_img* std0, const dfx_img* grad1, dfx_corr_item* out) {
  (void)std0; (void)grad1;   // both only feed the discarded uncertainty weight (dense_sfm.h:97-104)
  if (!c || !pose0 || !pose1 || !params || !out) return fail(DFX_E_INVALID, "dfx_sfm_error: null argument");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = ray_table_gc(c))) return rc;
  dfx_se3 p10;
  float R10[9], HM[9];
  relative_pose(*pose0, *pose1, R10, p10.t, nullptr, HM);
  dfx::SimplePairDev d;
  p10.q[0] = p10.q[1] = p10.q[2] = 0; p10.q[3] = 1;
  if ((rc = fill_simple(c, &p10, cam, img0, img1, dpt0, nullptr, nullptr, &d, R10))) return rc;
  const int blocks = simple_blocks(img0->w, img0->h);
  const size_t pbytes = (size_t)blocks * dfx::kSimpleRow * sizeof(float);
  if ((rc = grow_partials(c, pbytes))) return rc;
  void* tgt;
  if ((rc = result_target(c, sizeof(dfx_corr_item), &tgt))) return rc;
  dfx::DoneFlag done;
  if ((rc = new_done_flag(c, &done))) return rc;
  DFX_HIP(dfx::launch_sfm_error(d, (int)img0->w, (int)img0->h, params->huber_delta, blocks, c->partials, tgt, c->stream, done));
  return finish_result_polled(c, out, sizeof(dfx_corr_item), done);
}

// ---- batched EvaluateError / SE3 step -------------------------------------------------------------------------------
namespace {
// Hands n SimplePairDev to the kernels through the staging ring; simple_launched(slot) must follow the launches.  Default: the kernels read
// the pinned slot itself (simple_zerocopy above).  Otherwise, as in the batched step (dfx_sfm_step_batch_async): a device copy made on the
// context's copy stream, beside the kernels of the previous launch, and the launch stream waits for it.  (Rounds 1-4a copied on the launch
// stream: the 40 KB copy sat between the finalize kernel of one call and the reduction kernel of the next, 18-19 us of idle GPU per call
// in a kernel trace of back-to-back calls, 11.6 with the copy stream, profiles/r04_launch_gaps.txt.)
int upload_simple(dfx_ctx* c, const std::vector<dfx::SimplePairDev>& descs, const dfx::SimplePairDev** dev_out, int* slot_out) {
  int rc, slot;
  char* host;
  const size_t bytes = sizeof(dfx::SimplePairDev) * descs.size();
  if ((rc = stage_acquire(c, bytes, &slot, &host))) return rc;
  std::memcpy(host, descs.data(), bytes);
  if (c->sdesc_cap < bytes) {
    DFX_HIP(hipStreamSynchronize(c->stream));
    DFX_HIP(hipStreamSynchronize(c->copy_stream));
    if (c->sdesc_dev) DFX_HIP(hipFree(c->sdesc_dev));
    c->sdesc_dev = nullptr;
    const size_t cap = (bytes * 2 + 255) & ~(size_t)255;
    DFX_HIP(hipMalloc((void**)&c->sdesc_dev, cap * kStageSlots));
    c->sdesc_cap = cap;
    for (int i = 0; i < kStageSlots; ++i) c->sdesc_busy[i] = false;
  }
  if (simple_zerocopy(c)) {
    void* hdev = nullptr;
    DFX_HIP(hipHostGetDevicePointer(&hdev, host, 0));
    *dev_out = reinterpret_cast<const dfx::SimplePairDev*>(hdev);
    *slot_out = slot;
    return DFX_OK;
  }
  char* dd = c->sdesc_dev + (size_t)slot * c->sdesc_cap;
  if (c->sdesc_busy[slot]) DFX_HIP(hipStreamWaitEvent(c->copy_stream, c->sdesc_done[slot], 0));   // the last kernels that read this slot's device copy
  DFX_HIP(hipMemcpyAsync(dd, host, bytes, hipMemcpyHostToDevice, c->copy_stream));
  DFX_HIP(hipEventRecord(c->stage_ev[slot], c->copy_stream));   // the host slot is free again once the copy has run
  c->stage_used[slot] = true;
  c->stage_rel[slot] = false;
  DFX_HIP(hipStreamWaitEvent(c->stream, c->stage_ev[slot], 0));
  *dev_out = reinterpret_cast<const dfx::SimplePairDev*>(dd);
  *slot_out = slot;
  return DFX_OK;
}

// behind the kernels that read the slot's device copy
int simple_launched(dfx_ctx* c, int slot) {
  if (simple_zerocopy(c)) return stage_release(c, slot);   // the kernels read the staging slot itself: free again behind them
  DFX_HIP(hipEventRecord(c->sdesc_done[slot], c->stream));
  c->sdesc_busy[slot] = true;
  return DFX_OK;
}

// workgroups per pair of a batched simple kernel: ~20 per CU over the whole batch, at least one wave-row of pixels each.  What matters is how
// the row walk's items fall out of it (row_walk: waves per pair / bands -> row segments): 20 per CU = 40 workgroups per 640x480 pair of a
// 128-pair batch = 16 segments of exactly 30 rows and 80 waves per CU = four full rounds of the SE3 step's 20 resident waves; 24 (rounds 3-4a)
// gave 19 segments of 26 rows with a short last one (profiles/r04_launch_shape.txt: SE3 step 165 -> 160 us, EvaluateError 92 -> 88).
int batch_blocks(const dfx_ctx* c, uint32_t W, uint32_t H, int n, int wgs_per_cu = 20) {
  int b = simple_blocks(W, H);
  const int cap = (wgs_per_cu * c->cu_count + n - 1) / n;
  if (b > cap) b = cap;
  return b < 1 ? 1 : b;
}
}  // namespace

DFX_API int dfx_sfm_error_batch_async(dfx_ctx* c, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n, dfx_corr_item* out_items_dev) {
  if (!c || !params || !pairs || !out_items_dev) return fail(DFX_E_INVALID, "dfx_sfm_error_batch: null argument");
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "batch size %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = ray_table_gc(c))) return rc;
  if (!img_ok(&pairs[0].img0)) return fail(DFX_E_INVALID, "pair 0: img0 null or empty");
  const uint32_t W = pairs[0].img0.w, H = pairs[0].img0.h;
  std::vector<dfx::SimplePairDev> descs((size_t)n);
  for (int p = 0; p < n; ++p) {
    dfx_se3 p10;
    float R10[9], HM[9];
    relative_pose(pairs[p].pose0, pairs[p].pose1, R10, p10.t, nullptr, HM);
    p10.q[0] = p10.q[1] = p10.q[2] = 0; p10.q[3] = 1;
    if (pairs[p].img0.w != W || pairs[p].img0.h != H) return fail(DFX_E_INVALID, "pair %d: image size differs from pair 0 (one pyramid level per batch)", p);
    if ((rc = fill_simple(c, &p10, &pairs[p].cam, &pairs[p].img0, &pairs[p].img1, &pairs[p].dpt0, nullptr, nullptr, &descs[p], R10))) {
      g_last_error = "pair " + std::to_string(p) + ": " + g_last_error;
      return rc;
    }
  }
  // EvaluateError: 15 per CU (30 workgroups = segments of exactly 16 rows per 640x480 pair of a 128-pair batch) against the SE3 step's 20: 82.0-82.5 against 84.4-85.1 us
  // in same-box sweeps on two boxes (12: 88, 14: 82, 16: 86, 18: 83; the SE3 step is flat from 15 to 30; profiles/r06b_small_ops_shape.txt)
  const int blocks = batch_blocks(c, W, H, n, 15);
  if ((rc = grow_partials(c, (size_t)n * blocks * dfx::kSimpleRow * sizeof(float)))) return rc;
  const dfx::SimplePairDev* dd;
  int slot;
  hipEvent_t eb, ee;
  if ((rc = prof_events(c, &eb, &ee))) return rc;          // before the slot is handed out: no exit between acquiring a slot and releasing it but the launch
  if ((rc = upload_simple(c, descs, &dd, &slot))) return rc;
  const hipError_t le = dfx::launch_sfm_error_batch(dd, n, (int)W, (int)H, params->huber_delta, blocks, c->partials, out_items_dev, c->stream, eb, ee);
  rc = simple_launched(c, slot);                            // the slot gets its event whether or not the launch went out
  if (le != hipSuccess) { if (ee) (void)hipEventRecord(ee, c->stream); return fail(DFX_E_HIP, "k_sfm_error_batch launch failed: %s", hipGetErrorString(le)); }
  return rc;
}

DFX_API int dfx_sfm_error_batch(dfx_ctx* c, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n, dfx_corr_item* out_items_host) {
  if (!c || !out_items_host) return fail(DFX_E_INVALID, "dfx_sfm_error_batch: null argument");
  if (n <= 0) return fail(DFX_E_INVALID, "batch size %d", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const size_t bytes = sizeof(dfx_corr_item) * (size_t)n;
  if (bytes <= kDirectResultMax) {
    void* tgt;
    if ((rc = result_target(c, bytes, &tgt))) return rc;
    if ((rc = dfx_sfm_error_batch_async(c, params, pairs, n, (dfx_corr_item*)tgt))) return rc;
    return finish_result(c, out_items_host, bytes);
  }
  if (c->items_bytes < bytes) DFX_HIP(hipStreamSynchronize(c->stream));
  if ((rc = grow_dev((void**)&c->items_dev, &c->items_bytes, bytes, c->stream))) return rc;
  if ((rc = dfx_sfm_error_batch_async(c, params, pairs, n, (dfx_corr_item*)c->items_dev))) return rc;
  return fetch_result(c, c->items_dev, out_items_host, bytes);
}

DFX_API int dfx_se3_step_batch_async(dfx_ctx* c, const dfx_se3_pair* pairs, int n, float huber_delta, void* out_items_dev) {
  if (!c || !pairs || !out_items_dev) return fail(DFX_E_INVALID, "dfx_se3_step_batch: null argument");
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "batch size %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = ray_table_gc(c))) return rc;
  if (!img_ok(&pairs[0].img0)) return fail(DFX_E_INVALID, "pair 0: img0 null or empty");
  const uint32_t W = pairs[0].img0.w, H = pairs[0].img0.h;
  std::vector<dfx::SimplePairDev> descs((size_t)n);
  for (int p = 0; p < n; ++p) {
    if (pairs[p].img0.w != W || pairs[p].img0.h != H) return fail(DFX_E_INVALID, "pair %d: image size differs from pair 0 (one pyramid level per batch)", p);
    if ((rc = fill_simple(c, &pairs[p].pose_10, &pairs[p].cam, &pairs[p].img0, &pairs[p].img1, &pairs[p].dpt0, &pairs[p].grad1, nullptr, &descs[p]))) {
      g_last_error = "pair " + std::to_string(p) + ": " + g_last_error;
      return rc;
    }
    if (!descs[p].grad1) return fail(DFX_E_INVALID, "pair %d: grad1 is null", p);
  }
  const int blocks = batch_blocks(c, W, H, n);
  if ((rc = grow_partials(c, (size_t)n * blocks * dfx::kSimpleRow * sizeof(float)))) return rc;
  const dfx::SimplePairDev* dd;
  int slot;
  hipEvent_t eb, ee;
  if ((rc = prof_events(c, &eb, &ee))) return rc;
  if ((rc = upload_simple(c, descs, &dd, &slot))) return rc;
  const hipError_t le = dfx::launch_se3_step_batch(dd, n, (int)W, (int)H, huber_delta, blocks, c->partials, out_items_dev, c->stream, eb, ee);
  rc = simple_launched(c, slot);
  if (le != hipSuccess) { if (ee) (void)hipEventRecord(ee, c->stream); return fail(DFX_E_HIP, "k_se3_step_batch launch failed: %s", hipGetErrorString(le)); }
  return rc;
}

DFX_API int dfx_se3_step_batch(dfx_ctx* c, const dfx_se3_pair* pairs, int n, float huber_delta, void* out_items_host) {
  if (!c || !out_items_host) return fail(DFX_E_INVALID, "dfx_se3_step_batch: null argument");
  if (n <= 0) return fail(DFX_E_INVALID, "batch size %d", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const size_t bytes = dfx_item_size(6) * (size_t)n;
  if (bytes <= kDirectResultMax) {
    void* tgt;
    if ((rc = result_target(c, bytes, &tgt))) return rc;
    if ((rc = dfx_se3_step_batch_async(c, pairs, n, huber_delta, tgt))) return rc;
    return finish_result(c, out_items_host, bytes);
  }
  if (c->items_bytes < bytes) DFX_HIP(hipStreamSynchronize(c->stream));
  if ((rc = grow_dev((void**)&c->items_dev, &c->items_bytes, bytes, c->stream))) return rc;
  if ((rc = dfx_se3_step_batch_async(c, pairs, n, huber_delta, c->items_dev))) return rc;
  return fetch_result(c, c->items_dev, out_items_host, bytes);
}

// ---- SE3Aligner ------------------------------------------------------------------------------------------------
DFX_API int dfx_se3_step(dfx_ctx* c, const dfx_se3* pose_10, const dfx_cam* cam, const dfx_img* img0, const dfx_img* img1,
                         const dfx_img* dpt0, const dfx_img* grad1, float huber_delta, void* out_item) {
  if (!c || !out_item || !grad1) return fail(DFX_E_INVALID, "dfx_se3_step: null argument");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = ray_table_gc(c))) return rc;
  dfx::SimplePairDev d;
  if ((rc = fill_simple(c, pose_10, cam, img0, img1, dpt0, grad1, nullptr, &d))) return rc;
  const int blocks = simple_blocks(img0->w, img0->h);
  const size_t pbytes = (size_t)blocks * dfx::kSimpleRow * sizeof(float);
  if ((rc = grow_partials(c, pbytes))) return rc;
  void* tgt;
  if ((rc = result_target(c, dfx_item_size(6), &tgt))) return rc;
  dfx::DoneFlag done;
  if ((rc = new_done_flag(c, &done))) return rc;
  DFX_HIP(dfx::launch_se3_step(d, (int)img0->w, (int)img0->h, huber_delta, blocks, c->partials, tgt, c->stream, done));
  return finish_result_polled(c, out_item, dfx_item_size(6), done);
}

DFX_API int dfx_se3_warp(dfx_ctx* c, const dfx_se3* pose_10, const dfx_cam* cam, const dfx_img* img0, const dfx_img* img1,
                         const dfx_img* dpt0, const dfx_img* img2_out, dfx_corr_item* out) {
  if (!c || !out || !img2_out) return fail(DFX_E_INVALID, "dfx_se3_warp: null argument");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = ray_table_gc(c))) return rc;
  dfx::SimplePairDev d;
  if ((rc = fill_simple(c, pose_10, cam, img0, img1, dpt0, nullptr, img2_out, &d))) return rc;
  if ((rc = img_note_write(c, img2_out))) return rc;
  const int blocks = simple_blocks(img0->w, img0->h);
  const size_t pbytes = (size_t)blocks * dfx::kSimpleRow * sizeof(float);
  if ((rc = grow_partials(c, pbytes))) return rc;
  void* tgt;
  if ((rc = result_target(c, sizeof(dfx_corr_item), &tgt))) return rc;
  dfx::DoneFlag done;
  if ((rc = new_done_flag(c, &done))) return rc;
  DFX_HIP(dfx::launch_se3_warp(d, (int)img0->w, (int)img0->h, blocks, c->partials, tgt, c->stream, done));
  return finish_result_polled(c, out, sizeof(dfx_corr_item), done);
}

namespace {
void rot_to_quat(const double* R, float* qout) {   // rotation matrix -> unit quaternion (x y z w)
  double q[4];
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
  else if (R[0] > R[4] && R[0] > R[8]) { const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
  else if (R[4] > R[8]) { const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; }
  else { const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; }
  const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) qout[i] = (float)(q[i] / qn);
}

// Workgroups per tracker and iteration: every workgroup of an iteration first folds one partial row per workgroup of the previous one (the update is computed
// redundantly instead of in a launch of its own), and the evaluation is latency-bound (4-6 us); the count is capped at 256 (sweep: profiles/r05_tracker.txt)
int track_blocks(uint32_t W, uint32_t H) {
  const int cap = 256;   // 256: 0.209 ms per 640x480 frame (1024: 0.24, 512: 0.222, 384: 0.218, 192: 0.218, 128: 0.23; profiles/r05_tracker.txt)
  const int b = simple_blocks(W, H);
  return b < cap ? b : cap;
}

// n independent trackers with a common schedule; levels is candidate-major: levels[k * n_levels + l].
int track_frames_impl(dfx_ctx* c, int n, const dfx_se3* pose_init, const dfx_track_level* levels, int n_levels, float huber_delta,
                      dfx_track_result* out) {
  if (!c || !pose_init || !levels || !out) return fail(DFX_E_INVALID, "dfx_track_frame: null argument");
  if (n <= 0 || n > 4096) return fail(DFX_E_INVALID, "candidate count %d out of range [1,4096]", n);
  if (n_levels <= 0 || n_levels > 16) return fail(DFX_E_INVALID, "n_levels %d out of range [1,16]", n_levels);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if ((rc = ray_table_gc(c))) return rc;
  // one staged upload: [SimplePairDev x n_levels x n (level-major)][TrackState x n]; on the device the states continue as an array over iterations
  // (iteration k reads states[k - 1] and the partial rows of evaluation k - 1, workgroup 0 writes states[k]; dfx_misc_kernels.hip)
  const size_t sbytes = dfx::track_state_bytes();
  const size_t off_state = ((sizeof(dfx::SimplePairDev) * (size_t)n * n_levels + 15) / 16) * 16;
  const size_t total = off_state + sbytes * (size_t)n;
  int total_iters = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (levels[l].iterations < 0) return fail(DFX_E_INVALID, "candidate 0 level %d: negative iteration count", l);
    if (levels[l].iterations > 4096) return fail(DFX_E_INVALID, "level %d: %d iterations (at most 4096)", l, levels[l].iterations);
    total_iters += levels[l].iterations;
  }
  const size_t dev_total = off_state + sbytes * (size_t)n * ((size_t)total_iters + 1);
  // one tracker (the camera-rate case): descriptors and the initial state travel in the kernel arguments -- no staging slot, no upload in front of the first kernel
  const bool byval = n == 1 && total_iters > 0;
  alignas(16) char local[16 * sizeof(dfx::SimplePairDev) + 512];
  static_assert(sizeof(local) >= 16 * sizeof(dfx::SimplePairDev) + 16 + 256, "n_levels <= 16 descriptors + one state");
  int slot = -1;
  char* host = local;
  if (!byval && (rc = stage_acquire(c, total, &slot, &host))) return rc;
  dfx::SimplePairDev* hdesc = reinterpret_cast<dfx::SimplePairDev*>(host);
  // validate every level of every candidate first (no partial work on a bad argument)
  int max_blocks = 1;
  const dfx_se3 ident{ { 0, 0, 0, 1 }, { 0, 0, 0 } };
  auto bail = [&](int code) { if (slot >= 0) (void)stage_release(c, slot); return code; };
  for (int k = 0; k < n; ++k)
    for (int l = 0; l < n_levels; ++l) {
      const dfx_track_level& L = levels[(size_t)k * n_levels + l];
      const dfx_track_level& L0 = levels[l];
      if (L.iterations != L0.iterations || L.img0.w != L0.img0.w || L.img0.h != L0.img0.h)
        return bail(fail(DFX_E_INVALID, "candidate %d level %d: schedule / image size differs from candidate 0", k, l));
      if ((rc = fill_simple(c, &ident, &L.cam, &L.img0, &L.img1, &L.dpt0, &L.grad1, nullptr, &hdesc[(size_t)l * n + k]))) {
        g_last_error = "candidate " + std::to_string(k) + " level " + std::to_string(l) + ": " + g_last_error;
        return bail(rc);
      }
      const int b = track_blocks(L.img0.w, L.img0.h);
      if (b > max_blocks) max_blocks = b;
    }
  const size_t pfloats = (size_t)n * max_blocks * dfx::kSimpleRow;   // per evaluation; two buffers alternate
  if ((rc = grow_partials(c, 2 * pfloats * sizeof(float)))) return bail(rc);
  if (c->track_bytes < dev_total) {
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(fail(DFX_E_HIP, "dfx_track_frame: stream synchronisation failed"));
    if (c->track_state_dev) (void)hipFree(c->track_state_dev);
    c->track_state_dev = nullptr;
    c->track_bytes = 0;
    if (hipMalloc(&c->track_state_dev, dev_total * 2) != hipSuccess) return bail(fail(DFX_E_HIP, "dfx_track_frame: %zu bytes of tracker state", dev_total * 2));
    c->track_bytes = dev_total * 2;
  }
  for (int k = 0; k < n; ++k) {
    double R[9], t[3] = { pose_init[k].t[0], pose_init[k].t[1], pose_init[k].t[2] };
    quat_to_R(pose_init[k].q, R);
    dfx::track_state_init(host + off_state + sbytes * (size_t)k, R, t);
  }
  if (!byval) {
    if (hipMemcpyAsync(c->track_state_dev, host, total, hipMemcpyHostToDevice, c->stream) != hipSuccess) return bail(fail(DFX_E_HIP, "dfx_track_frame: descriptor upload failed"));
    if ((rc = stage_release(c, slot))) return rc;
  }
  const dfx::SimplePairDev* ddesc = reinterpret_cast<const dfx::SimplePairDev*>(c->track_state_dev);
  char* dstates = (char*)c->track_state_dev + off_state;
  auto state_at = [&](int k) { return dstates + sbytes * (size_t)n * (size_t)k; };
  int it_done = 0, blocks_prev = 0;
  for (int l = n_levels - 1; l >= 0; --l) {
    const int W = (int)levels[l].img0.w, H = (int)levels[l].img0.h;
    const int blocks = track_blocks(levels[l].img0.w, levels[l].img0.h);
    for (int it = 0; it < levels[l].iterations; ++it) {
      // evaluation `it_done` at states[it_done] (= states[0] as uploaded, or the update this launch computes from evaluation it_done - 1)
      const void* sin = state_at(it_done > 0 ? it_done - 1 : 0);
      DFX_HIP(dfx::launch_track_iteration(ddesc + (size_t)l * n, n, sin, state_at(it_done), c->partials + (size_t)((it_done + 1) & 1) * pfloats, blocks_prev, W, H, huber_delta, blocks,
                                          c->partials + (size_t)(it_done & 1) * pfloats, c->stream, byval ? &hdesc[l] : nullptr,
                                          byval && it_done == 0 ? host + off_state : nullptr));
      blocks_prev = blocks;
      ++it_done;
    }
  }
  // the last update goes straight into the pinned, device-mapped result area (no copy operation behind the kernels)
  void* rdev = nullptr;
  if ((rc = result_target(c, sbytes * (size_t)n, &rdev))) return rc;
  dfx::DoneFlag done;
  if (it_done > 0) {
    if (n == 1 && (rc = new_done_flag(c, &done))) return rc;
    DFX_HIP(dfx::launch_track_final(n, state_at(it_done - 1), rdev, c->partials + (size_t)((it_done + 1) & 1) * pfloats, blocks_prev, c->stream, done));
  } else DFX_HIP(hipMemcpyAsync(c->result_host, state_at(0), sbytes * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  if (done.flag) {
    if ((rc = finish_result_polled(c, c->result_host, 0, done))) return rc;   // (the records are read in place below)
  } else if ((rc = wait_stream(c))) return rc;
  const double area = (double)levels[0].img0.w * levels[0].img0.h;
  for (int k = 0; k < n; ++k) {
    double R[9], t[3];
    float residual, inl;
    int fails, iters;
    dfx::track_state_read(c->result_host + sbytes * (size_t)k, R, t, &residual, &inl, &fails, &iters);
    rot_to_quat(R, out[k].pose_ck.q);
    for (int i = 0; i < 3; ++i) out[k].pose_ck.t[i] = (float)t[i];
    out[k].residual = residual;
    out[k].inliers = (uint64_t)(inl + 0.5f);
    out[k].inliers_frac = (float)(inl / area);
    out[k].error = inl > 0 ? residual / inl : INFINITY;
    out[k].iterations = iters;
    out[k].solver_failures = fails;
  }
  return DFX_OK;
}
}  // namespace

DFX_API int dfx_track_frame(dfx_ctx* c, const dfx_se3* pose_init, const dfx_track_level* levels, int n_levels, float huber_delta,
                            dfx_track_result* out) {
  return track_frames_impl(c, 1, pose_init, levels, n_levels, huber_delta, out);
}

DFX_API int dfx_track_frame_batch(dfx_ctx* c, int n, const dfx_se3* pose_init, const dfx_track_level* levels, int n_levels,
                                  float huber_delta, dfx_track_result* out) {
  return track_frames_impl(c, n, pose_init, levels, n_levels, huber_delta, out);
}

// ---- SparseGeometricFactor::linearize: all factors of a round in ONE launch ------------------------------------------------------------
namespace {
// rows_dev != null: rows stay on the device (enqueue only); rows_host != null: one device-to-host copy of all rows, blocking
// gram_dev / gram_host (one of them, with rows_dev = rows_host = null): the rows stay in the context's scratch and their Gram blocks are the result
int sparse_geo_batch_impl(dfx_ctx* c, int cs, const dfx_sparse_geo_factor* f, int n, float huber_delta, float avg_dpt, float* rows_dev, float* rows_host,
                          float* gram_dev = nullptr, float* gram_host = nullptr) {
  if (!c || !f || (!rows_dev && !rows_host && !gram_dev && !gram_host)) return fail(DFX_E_INVALID, "dfx_sparse_geometric_linearize_batch: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "factor count %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const size_t nc = 12 + 2 * (size_t)cs + 1;
  size_t total_pts = 0, host_pts = 0;
  int max_pts = 0;
  for (int k = 0; k < n; ++k) {
    const dfx_sparse_geo_factor& q = f[k];
    if (!q.code0 || !q.code1 || !q.points_xy) return fail(DFX_E_INVALID, "factor %d: null code / points", k);
    if (q.n_points <= 0 || q.n_points > (1 << 20)) return fail(DFX_E_INVALID, "factor %d: n_points %d out of range", k, q.n_points);
    if (!img_ok(&q.prx0_orig)) return fail(DFX_E_INVALID, "factor %d: prx0_orig: null or empty image view", k);
    const uint32_t W = q.prx0_orig.w, H = q.prx0_orig.h;
    if ((rc = check_img(&q.prx0_orig, "prx0_orig", W, H, 4)) || (rc = check_img(&q.prx1_orig, "prx1_orig", W, H, 4)) ||
        (rc = check_img(&q.prx0_jac, "prx0_jac", W * (uint32_t)cs, H, 4)) || (rc = check_img(&q.prx1_jac, "prx1_jac", W * (uint32_t)cs, H, 4)) ||
        (rc = check_img(&q.dpt1_grad, "dpt1_grad", W, H, 8))) {
      g_last_error = "factor " + std::to_string(k) + ": " + g_last_error;
      return rc;
    }
    if (((uintptr_t)q.dpt1_grad.ptr | q.dpt1_grad.pitch_bytes) & 7) return fail(DFX_E_INVALID, "factor %d: dpt1_grad: pointer/pitch must be 8-byte aligned", k);
    if ((((uintptr_t)q.prx0_jac.ptr | q.prx0_jac.pitch_bytes) | ((uintptr_t)q.prx1_jac.ptr | q.prx1_jac.pitch_bytes)) & 15)
      return fail(DFX_E_INVALID, "factor %d: prx_jac: pointer/pitch must be 16-byte aligned", k);
    if (!q.points_on_device) {
      for (int i = 0; i < q.n_points; ++i)
        if (q.points_xy[2 * i] < 0 || q.points_xy[2 * i] >= (int)W || q.points_xy[2 * i + 1] < 0 || q.points_xy[2 * i + 1] >= (int)H)
          return fail(DFX_E_INVALID, "factor %d: point %d = (%d, %d) outside the %ux%u image", k, i, q.points_xy[2 * i], q.points_xy[2 * i + 1], W, H);
      host_pts += (size_t)q.n_points;
    }
    total_pts += (size_t)q.n_points;
    max_pts = std::max(max_pts, (int)q.n_points);
  }
  const size_t dsz = dfx::sparse_geo_desc_bytes();
  const size_t off_pts = ((size_t)n * dsz + 255) & ~(size_t)255;
  const size_t up = off_pts + host_pts * 8;
  const size_t off_rows = (up + 255) & ~(size_t)255;
  const size_t row_bytes = total_pts * nc * sizeof(float);
  const size_t ne = nc * (nc + 1) / 2;
  const size_t off_gram = (off_rows + (rows_dev ? 0 : row_bytes) + 255) & ~(size_t)255;
  const size_t gram_bytes = (size_t)n * ne * sizeof(float);
  const size_t need = (gram_host ? off_gram + gram_bytes : off_rows + (rows_dev ? 0 : row_bytes));
  if (c->sg_bytes < need) DFX_HIP(hipStreamSynchronize(c->stream));
  if ((rc = grow_dev((void**)&c->sg_dev, &c->sg_bytes, need, c->stream))) return rc;
  float* const rows_base = rows_dev ? rows_dev : reinterpret_cast<float*>(c->sg_dev + off_rows);
  int slot;
  char* host;
  if ((rc = stage_acquire(c, up, &slot, &host))) return rc;
  size_t pt_off = 0, row_off = 0;
  for (int k = 0; k < n; ++k) {
    const dfx_sparse_geo_factor& q = f[k];
    float R10[9], t10[3], M[9], HM[9];
    relative_pose(q.pose0, q.pose1, R10, t10, M, HM);
    const float cam6[6] = { q.cam.fx, q.cam.fy, q.cam.u0, q.cam.v0, q.cam.w, q.cam.h };
    const int* pts_dev = (const int*)q.points_xy;
    if (!q.points_on_device) {
      std::memcpy(host + off_pts + pt_off * 8, q.points_xy, (size_t)q.n_points * 8);
      pts_dev = reinterpret_cast<const int*>(c->sg_dev + off_pts + pt_off * 8);
      pt_off += (size_t)q.n_points;
    }
    dfx::sparse_geo_fill(host + (size_t)k * dsz, R10, t10, M, HM, cam6, q.code0, q.code1, cs, (const float*)q.prx0_orig.ptr, (uint32_t)q.prx0_orig.pitch_bytes,
                         (const float*)q.prx0_jac.ptr, (uint32_t)q.prx0_jac.pitch_bytes, (const float*)q.prx1_orig.ptr, (uint32_t)q.prx1_orig.pitch_bytes,
                         (const float*)q.prx1_jac.ptr, (uint32_t)q.prx1_jac.pitch_bytes, (const float*)q.dpt1_grad.ptr, (uint32_t)q.dpt1_grad.pitch_bytes,
                         pts_dev, q.n_points, (int)q.prx0_orig.w, (int)q.prx0_orig.h, rows_base + row_off * nc, huber_delta, avg_dpt);
    row_off += (size_t)q.n_points;
  }
  {   // the slot gets its event whether or not the copy went out (no exit between acquiring a slot and releasing it)
    const hipError_t ce = hipMemcpyAsync(c->sg_dev, host, up, hipMemcpyHostToDevice, c->stream);
    rc = stage_release(c, slot);
    if (ce != hipSuccess) return fail(DFX_E_HIP, "hipMemcpyAsync (descriptors) failed: %s", hipGetErrorString(ce));
    if (rc) return rc;
  }
  DFX_HIP(dfx::launch_sparse_geometric_batch(cs, c->sg_dev, n, max_pts, c->stream));
  if (gram_dev || gram_host) {
    float* const g = gram_dev ? gram_dev : reinterpret_cast<float*>(c->sg_dev + off_gram);
    DFX_HIP(dfx::launch_rows_gram(cs, c->sg_dev, n, g, c->stream));
    if (!gram_host) return DFX_OK;
    if (gram_bytes <= kDirectResultMax) return fetch_result(c, g, gram_host, gram_bytes);
    DFX_HIP(hipMemcpyAsync(gram_host, g, gram_bytes, hipMemcpyDeviceToHost, c->stream));
    return wait_stream(c);
  }
  if (!rows_host) return DFX_OK;
  if (row_bytes <= kDirectResultMax) return fetch_result(c, rows_base, rows_host, row_bytes);
  // a whole round of rows (18 MB for 120 factors x 500 points at CS = 32): straight into the caller's memory, no bounce through the result area
  DFX_HIP(hipMemcpyAsync(rows_host, rows_base, row_bytes, hipMemcpyDeviceToHost, c->stream));
  if ((rc = wait_stream(c))) return rc;
  return DFX_OK;
}
}  // namespace

DFX_API int dfx_sparse_geometric_linearize_batch_async(dfx_ctx* c, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta, float avg_dpt,
                                                       float* rows_dev) {
  if (!rows_dev) return fail(DFX_E_INVALID, "dfx_sparse_geometric_linearize_batch_async: null row buffer");
  return sparse_geo_batch_impl(c, cs, factors, n, huber_delta, avg_dpt, rows_dev, nullptr);
}

DFX_API int dfx_sparse_geometric_linearize_batch(dfx_ctx* c, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta, float avg_dpt,
                                                 float* rows_host) {
  if (!rows_host) return fail(DFX_E_INVALID, "dfx_sparse_geometric_linearize_batch: null row buffer");
  return sparse_geo_batch_impl(c, cs, factors, n, huber_delta, avg_dpt, nullptr, rows_host);
}

// one factor per blocking call (the reference's pattern): the same kernel with a batch of one
DFX_API int dfx_sparse_geometric_gram_batch_async(dfx_ctx* c, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta, float avg_dpt, float* gram_dev) {
  if (!gram_dev) return fail(DFX_E_INVALID, "dfx_sparse_geometric_gram_batch_async: null output");
  return sparse_geo_batch_impl(c, cs, factors, n, huber_delta, avg_dpt, nullptr, nullptr, gram_dev, nullptr);
}
DFX_API int dfx_sparse_geometric_gram_batch(dfx_ctx* c, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta, float avg_dpt, float* gram_host) {
  if (!gram_host) return fail(DFX_E_INVALID, "dfx_sparse_geometric_gram_batch: null output");
  return sparse_geo_batch_impl(c, cs, factors, n, huber_delta, avg_dpt, nullptr, nullptr, nullptr, gram_host);
}

DFX_API int dfx_sparse_geometric_linearize(dfx_ctx* c, int cs, const dfx_se3* pose0, const dfx_se3* pose1, const float* code0,
                                           const float* code1, const dfx_cam* cam, const int32_t* points_xy, int n_points,
                                           const dfx_img* prx0_orig, const dfx_img* prx0_jac, const dfx_img* prx1_orig,
                                           const dfx_img* prx1_jac, const dfx_img* dpt1_grad, float huber_delta, float avg_dpt,
                                           float* rows_host) {
  if (!c || !pose0 || !pose1 || !code0 || !code1 || !cam || !points_xy || !rows_host || !prx0_orig || !prx0_jac || !prx1_orig || !prx1_jac || !dpt1_grad)
    return fail(DFX_E_INVALID, "dfx_sparse_geometric_linearize: null argument");
  dfx_sparse_geo_factor f;
  std::memset(&f, 0, sizeof(f));
  f.pose0 = *pose0; f.pose1 = *pose1; f.cam = *cam; f.code0 = code0; f.code1 = code1; f.points_xy = points_xy; f.n_points = n_points; f.points_on_device = 0;
  f.prx0_orig = *prx0_orig; f.prx0_jac = *prx0_jac; f.prx1_orig = *prx1_orig; f.prx1_jac = *prx1_jac; f.dpt1_grad = *dpt1_grad;
  return sparse_geo_batch_impl(c, cs, &f, 1, huber_delta, avg_dpt, nullptr, rows_host);
}

// ---- image-proc ------------------------------------------------------------------------------------------------
static int upload_code(dfx_ctx* c, int cs, const float* code, float** code_dev_out) {
  int rc, slot;
  char* host;
  if (!c->code_dev) DFX_HIP(hipMalloc((void**)&c->code_dev, sizeof(float) * 64 * kStageSlots));
  if ((rc = stage_acquire(c, sizeof(float) * 64, &slot, &host))) return rc;
  std::memcpy(host, code, sizeof(float) * (size_t)cs);
  float* dst = c->code_dev + (size_t)slot * 64;
  DFX_HIP(hipMemcpyAsync(dst, host, sizeof(float) * (size_t)cs, hipMemcpyHostToDevice, c->stream));
  if ((rc = stage_release(c, slot))) return rc;
  *code_dev_out = dst;
  return DFX_OK;
}

DFX_API int dfx_update_depth(dfx_ctx* c, int cs, const float* code, const dfx_img* prx_orig, const dfx_img* prx_jac,
                             float avg_dpt, const dfx_img* dpt_out) {
  if (!c || !code) return fail(DFX_E_INVALID, "dfx_update_depth: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (!img_ok(prx_orig)) return fail(DFX_E_INVALID, "prx_orig: null or empty image view");
  const uint32_t W = prx_orig->w, H = prx_orig->h;
  if ((rc = check_img(prx_orig, "prx_orig", W, H, 4))) return rc;
  if ((rc = check_img(dpt_out, "dpt_out", W, H, 4))) return rc;
  if ((rc = check_img(prx_jac, "prx_jac", W * (uint32_t)cs, H, 4))) return rc;
  if (((uintptr_t)prx_jac->ptr | prx_jac->pitch_bytes) & 15) return fail(DFX_E_INVALID, "prx_jac: pointer/pitch must be 16-byte aligned");
  float* code_dev;
  if ((rc = upload_code(c, cs, code, &code_dev))) return rc;
  if ((rc = img_note_write(c, dpt_out))) return rc;
  DFX_HIP(dfx::launch_update_depth(cs, code_dev, (const float*)prx_orig->ptr, (uint32_t)prx_orig->pitch_bytes,
                                   (const float*)prx_jac->ptr, (uint32_t)prx_jac->pitch_bytes, avg_dpt, (float*)dpt_out->ptr,
                                   (uint32_t)dpt_out->pitch_bytes, (int)W, (int)H, c->stream));
  // the reference's UpdateDepth returns after CudaCheckLastError = cudaDeviceSynchronize (cu_image_proc.cpp:276)
  if ((rc = wait_stream(c))) return rc;
  return DFX_OK;
}

// ---- batched decoder + the reference's real hot entry -----------------------------------------------------------------------
namespace {
// jobs: host-side list of decode jobs of one image size; uploads the descriptors through the staging ring and enqueues ONE launch
// held_slots (optional, zero-copy job lists only): the slot is NOT released here -- its event would sit between this launch and the caller's next kernel, and a
// marker between two dependent kernels costs the GPU 4-5 us (5.6 us between a pair's decode and its step in the kernel trace) -- the caller releases it behind
// its own launches.
int update_depth_jobs(dfx_ctx* c, int cs, const std::vector<dfx::DepthJobDev>& jobs, float avg_dpt, uint32_t W, uint32_t H, std::vector<int>* held_slots = nullptr) {
  const int n = (int)jobs.size();
  if (n == 0) return DFX_OK;
  int rc, slot;
  char* host;
  const size_t bytes = sizeof(dfx::DepthJobDev) * (size_t)n;
  if ((rc = stage_acquire(c, bytes, &slot, &host))) return rc;
  std::memcpy(host, jobs.data(), bytes);
  if (simple_zerocopy(c)) {   // the kernel reads the job list out of the pinned slot (see simple_zerocopy)
    void* hdev = nullptr;
    DFX_HIP(hipHostGetDevicePointer(&hdev, host, 0));
    DFX_HIP(dfx::launch_update_depth_batch(cs, reinterpret_cast<const dfx::DepthJobDev*>(hdev), n, avg_dpt, (int)W, (int)H, c->stream));
    if (held_slots) { held_slots->push_back(slot); return DFX_OK; }
    return stage_release(c, slot);
  }
  if (c->jobs_cap < (size_t)n) {
    DFX_HIP(hipStreamSynchronize(c->stream));
    if (c->jobs_dev) DFX_HIP(hipFree(c->jobs_dev));
    c->jobs_dev = nullptr;
    const size_t cap = (size_t)n * 2;
    DFX_HIP(hipMalloc((void**)&c->jobs_dev, sizeof(dfx::DepthJobDev) * cap * kStageSlots));
    c->jobs_cap = cap;
  }
  dfx::DepthJobDev* dd = c->jobs_dev + (size_t)slot * c->jobs_cap;
  DFX_HIP(hipMemcpyAsync(dd, host, bytes, hipMemcpyHostToDevice, c->stream));
  if ((rc = stage_release(c, slot))) return rc;
  DFX_HIP(dfx::launch_update_depth_batch(cs, dd, n, avg_dpt, (int)W, (int)H, c->stream));
  return DFX_OK;
}

int fill_depth_job(int cs, const float* code, const dfx_img* prx_orig, const dfx_img* prx_jac, const dfx_img* dpt_out, uint32_t W, uint32_t H,
                   dfx::DepthJobDev* j) {
  int rc;
  if (!code) return fail(DFX_E_INVALID, "null code");
  if ((rc = check_img(prx_orig, "prx_orig", W, H, 4))) return rc;
  if ((rc = check_img(dpt_out, "dpt_out", W, H, 4))) return rc;
  if ((rc = check_img(prx_jac, "prx_jac", W * (uint32_t)cs, H, 4))) return rc;
  if (((uintptr_t)prx_jac->ptr | prx_jac->pitch_bytes) & 15) return fail(DFX_E_INVALID, "prx_jac: pointer/pitch must be 16-byte aligned");
  std::memset(j->code, 0, sizeof(j->code));
  std::memcpy(j->code, code, sizeof(float) * (size_t)cs);
  j->prx = (const float*)prx_orig->ptr; j->jac = (const float*)prx_jac->ptr; j->out = (float*)dpt_out->ptr;
  j->pitch_prx = (uint32_t)prx_orig->pitch_bytes; j->pitch_jac = (uint32_t)prx_jac->pitch_bytes; j->pitch_out = (uint32_t)dpt_out->pitch_bytes;
  j->_pad = 0;
  return DFX_OK;
}
}  // namespace

DFX_API int dfx_update_depth_batch_async(dfx_ctx* c, int cs, int n, const float* codes, const dfx_img* prx_orig, const dfx_img* prx_jac, float avg_dpt,
                                         const dfx_img* dpt_out) {
  if (!c || !codes || !prx_orig || !prx_jac || !dpt_out) return fail(DFX_E_INVALID, "dfx_update_depth_batch: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "batch size %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (!img_ok(&prx_orig[0])) return fail(DFX_E_INVALID, "job 0: prx_orig null or empty");
  const uint32_t W = prx_orig[0].w, H = prx_orig[0].h;
  std::vector<dfx::DepthJobDev> jobs((size_t)n);
  for (int k = 0; k < n; ++k)
    if ((rc = fill_depth_job(cs, codes + (size_t)k * cs, &prx_orig[k], &prx_jac[k], &dpt_out[k], W, H, &jobs[k])) || (rc = img_note_write(c, &dpt_out[k]))) {
      g_last_error = "job " + std::to_string(k) + ": " + g_last_error;
      return rc;
    }
  return update_depth_jobs(c, cs, jobs, avg_dpt, W, H);
}

static int sfm_linearize_batch_impl(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, const dfx_img* prx0_orig,
                                    const float* codes0, int n, void* out_items_dev, bool allow_defer);

DFX_API int dfx_sfm_linearize_batch_async(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, const dfx_img* prx0_orig,
                                          const float* codes0, int n, void* out_items_dev) {
  return sfm_linearize_batch_impl(c, cs, params, pairs, prx0_orig, codes0, n, out_items_dev, true);
}

static int sfm_linearize_batch_impl(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, const dfx_img* prx0_orig,
                                    const float* codes0, int n, void* out_items_dev, bool allow_defer) {
  if (!c || !params || !pairs || !prx0_orig || !codes0 || !out_items_dev) return fail(DFX_E_INVALID, "dfx_sfm_linearize_batch: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "batch size %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  // UpdateDepthMaps once per DISTINCT keyframe depth map of the batch (the reference decodes it again for every factor that
  // shares the keyframe, photometric_factor.cpp:229,332-341): pairs that share dpt0 must agree on code, prx_orig and prx_jac.
  std::map<std::pair<uint32_t, uint32_t>, std::vector<dfx::DepthJobDev>> jobs;   // one decoder launch per image size (pyramid level) of the batch
  std::unordered_map<const void*, int> first;   // depth-map pointer -> first pair that writes it
  auto same_img = [](const dfx_img& a, const dfx_img& b) { return a.ptr == b.ptr && a.pitch_bytes == b.pitch_bytes && a.w == b.w && a.h == b.h; };
  for (int p = 0; p < n; ++p) {
    if (!img_ok(&pairs[p].img0)) return fail(DFX_E_INVALID, "pair %d: img0 null or empty", p);
    const uint32_t W = pairs[p].img0.w, H = pairs[p].img0.h;
    const void* key = pairs[p].dpt0.ptr;
    auto hit = first.find(key);
    if (hit == first.end()) {
      dfx::DepthJobDev j;
      if ((rc = fill_depth_job(cs, codes0 + (size_t)p * cs, &prx0_orig[p], &pairs[p].prx0_jac, &pairs[p].dpt0, W, H, &j)) || (rc = img_note_write(c, &pairs[p].dpt0))) {
        g_last_error = "pair " + std::to_string(p) + ": " + g_last_error;
        return rc;
      }
      first.emplace(key, p); jobs[std::make_pair(W, H)].push_back(j);
    } else {
      const int q = hit->second;
      if (std::memcmp(codes0 + (size_t)p * cs, codes0 + (size_t)q * cs, sizeof(float) * (size_t)cs) != 0 || !same_img(pairs[p].dpt0, pairs[q].dpt0) ||
          !same_img(prx0_orig[p], prx0_orig[q]) || !same_img(pairs[p].prx0_jac, pairs[q].prx0_jac))
        return fail(DFX_E_INVALID, "pairs %d and %d write the same depth map from different codes / decoder images", q, p);
    }
  }
  {   // the decoder's job lists take staging slots while the step holds its own: size the ring now (see stage_reserve)
    size_t most = 0;
    for (auto& lv : jobs) most = std::max(most, lv.second.size());
    if ((rc = stage_reserve(c, sizeof(dfx::DepthJobDev) * most))) return rc;
  }
  std::vector<int> held;
  const std::function<int()> decode = [&]() -> int {
    for (auto& lv : jobs) {
      const int r = update_depth_jobs(c, cs, lv.second, params->avg_dpt, lv.first.first, lv.first.second, &held);
      if (r) return r;
    }
    return DFX_OK;
  };
  const std::function<int()> release = [&]() -> int {   // the job lists' slots get their events behind the step's kernels
    int r = DFX_OK;
    for (int sl : held) { const int q = stage_release(c, sl); if (q) r = q; }
    held.clear();
    return r;
  };
  if ((int)jobs.size() > kStageSlots - 2) {   // one staging slot per image size + the step's own must fit the ring: otherwise decode first, as before
    rc = decode();
    const int r2 = release();
    if (rc || r2) return rc ? rc : r2;
    return sfm_step_batch_impl(c, cs, params, pairs, n, out_items_dev, allow_defer);
  }
  return sfm_step_batch_impl(c, cs, params, pairs, n, out_items_dev, allow_defer, nullptr, 0, nullptr, &decode, &release);
}

DFX_API int dfx_sfm_linearize_batch(dfx_ctx* c, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, const dfx_img* prx0_orig,
                                    const float* codes0, int n, void* out_items_host) {
  if (!c || !out_items_host) return fail(DFX_E_INVALID, "dfx_sfm_linearize_batch: null argument");
  if (n <= 0) return fail(DFX_E_INVALID, "batch size %d", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const size_t bytes = dfx_item_size(12 + cs) * (size_t)n;
  if (bytes <= kDirectResultMax) {
    void* tgt;
    if ((rc = result_target(c, bytes, &tgt))) return rc;
    dfx::DoneFlag done;
    if (n == 1 && (rc = new_done_flag(c, &done))) return rc;
    c->done_armed = done.flag ? &done : nullptr;
    rc = sfm_linearize_batch_impl(c, cs, params, pairs, prx0_orig, codes0, n, tgt, false);
    c->done_armed = nullptr;
    if (rc) return rc;
    return finish_result_polled(c, out_items_host, bytes, done);
  }
  if (c->items_bytes < bytes) DFX_HIP(hipStreamSynchronize(c->stream));
  if ((rc = grow_dev((void**)&c->items_dev, &c->items_bytes, bytes, c->stream))) return rc;
  if ((rc = sfm_linearize_batch_impl(c, cs, params, pairs, prx0_orig, codes0, n, c->items_dev, false))) return rc;
  return fetch_result(c, c->items_dev, out_items_host, bytes);
}


DFX_API int dfx_sobel_gradients(dfx_ctx* c, const dfx_img* img, const dfx_img* grad_out) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (!img_ok(img)) return fail(DFX_E_INVALID, "img: null or empty image view");
  if ((rc = check_img(img, "img", img->w, img->h, 4))) return rc;
  if ((rc = check_img(grad_out, "grad", img->w, img->h, 8))) return rc;
  if (((uintptr_t)grad_out->ptr | grad_out->pitch_bytes) & 7) return fail(DFX_E_INVALID, "grad: pointer/pitch must be 8-byte aligned");
  DFX_HIP(dfx::launch_sobel((const float*)img->ptr, (uint32_t)img->pitch_bytes, (float*)grad_out->ptr, (uint32_t)grad_out->pitch_bytes,
                            (int)img->w, (int)img->h, c->stream));
  if ((rc = wait_stream(c))) return rc;
  return DFX_OK;
}

DFX_API int dfx_gaussian_blur_down(dfx_ctx* c, const dfx_img* in, const dfx_img* out) {
  if (!c) return fail(DFX_E_INVALID, "null context");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (!img_ok(in) || !img_ok(out)) return fail(DFX_E_INVALID, "null or empty image view");
  if ((rc = check_img(in, "in", in->w, in->h, 4))) return rc;
  if ((rc = check_img(out, "out", out->w, out->h, 4))) return rc;
  if ((rc = img_note_write(c, out))) return rc;
  DFX_HIP(dfx::launch_blur_down((const float*)in->ptr, (uint32_t)in->pitch_bytes, (int)in->w, (int)in->h, (float*)out->ptr,
                                (uint32_t)out->pitch_bytes, (int)out->w, (int)out->h, c->stream));
  if ((rc = wait_stream(c))) return rc;
  return DFX_OK;
}

// ---- Frame::FillPyramids for n frames: one launch per pyramid level (core/mapping/frame.h:80-94, core/deepfactors.cpp:616-630) ----------------
DFX_API int dfx_build_pyramid_batch_async(dfx_ctx* c, const dfx_pyramid* frames, int n) {
  if (!c || !frames) return fail(DFX_E_INVALID, "dfx_build_pyramid_batch: null argument");
  if (n <= 0 || n > 65535) return fail(DFX_E_INVALID, "frame count %d out of range [1,65535]", n);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  const int L = frames[0].levels;
  if (L < 1 || L > DFX_MAX_PYR_LEVELS) return fail(DFX_E_INVALID, "pyramid of %d levels (1 .. %d)", L, DFX_MAX_PYR_LEVELS);
  const size_t dbytes = sizeof(dfx::PyrLevelDev) * (size_t)n * L;
  c->pyr_build.assign(dbytes, 0);   // (padding bytes zeroed: the block is compared with the previous build's)
  dfx::PyrLevelDev* hd = reinterpret_cast<dfx::PyrLevelDev*>(c->pyr_build.data());
  bool rows_ok[DFX_MAX_PYR_LEVELS];
  for (int i = 0; i < DFX_MAX_PYR_LEVELS; ++i) rows_ok[i] = true;
  std::vector<const void*> written;
  for (int k = 0; k < n && !rc; ++k) {
    const dfx_pyramid& f = frames[k];
    if (f.levels != L) rc = fail(DFX_E_INVALID, "frame %d: %d levels, frame 0 has %d (one schedule per batch)", k, f.levels, L);
    for (int i = 0; i < L && !rc; ++i) {
      if (!img_ok(&f.img[i])) { rc = fail(DFX_E_INVALID, "frame %d: level %d image null or empty", k, i); break; }
      const uint32_t W = f.img[i].w, H = f.img[i].h;
      if (W != frames[0].img[i].w || H != frames[0].img[i].h) { rc = fail(DFX_E_INVALID, "frame %d: level %d is %ux%u, frame 0 has %ux%u", k, i, W, H, frames[0].img[i].w, frames[0].img[i].h); break; }
      if (i > 0 && (W != f.img[i - 1].w / 2 || H != f.img[i - 1].h / 2)) { rc = fail(DFX_E_INVALID, "frame %d: level %d is %ux%u, half of level %d is %ux%u", k, i, W, H, i - 1, f.img[i - 1].w / 2, f.img[i - 1].h / 2); break; }
      if ((rc = check_img(&f.img[i], "img", W, H, 4))) break;
      dfx::PyrLevelDev& d = hd[(size_t)i * n + k];
      d.in = (const float*)f.img[i].ptr; d.pitch_in = (uint32_t)f.img[i].pitch_bytes; d.W = (int)W; d.H = (int)H;
      d.grad = nullptr; d.pitch_grad = 0;
      if (f.grad[i].ptr) {   // (UploadLiveFrame leaves the live frame's level-0 gradient out, deepfactors.cpp:620-625: a null view skips a level's gradient)
        if ((rc = check_img(&f.grad[i], "grad", W, H, 8))) break;
        if (((uintptr_t)f.grad[i].ptr | f.grad[i].pitch_bytes) & 7) { rc = fail(DFX_E_INVALID, "grad: pointer/pitch must be 8-byte aligned"); break; }
        d.grad = (float*)f.grad[i].ptr; d.pitch_grad = (uint32_t)f.grad[i].pitch_bytes;
        written.push_back(f.grad[i].ptr);
      }
      d.out = nullptr; d.pitch_out = 0; d.OW = 0; d.OH = 0;
      if (i + 1 < L) {
        if (!img_ok(&f.img[i + 1])) { rc = fail(DFX_E_INVALID, "frame %d: level %d image null or empty", k, i + 1); break; }
        d.out = (float*)f.img[i + 1].ptr; d.pitch_out = (uint32_t)f.img[i + 1].pitch_bytes; d.OW = (int)f.img[i + 1].w; d.OH = (int)f.img[i + 1].h;
        written.push_back(f.img[i + 1].ptr);
      }
      // the row-streaming kernel (k_pyr_rows): 8-byte image loads, 16-byte gradient stores, 32-bit row offsets
      if ((((uintptr_t)d.in | d.pitch_in) & 7) || (d.grad && (((uintptr_t)d.grad | d.pitch_grad) & 15)) || (d.out && (((uintptr_t)d.out | d.pitch_out) & 3)) ||
          (uint64_t)d.pitch_in * H >= (1ull << 31) || (uint64_t)d.pitch_grad * H >= (1ull << 31))
        rows_ok[i] = false;
    }
    if (rc) g_last_error = "frame " + std::to_string(k) + ": " + g_last_error;
  }
  if (!rc) rc = img_note_writes(c, written);
  if (rc) return rc;
  // The same buffers as the previous build of this context (a camera's live frame, a ring of frames: UploadLiveFrame, deepfactors.cpp:616-630, fills the same
  // pyramids frame after frame): the descriptors are still in device memory -- every launch reads them there, nothing is staged.
  const bool cached = DFX_PYR_DESC_CACHE && L > 1 && c->pyr_dev && c->pyr_mirror_valid && c->pyr_last == c->pyr_build;
  int slot = -1;
  char* host = nullptr;
  if (!cached) {
    c->pyr_mirror_valid = false;
    if ((rc = stage_acquire(c, dbytes, &slot, &host))) return rc;
    std::memcpy(host, c->pyr_build.data(), dbytes);
  }
  // Descriptors: no copy command in front of the build.  The FIRST launch reads the pinned staging slot itself (zero-copy) and its workgroup 0 mirrors the
  // descriptors of its own and all later levels into device memory, where the launches behind it read them (every one of the ~10^4 workgroups of a level
  // starts with its descriptor: levels 1-3 of a 64-frame build out of host memory measured 17.8 / 10.0 / 8.6 us against 15.8 / 8.4 / 6.3 out of device
  // memory, level 0 the same either way).  Rounds 5-6 uploaded them with hipMemcpyAsync on the stream: a 4.4 us blit kernel and 6.3 us of idle GPU in front of
  // it per 64-frame build (rocprofv3 timeline, profiles/r06_pyramid.txt) -- 10.7 of 72.4 us.
  // (Measured and dropped in round 6: issuing level 0 twice -- blur-down only in front of the level 1.. chain, gradient only on a second stream beside it -- to hide
  // the chain's launch latencies behind the 157 MB of gradient stores.  The blur-only kernel still takes 46 us of the combined kernel's 52 (a wave's walk is bound
  // by load latency per row, not by its bytes), and the cross-stream hand-over costs more than the chain: 88 -> 111 us per 64-frame build; profiles/r06_pyramid.txt.)
  void* hdev = nullptr;
  if (!cached) {
    const hipError_t ge = hipHostGetDevicePointer(&hdev, host, 0);
    if (ge != hipSuccess) { (void)stage_release(c, slot); return fail(DFX_E_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(ge)); }
  }
  if (L > 1 && !cached) {
    if (c->pyr_bytes < dbytes) (void)hipStreamSynchronize(c->stream);
    if ((rc = grow_dev((void**)&c->pyr_dev, &c->pyr_bytes, dbytes, c->stream))) { (void)stage_release(c, slot); return rc; }
  }
  const dfx::PyrLevelDev* hostdev = reinterpret_cast<const dfx::PyrLevelDev*>(hdev);
  dfx::PyrLevelDev* mirror = L > 1 ? reinterpret_cast<dfx::PyrLevelDev*>(c->pyr_dev) : nullptr;
  bool mirrored = cached;  // a launch has left the descriptors in device memory
  if ((rc = ensure_done_flag(c))) { if (slot >= 0) (void)stage_release(c, slot); return rc; }
  dfx::PyrStart start;     // the build's first launch reports that it is running (wait_pyr_started): the staging slot needs no event behind the build
  start.word = c->done_flag_dev + kPyrStartWord;
  if (!cached) {
    start.seq = ++c->pyr_seq;
    if (start.seq == 0) start.seq = ++c->pyr_seq;   // (0 = "slot not guarded by a build" in stage_pyr)
  }
  bool signalled = cached;   // (no slot to guard)
  // The small levels as ONE launch (k_pyr_tail: a few bands per frame, each workgroup with its rows of those levels in LDS): from the first level k0 >= 1 whose
  // image is at most kPyrTailMaxPixels (160 x 120 of a 640 x 480 build), when that is at least two levels: 6.0 + 4.4 us of launches and a boundary become one
  int k0 = L, tail_nb = 0, tail_rp = 0;
  size_t tail_lds = 0;
  for (int i = 1; i + 1 < L; ++i)
    if ((size_t)frames[0].img[i].w * frames[0].img[i].h <= kPyrTailMaxPixels) { k0 = i; break; }
  if (k0 < L) {
    int Hs[DFX_MAX_PYR_LEVELS], Ws[DFX_MAX_PYR_LEVELS];
    for (int i = k0; i < L; ++i) { Hs[i - k0] = (int)frames[0].img[i].h; Ws[i - k0] = (int)frames[0].img[i].w; }
    tail_nb = dfx::pyr_tail_plan(Hs, Ws, L - k0, n, &tail_rp, &tail_lds);
    if (tail_nb == 0) k0 = L;
  }
  for (int i = 0; i < k0; ++i) {
    bool any = false;   // (the last level of a batch whose frames all skip its gradient has nothing to do)
    for (int k = 0; k < n; ++k) any = any || hd[(size_t)i * n + k].grad || hd[(size_t)i * n + k].out;
    if (!any) continue;
    const bool first = !mirrored && mirror;
    hipError_t e = dfx::launch_pyr_level((mirrored ? mirror : hostdev) + (size_t)i * n, n, (int)frames[0].img[i].w, (int)frames[0].img[i].h, c->stream, rows_ok[i],
                                         first ? mirror + (size_t)i * n : nullptr, first ? (L - i) * n : 0, signalled ? dfx::PyrStart{} : start, c->cu_count);
    if (e != hipSuccess) { c->pyr_mirror_valid = false; if (slot >= 0) (void)stage_release(c, slot); return fail(DFX_E_HIP, "k_pyr_level launch failed: %s", hipGetErrorString(e)); }
    mirrored = mirrored || first;
    signalled = true;
  }
  if (k0 < L) {
    const hipError_t e = dfx::launch_pyr_tail(mirrored ? mirror : hostdev, n, k0, L, tail_nb, tail_rp, tail_lds, c->stream);   // (not mirrored: level 0 had nothing to do)
    if (e != hipSuccess) { c->pyr_mirror_valid = false; if (slot >= 0) (void)stage_release(c, slot); return fail(DFX_E_HIP, "k_pyr_tail launch failed: %s", hipGetErrorString(e)); }
  }
  if (cached) return DFX_OK;
  if (mirrored && mirror) { c->pyr_last.swap(c->pyr_build); c->pyr_mirror_valid = true; }   // (the mirror holds ALL levels only when the first launch was level 0's)
  if (!signalled) return stage_release(c, slot);   // (nothing launched that reports: the event)
  c->stage_used[slot] = true;
  c->stage_pyr[slot] = start.seq;
  return DFX_OK;
}

DFX_API int dfx_debug_pyramid_launch_shape(int w, int h, int n, int cus, int* rows_per_segment, int* workgroups, int* waves_per_workgroup) {
  if (w < 2 || (w & 1) || h < 1 || n < 1 || cus < 0) return fail(DFX_E_INVALID, "dfx_debug_pyramid_launch_shape: w even >= 2, h >= 1, n >= 1, cus >= 0");
  int wpg, gps, R;
  dfx::pyr_rows_shape(w, h, n, cus, &wpg, &gps, &R);
  if (rows_per_segment) *rows_per_segment = R;
  if (workgroups) *workgroups = gps * ((h + R - 1) / R) * n;
  if (waves_per_workgroup) *waves_per_workgroup = wpg;
  return DFX_OK;
}

DFX_API int dfx_build_pyramid(dfx_ctx* c, const dfx_pyramid* frame) {
  int rc;
  if ((rc = dfx_build_pyramid_batch_async(c, frame, 1))) return rc;
  if ((rc = wait_stream(c))) return rc;   // like the reference's per-level calls (CudaCheckLastError = cudaDeviceSynchronize, cu_image_proc.cpp:111,185)
  return DFX_OK;
}

DFX_API int dfx_squared_error(dfx_ctx* c, const dfx_img* a, const dfx_img* b, float* out) {
  if (!c || !out) return fail(DFX_E_INVALID, "null argument");
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (!img_ok(a)) return fail(DFX_E_INVALID, "a: null or empty image view");
  if ((rc = check_img(a, "a", a->w, a->h, 4))) return rc;
  if ((rc = check_img(b, "b", a->w, a->h, 4))) return rc;
  const int blocks = simple_blocks(a->w, a->h);
  const size_t pbytes = (size_t)blocks * dfx::kSimpleRow * sizeof(float);
  if ((rc = grow_partials(c, pbytes))) return rc;
  void* tgt;
  if ((rc = result_target(c, sizeof(float), &tgt))) return rc;
  dfx::DoneFlag done;
  if ((rc = new_done_flag(c, &done))) return rc;
  DFX_HIP(dfx::launch_squared_error((const float*)a->ptr, (uint32_t)a->pitch_bytes, (const float*)b->ptr, (uint32_t)b->pitch_bytes,
                                    (int)a->w, (int)a->h, blocks, c->partials, (float*)tgt, c->stream, done));
  return finish_result_polled(c, out, sizeof(float), done);
}

DFX_API int dfx_depth_aligner_step(dfx_ctx* c, int cs, const float* code, const dfx_img* target_dpt, const dfx_img* prx_orig,
                                   const dfx_img* prx_jac, float avg_dpt, void* out_item) {
  if (!c || !code || !out_item) return fail(DFX_E_INVALID, "dfx_depth_aligner_step: null argument");
  if (!cs_supported(cs)) return fail(DFX_E_INVALID, "unsupported code size %d (16, 32, 64)", cs);
  int rc;
  if ((rc = ensure_device(c))) return rc;
  if (!img_ok(prx_orig)) return fail(DFX_E_INVALID, "prx_orig: null or empty image view");
  const uint32_t W = prx_orig->w, H = prx_orig->h;
  if ((rc = check_img(prx_orig, "prx_orig", W, H, 4))) return rc;
  if ((rc = check_img(target_dpt, "target_dpt", W, H, 4))) return rc;
  if ((rc = check_img(prx_jac, "prx_jac", W * (uint32_t)cs, H, 4))) return rc;
  if (((uintptr_t)prx_jac->ptr | prx_jac->pitch_bytes) & 15) return fail(DFX_E_INVALID, "prx_jac: pointer/pitch must be 16-byte aligned");
  // current depth into scratch (the kernel body of the reference recomputes DepthFromCode per pixel)
  const size_t dbytes = (size_t)W * H * sizeof(float);
  if (c->depth_scratch_bytes < dbytes) DFX_HIP(hipStreamSynchronize(c->stream));
  if ((rc = grow_dev((void**)&c->depth_scratch, &c->depth_scratch_bytes, dbytes, c->stream))) return rc;
  float* code_dev;
  if ((rc = upload_code(c, cs, code, &code_dev))) return rc;
  DFX_HIP(dfx::launch_update_depth(cs, code_dev, (const float*)prx_orig->ptr, (uint32_t)prx_orig->pitch_bytes,
                                   (const float*)prx_jac->ptr, (uint32_t)prx_jac->pitch_bytes, avg_dpt, c->depth_scratch, W * 4,
                                   (int)W, (int)H, c->stream));
  // pseudo-pair descriptor: travels in the kernel arguments
  dfx::SfmPairDev hd_;
  dfx::SfmPairDev* hd = &hd_;
  std::memset(hd, 0, sizeof(*hd));
  hd->fx = hd->fy = 1.f;
  hd->img0 = (const float*)target_dpt->ptr; hd->pitch_img0 = (uint32_t)target_dpt->pitch_bytes;
  hd->dpt0 = c->depth_scratch; hd->pitch_dpt0 = W * 4;
  hd->jac = (const float*)prx_jac->ptr; hd->pitch_jac = (uint32_t)prx_jac->pitch_bytes;
  const int bpp = auto_step_blocks(c, W, H, 1, cs);
  const size_t pbytes = dfx::sfm_step_partials_bytes(cs, 1, bpp);
  if ((rc = grow_partials(c, pbytes))) return rc;
  const size_t ibytes = dfx_item_size(cs);
  void* tgt;
  if ((rc = result_target(c, ibytes, &tgt))) return rc;
  dfx::DoneFlag done;
  if ((rc = new_done_flag(c, &done))) return rc;
  if ((rc = ensure_fin_scratch(c))) return rc;
  DFX_HIP(dfx::launch_depth_aligner_step(cs, hd, (int)W, (int)H, avg_dpt, bpp, c->partials, tgt, c->stream,
                                         prx_jac->pitch_bytes == (size_t)W * cs * 4, resolve_mfma(c, cs), done.flag ? &done : nullptr, c->fin_scratch, c->fin_cnt));
  return finish_result_polled(c, out_item, ibytes, done);
}

}  // extern "C"
