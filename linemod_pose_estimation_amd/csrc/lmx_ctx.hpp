// The device context of liblmx.so (struct lmx_ctx) and the helpers its translation units share.  Nothing here crosses the C ABI
// (include/lmx.h).  The host side of the library is split along the context's life:
//   lmx_bank.cpp     host bank: create / add_class / accessors / NORMAL_LUT, lmx_last_error
//   lmx_ctx.cpp      context create / destroy, geometry and the device-resident bank, uploads (frames, masks, raw camera frames),
//                    the hooks device groups use (lmx_group.cpp)
//   lmx_enqueue.cpp  the per-batch kernel chain (plain, small-batch, hipGraph), lmx_ctx_enqueue, lmx_match / lmx_match_batch
//   lmx_collect.cpp  read-back and finalisation (std::sort + std::unique), the device-side consumer chain, gather-block export / merge
//   lmx_cluster.cpp  the reference's voting / cluster / NMS chain on the host
//   lmx_cache.cpp    bank fingerprint, binary bank files, the per-request caches (lmx_bank_load_yaml_cached, lmx_ctx_acquire)
//   lmx_debug.cpp    introspection, per-kernel timing, test hooks
#pragma once

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "lmx_internal.hpp"

#define LMX_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? LMX_ERR_NO_DEVICE : LMX_ERR_HIP; \
    }                                                                                          \
  } while (0)

namespace lmx {

struct ModalityBuffers {
  uint8_t* bgr[kMaxLevels] = {nullptr, nullptr, nullptr, nullptr};  // ColorGradient: colour source pyramid
  uint16_t* depth = nullptr;                                          // DepthNormal: level-0 depth (mm)
};

struct ProfEvent { int kernel; hipEvent_t start, stop; };

}  // namespace lmx

using namespace lmx;   // internal header: every includer is a translation unit of the library's host side

struct lmx_ctx {
  std::recursive_mutex call_mutex;   // lmx_ctx_lock / lmx_ctx_unlock; taken by the synchronous composites
  const lmx_bank* bank = nullptr;
  lmx_ctx_desc desc{};
  int device = 0;
  hipStream_t stream = nullptr;  // lane 0's stream: the caller's (desc.stream) or a private one; uploads run here
  bool own_stream = false;
  // LMX_CTX_OVERLAP: further lanes = private streams + their own intermediate buffers (quantised images, memories, colour
  // pyramid levels >= 1, candidates).  Output slot k runs on lane k % n_lanes and up to two enqueues per lane may be
  // outstanding, so each stream always has the next batch queued behind the running one and the other lanes' kernels fill the
  // tail of every kernel (the last, partially filled wave of workgroups).  kp.fb / mb[].bgr[l>=1] / d_cands always hold the view of the lane
  // of the most recent enqueue.
#ifndef LMX_LANES
#define LMX_LANES 3
#endif
  static constexpr int kLanes = LMX_LANES;  // measured at 64 frames per batch: 1 lane 118 k, 2: 134.7 k, 3: 138.9 k, 4: 137.2 k frames/s (round 2); round 4's kernels: 2: 154.2 k, 3: 154.3 k, 4: 152.9 k, 5: 143.5 k
  int n_lanes = 1;
  hipStream_t lane_stream[kLanes] = {};
  // Host-frame boundary (the reference hands match() host images every call): uploads rotate over `n_sets` frame sets, each
  // with its own device frames and pinned staging, and run on a private copy stream.  An upload waits (on the device) only for
  // the enqueues that still read the set it overwrites and (on the host) for the previous transfer out of that set's staging;
  // an enqueue reads the most recently uploaded set behind its `h2d_done` event.  So the transfer of batch i+1 overlaps the
  // kernels of batch i, and nothing synchronises the host with the lanes.
  static constexpr int kSets = kLanes + 1;
  struct FrameSet {
    uint8_t* bgr[kMaxModalities] = {};      // level-0 colour frames [F][H][W][3]
    uint16_t* depth[kMaxModalities] = {};   // level-0 depth frames [F][H][W]
    uint8_t* h_stage = nullptr;             // pinned staging for pageable sources
    PullEntry* h_tab = nullptr;             // pinned [M][F] table of caller-owned pinned images (k_pull_frames), and its device view
    PullEntry* d_tab = nullptr;
    hipEvent_t h2d_done = nullptr;          // recorded on the copy stream behind the set's most recent upload
    bool h2d_recorded = false;
    hipEvent_t read_done[kLanes] = {};      // recorded on a lane's stream behind the last kernel of an enqueue that reads the set
    bool read_recorded[kLanes] = {};
    // Small batches (<= kStoreFrames frames) from pageable memory skip staging and DMA: the host writes the frames straight into
    // these fine-grained device buffers through the PCIe BAR with non-temporal stores (scripts/microbench/bar_store.hip: 45.7 GB/s
    // from one thread, 34 us for a 640x480 RGB-D frame, against 32 us of staging + 49 us until the DMA has landed).
    uint8_t* store_buf[kMaxModalities] = {};
    bool stored = false;                    // the set's current frames live in store_buf
    uint32_t* store_flag = nullptr;         // host-visible device words next to store_buf: progress of a STREAMED store, modality m at [32 * m] (StreamWait)
    int n_uploaded = 0;                     // frames the most recent upload put into the set (an enqueue may use fewer, not more)
    // Detector::match's `masks` argument: level-0 masks [F][H][W] per modality for the set's current frames (lmx_ctx_upload_masks),
    // allocated on first use; `masked[m]` is cleared by every upload into the set
    uint8_t* mask[kMaxModalities] = {};
    bool masked[kMaxModalities] = {};
    // lmx_ctx_upload_raw: the uncropped camera frames of this set (pinned staging + device copy), grown on demand.  Per set, so that the
    // staging of batch i + 1 overlaps the transfer and the pre-processing kernels of batch i like lmx_ctx_upload's does
    uint8_t* h_raw = nullptr;
    uint8_t* d_raw = nullptr;
    size_t raw_bytes = 0;
  };
  uint8_t* h_mask_stage = nullptr;          // pinned [F][H][W], one modality at a time
  hipEvent_t mask_h2d = nullptr;
  static constexpr int kStoreFrames = 2;
  bool store_ok = false;                    // large-BAR device, buffers allocated, not switched off (LMX_NO_STORE_UPLOAD)
  // lmx_match with a fresh host frame: the level-0 quantisers are launched first and wait, tile by tile, for the rows the calling thread is still
  // storing (StreamWait in lmx_internal.hpp).  LMX_NO_STREAM_STORE=1 restores "store, then launch"; LMX_STREAM_TIMEOUT_US bounds a workgroup's wait
  // (default one second: a missing store is an error reported by collect, never a hang); LMX_TEST_DROP_STREAM_STORE=1 is the test hook that leaves
  // the depth rows out.  All read once, when the context is created.
  // LMX_MATCH_TRACE=1 (read once): host-side phases of the one-frame call, averaged, printed when the context is destroyed
  bool trace_match = false;
  enum { TM_UPLOAD, TM_LAUNCH_COLOR, TM_STORE_COLOR, TM_LAUNCH_DEPTH, TM_STORE_DEPTH, TM_LAUNCH_REST, TM_WAIT, TM_FINALIZE, TM_COUNT };
  double tm_acc[TM_COUNT] = {};
  long tm_n = 0;
  bool stream_ok = false, env_test_drop_stream = false;
  uint32_t stream_seq = 0, stream_timeout_ticks = 100000000u;
  // `end` 0: the calling thread stores every band, top to bottom.  +1 / -1: two threads share the modality, this one claims bands from the top /
  // from the bottom (stream_claim[m], reset by the caller before the second thread is started) until none is left; see StreamWait::flag_hi
  void store_modality_streamed(lmx_ctx::FrameSet& fs, int m, int n_frames, const lmx_image* sources, uint32_t seq, int end = 0);
  void stream_reset_hi(lmx_ctx::FrameSet& fs, int m, int n_frames, uint32_t seq);   // "nothing stored from the bottom yet" for this call
  std::atomic<int> stream_claim[lmx::kMaxModalities];
  int stream_band_rows = 96;        // LMX_STREAM_BAND_ROWS (rows per progress update; 48 and 96 measured best, profiles/r04_single_frame_latency.txt)
  FrameSet sets[kSets];
  int n_sets = 2;
  int cur_set = 0;                  // the set the next enqueue reads (= the most recent upload)
  hipStream_t copy_stream = nullptr;
  // lmx_ctx_upload_raw: the pre-processing kernels run here, behind the raw frames' transfer, so that the copy stream can already move
  // the next batch while they work (created on first use)
  hipStream_t pre_stream = nullptr;
  hipEvent_t raw_dma_done = nullptr;
  std::unique_ptr<lmx::CopyPool> pool;
  FrameBuffers lane_fb[kLanes];
  uint8_t* lane_bgr[kLanes][kMaxModalities][kMaxLevels] = {};
  Candidate* lane_cands[kLanes] = {};
  hipStream_t cur_stream = nullptr;  // stream of the stage being issued (ScopedKernel records its events there)
  int last_slot = 0;
  int L = 0, M = 0, F = 0;
  uint32_t cap_total = 0;  // capacity of the shared candidate / match lists (max_candidates * max_batch)
  KernelParams kp{};
  ModalityBuffers mb[kMaxModalities];
  std::vector<void*> allocs;
  // device bank
  DeviceBankView dbank{};
  int n_classes = 0;
  std::vector<std::string> class_names;
  int32_t* d_class_slot = nullptr;
  std::vector<int32_t> cur_slots;
  uint8_t* d_normal_bins = nullptr;  // the bank's NORMAL_LUT as median bins (k_depth_quantize)
  // outputs
  Candidate* d_cands = nullptr;
  // Output slots (two per lane) so that enqueues can run while earlier ones are being collected on the host.
  // Slot layout (device and pinned host mirror): [64 B header: cand_count @0, match_count @4][records]; on the device the candidate
  // list's stripe counters (lmx::kStripeAreaBytes) sit in front of the header, and cand_count is written by k_refine from them.
  static constexpr size_t kFirstSlice = 2048;  // records published with the header; more are fetched on demand by collect
  static constexpr int kSlots = 2 * kLanes;   // 2 per lane; without LMX_CTX_OVERLAP only the first two are used
  int n_slots = 2;
  uint8_t* d_out_slot[kSlots] = {};
  uint8_t* h_out_slot[kSlots] = {};   // pinned host mirrors
  uint8_t* h_out_dev[kSlots] = {};    // their device-side addresses (hipHostGetDevicePointer)
  hipEvent_t done[kSlots] = {};
  int slot_frames[kSlots] = {};
  int head = 0;         // slot the next enqueue writes
  int outstanding = 0;  // enqueued and not yet collected (<= kSlots)
  uint32_t* d_pub_counter = nullptr;   // [kSlots] ticket counters of k_refine's folded read-back (zero between batches)
  // folded read-back: the batch's sequence number (never 0) lands in word 7 of the slot's pinned header behind the records; collect() polls it for
  // a bounded time before it falls back to the slot's event.  0 = the slot's enqueue did not fold its read-back
  uint32_t pub_seq[kSlots] = {};
  uint32_t pub_seq_counter = 0;
  std::unique_ptr<lmx::LaunchHelper> launch_helper;   // queues the rest of a one-frame call's chain (and stores the depth frame) while the caller stores the colour frame
  // Small batches through lmx_match / lmx_match_batch: the direct stores of the frames are deferred to the enqueue, which interleaves
  // them with the launches (colour frames -> colour kernels -> depth frames while those run -> the rest): see issue_small
  const lmx_image* deferred_sources = nullptr;
  int deferred_frames = 0;
  uint8_t* d_out = nullptr;  // slot of the most recent enqueue
  uint8_t* h_out = nullptr;  // slot being collected
  size_t h_out_records = 0;
  size_t h_stage_bytes = 0;    // per frame set
  size_t frame_bytes[kMaxModalities] = {0, 0, 0, 0};
  // hipGraph cache (LMX_CTX_HIPGRAPH)
  struct GraphEntry { int slot; int set; int n_frames; uint32_t threshold_bits; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  // device form of finalise + cluster (lmx_ctx_collect_clusters): side-car and output buffers, allocated on first use
  double* d_f2_dists = nullptr;
  int32_t* d_f2_rects = nullptr;
  size_t f2_templates = 0;
  lmx_cluster_params f2_params{};
  bool f2_sidecar = false;
  std::vector<double> f2_host_dists;     // host copies for the fallback path
  std::vector<int32_t> f2_host_rects;
  // outputs of k_f2_finalize_cluster live in PINNED host memory (the kernel writes them through the mapping): one stream sync, no
  // device-to-host copies (round 3: the four copies cost three times the kernel)
  uint8_t* h_f2_out = nullptr;           // [F][F2_MAX] matches | [F][4] counts | [F][F2_MAX] clusters | [F][F2_MAX] members
  lmx_match_t* d_f2_matches = nullptr;   // device views into h_f2_out
  uint32_t* d_f2_counts = nullptr;
  lmx_cluster_t* d_f2_clusters = nullptr;
  int32_t* d_f2_members = nullptr;
  uint8_t* d_f2_scratch = nullptr;
  hipStream_t f2_stream = nullptr;       // the kernel's own stream: a lane's stream may already hold later batches
  // stats / profiling
  int64_t stat_cands = 0, stat_matches = 0;
  uint32_t profiling = 0;  // bitmask over kernel ids
  std::vector<ProfEvent> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  double k_ms[K_COUNT] = {0};
  int64_t k_launches[K_COUNT] = {0};
  float last_threshold = 0.f;
  // LMX_COLLECT_TRACE=1 (read once, at context creation): collect() prints its host-side split to stderr -- wait for the slot, fetch of
  // the records beyond the first slice, grouping by frame, restore insertion order + std::sort + std::unique
  bool trace_collect = false;
  // the rest of the LMX_* environment a context consults, read ONCE when it is created (a per-upload or per-launch getenv is a libc
  // lock and a string scan on the hot path): LMX_PINNED_MODE (0 pull kernel, 1 per-image DMA, 2 stage; -1 = by flags),
  // LMX_NO_SMALL_CHAIN, LMX_DEBUG_COLLECT, LMX_UPLOAD_THREADS
  int env_pinned_mode = -1;
  bool env_no_small_chain = false, env_debug_collect = false, env_no_header_poll = false, env_no_launch_thread = false, env_one_store_thread = false, env_no_delegate_first = false;   // LMX_NO_HEADER_POLL, LMX_NO_LAUNCH_THREAD: A/B switches
  int cand_stripes = 0;   // stripes of the candidate list in use; 0 = by batch size (stripes_for), LMX_CAND_STRIPES = 1, 2, 4, ... 64 fixes it (A/B switch, read once)
  // One or two frames per call: few candidates, and every workgroup of k_refine starts by reading all stripe counters -- 64 lines cost the
  // call 2 us, 8 cost nothing measurable (profiles/r03_single_frame_stripes.txt); batches: 64, where the appends would otherwise queue
  int stripes_for(int n_frames) const { return cand_stripes ? cand_stripes : (n_frames <= kStoreFrames ? 8 : lmx::kCandStripes); }
  int env_upload_threads = 0;

  uint32_t* d_cand_count() { return reinterpret_cast<uint32_t*>(d_out); }
  uint32_t* d_match_count() { return reinterpret_cast<uint32_t*>(d_out + 4); }
  lmx_raw_match_t* d_records() { return reinterpret_cast<lmx_raw_match_t*>(d_out + 64); }
};

namespace lmx {

template <typename T>
inline lmx_status dev_alloc(lmx_ctx* c, T** p, size_t count, bool zero) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 256);
  LMX_HIP(hipMalloc(&q, bytes));
  c->allocs.push_back(q);
  if (zero) LMX_HIP(hipMemsetAsync(q, 0, bytes, c->stream));
  *p = reinterpret_cast<T*>(q);
  return LMX_OK;
}

template <typename T>
inline lmx_status dev_upload(lmx_ctx* c, const T** p, const std::vector<T>& v) {
  T* q = nullptr;
  lmx_status st = dev_alloc(c, &q, std::max<size_t>(v.size(), 1), false);
  if (st != LMX_OK) return st;
  if (!v.empty()) LMX_HIP(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return LMX_OK;
}

// lmx_ctx.cpp
bool get_events(lmx_ctx* c, hipEvent_t* a, hipEvent_t* b);
void drain_profiling(lmx_ctx* c);
void select_lane(lmx_ctx* c, int lane);   // points the context's working view (kp.fb, derived colour pyramid levels, candidate list) at one lane's buffers
void select_set(lmx_ctx* c, int set);     // points the level-0 frame pointers at one frame set
lmx_status sync_lanes(lmx_ctx* c);        // host-side wait for everything queued on every lane and on the copy stream
int upload_threads(const lmx_ctx* c);
void store_modality(lmx_ctx* c, lmx_ctx::FrameSet& fs, int m, int n_frames, const lmx_image* sources);
// lmx_cluster.cpp
lmx_status check_vote_rings(const double* dists, size_t n, const lmx_cluster_params* pp);

struct ScopedKernel {
  lmx_ctx* c; int id; hipEvent_t a{}, b{}; bool on = false;
  ScopedKernel(lmx_ctx* c_, int id_) : c(c_), id(id_) {
    if (((c->profiling >> id) & 1u) && get_events(c, &a, &b)) { on = true; (void)hipEventRecord(a, c->cur_stream); }
  }
  ~ScopedKernel() {
    if (on) { (void)hipEventRecord(b, c->cur_stream); c->pending.push_back({id, a, b}); }
  }
};

}  // namespace lmx
