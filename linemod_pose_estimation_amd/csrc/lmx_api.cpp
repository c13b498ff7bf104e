// Host side of liblmx.so: the C ABI of include/lmx.h over the HIP kernels in lmx_kernels.hip.
// Mirrors the call surface of cv::linemod::Detector as the reference uses it
// (/root/reference/src/rgbdDetector.cpp:31-34, :1668-1680); see include/lmx.h for the per-function mapping.
// There is no CPU compute path in this library: without a usable HIP device every compute call fails.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <cctype>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "lmx_internal.hpp"
#include "lmx_sort_emul.hpp"

namespace lmx {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

#define LMX_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? LMX_ERR_NO_DEVICE : LMX_ERR_HIP; \
    }                                                                                          \
  } while (0)

static const char* kKernelNames[K_COUNT] = {"k_pre",            "k_color_quantize", "k_depth_quantize", "k_nn_down2",   "k_spread_linearize",
                                            "k_pack_nibbles",   "k_score_coarse",   "k_refine"};

// upstream Match ordering; class identity is the class index
struct HostMatch {
  lmx_match_t m;
  bool operator<(const HostMatch& r) const {
    if (m.similarity != r.m.similarity) return m.similarity > r.m.similarity;
    return m.template_id < r.m.template_id;
  }
  bool operator==(const HostMatch& r) const {
    return m.x == r.m.x && m.y == r.m.y && m.similarity == r.m.similarity && m.class_index == r.m.class_index;
  }
};

// records of ONE frame -> upstream output order.  Insertion order is restored from order_key, then the very
// same std::sort / std::unique upstream applies (libstdc++'s tie order is part of the observable result).
static void finalize_frame(std::vector<const lmx_raw_match_t*>& recs, std::vector<HostMatch>& out) {
  // back into upstream's insertion order (the keys are distinct: one record per class, template and coarse position).  Long lists sort
  // (key, pointer) pairs instead of dereferencing a pointer per comparison: the records lie in pinned memory in arrival order
  if (recs.size() > 4096) {
    std::vector<std::pair<uint64_t, const lmx_raw_match_t*>> keyed(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) keyed[i] = {recs[i]->order_key, recs[i]};
    std::sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, const lmx_raw_match_t*>& a, const std::pair<uint64_t, const lmx_raw_match_t*>& b) { return a.first < b.first; });
    for (size_t i = 0; i < recs.size(); ++i) recs[i] = keyed[i].second;
  } else {
    std::sort(recs.begin(), recs.end(), [](const lmx_raw_match_t* a, const lmx_raw_match_t* b) { return a->order_key < b->order_key; });
  }
  out.clear();
  out.reserve(recs.size());
  for (const lmx_raw_match_t* r : recs) {
    HostMatch h;
    h.m.x = r->x; h.m.y = r->y; h.m.similarity = r->similarity; h.m.template_id = r->template_id; h.m.class_index = r->class_index;
    out.push_back(h);
  }
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
}

struct ModalityBuffers {
  uint8_t* bgr[kMaxLevels] = {nullptr, nullptr, nullptr, nullptr};  // ColorGradient: colour source pyramid
  uint16_t* depth = nullptr;                                          // DepthNormal: level-0 depth (mm)
};

struct ProfEvent { int kernel; hipEvent_t start, stop; };

}  // namespace lmx

using namespace lmx;

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
  static constexpr int kLanes = LMX_LANES;  // measured at 64 frames per batch: 1 lane 118 k, 2: 134.7 k, 3: 138.9 k, 4: 137.2 k frames/s
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
  bool env_no_small_chain = false, env_debug_collect = false;
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
static lmx_status dev_alloc(lmx_ctx* c, T** p, size_t count, bool zero) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 256);
  LMX_HIP(hipMalloc(&q, bytes));
  c->allocs.push_back(q);
  if (zero) LMX_HIP(hipMemsetAsync(q, 0, bytes, c->stream));
  *p = reinterpret_cast<T*>(q);
  return LMX_OK;
}

template <typename T>
static lmx_status dev_upload(lmx_ctx* c, const T** p, const std::vector<T>& v) {
  T* q = nullptr;
  lmx_status st = dev_alloc(c, &q, std::max<size_t>(v.size(), 1), false);
  if (st != LMX_OK) return st;
  if (!v.empty()) LMX_HIP(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return LMX_OK;
}

static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

// Default NORMAL_LUT (restatement-defined, DESIGN.md: upstream's normal_lut.i is not in the reference repository): the label is
// the azimuth octant of the image-plane projection (nx, ny) of the normal, taken at the cell centre (2*v1 - 19, 2*v2 - 19); nz
// (v3) does not enter.  a = |cx|, b = |cy|: 2ab < a^2 - b^2 -> octant 0 (cx > 0) or 4; 2ab < b^2 - a^2 -> 2 (cy > 0) or 6; else
// the diagonal 1 / 3 / 5 / 7 by the signs.  Exact integer rule without ties (cx, cy odd).
void default_normal_lut(uint8_t* out) {
  for (int v3 = 0; v3 < 20; ++v3)
    for (int v2 = 0; v2 < 20; ++v2)
      for (int v1 = 0; v1 < 20; ++v1) {
        const int cx = 2 * v1 - 19, cy = 2 * v2 - 19;
        const int a = cx < 0 ? -cx : cx, b = cy < 0 ? -cy : cy;
        int k;
        if (2 * a * b < a * a - b * b) k = cx > 0 ? 0 : 4;
        else if (2 * a * b < b * b - a * a) k = cy > 0 ? 2 : 6;
        else if (cx > 0) k = cy > 0 ? 1 : 7;
        else k = cy > 0 ? 3 : 5;
        out[(v3 * 20 + v2) * 20 + v1] = (uint8_t)(1u << k);
      }
}

bool normal_lut_to_bins(const uint8_t* lut, uint8_t* bins) {
  for (int i = 0; i < LMX_NORMAL_LUT_SIZE; ++i) {
    const uint8_t v = lut[i];
    if (v & (v - 1)) return false;  // more than one bit set
    bins[i] = v ? (uint8_t)(__builtin_ctz(v) + 1) : 0;
  }
  return true;
}

// 8000 raw bytes, or text with 8000 integers separated by anything that is not a digit (C initialiser syntax of OpenCV's
// normal_lut.i: braces, commas, comments are skipped)
lmx_status normal_lut_from_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { set_error("cannot open normal LUT '%s'", path); return LMX_ERR_IO; }
  std::vector<uint8_t> buf;
  uint8_t tmp[1 << 14];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  std::vector<uint8_t> bins(LMX_NORMAL_LUT_SIZE);
  if (buf.size() == LMX_NORMAL_LUT_SIZE && normal_lut_to_bins(buf.data(), bins.data())) { out = buf; return LMX_OK; }
  std::vector<uint8_t> vals;
  for (size_t i = 0; i < buf.size();) {
    const uint8_t ch = buf[i];
    if (ch == '/' && i + 1 < buf.size() && buf[i + 1] == '/') { while (i < buf.size() && buf[i] != '\n') ++i; continue; }
    if (ch == '/' && i + 1 < buf.size() && buf[i + 1] == '*') {
      i += 2;
      while (i + 1 < buf.size() && !(buf[i] == '*' && buf[i + 1] == '/')) ++i;
      i += 2;
      continue;
    }
    if (ch >= '0' && ch <= '9') {
      if (i > 0 && (std::isalpha(buf[i - 1]) || buf[i - 1] == '_' || buf[i - 1] == '[')) {  // part of an identifier or of a dimension like [20]
        while (i < buf.size() && (std::isalnum(buf[i]) || buf[i] == '_')) ++i;
        continue;
      }
      unsigned long v = 0;
      int base = 10;
      if (ch == '0' && i + 1 < buf.size() && (buf[i + 1] == 'x' || buf[i + 1] == 'X')) { base = 16; i += 2; }
      while (i < buf.size() && std::isxdigit(buf[i]) && (base == 16 || std::isdigit(buf[i]))) {
        v = v * base + (unsigned long)(std::isdigit(buf[i]) ? buf[i] - '0' : (std::tolower(buf[i]) - 'a' + 10));
        ++i;
      }
      if (v > 255) { set_error("normal LUT '%s': value %lu does not fit a byte", path, v); return LMX_ERR_PARSE; }
      vals.push_back((uint8_t)v);
      continue;
    }
    ++i;
  }
  if (vals.size() != LMX_NORMAL_LUT_SIZE) { set_error("normal LUT '%s': %zu values, expected %d (20 x 20 x 20)", path, vals.size(), LMX_NORMAL_LUT_SIZE); return LMX_ERR_PARSE; }
  if (!normal_lut_to_bins(vals.data(), bins.data())) { set_error("normal LUT '%s': entries must be 0 or a single bit (1, 2, 4, ..., 128)", path); return LMX_ERR_PARSE; }
  out = vals;
  return LMX_OK;
}

// True when every feature of pyramid level l packs into the banded table entry (label:3 | matrix row:17 | column:12).
static bool level_features_pack(const lmx_bank* b, int l, int L, int M, int T, int Hc) {
  const int per = L * M;
  for (const auto& kv : b->classes) {
    const ClassData& cd = kv.second;
    for (long t = 0; t < cd.n_pyramids; ++t)
      for (int m = 0; m < M; ++m) {
        const int32_t* tm = &cd.templates[((size_t)t * per + (size_t)l * M + m) * 5];
        for (int f = 0; f < tm[4]; ++f) {
          const int32_t* ft = &cd.features[((size_t)tm[3] + f) * 3];
          if (ft[0] < 0 || ft[1] < 0 || ft[0] / T >= 4096) return false;
          if ((long)((ft[1] % T) * T + (ft[0] % T)) * Hc + ft[1] / T >= (1L << 17)) return false;
        }
      }
  }
  return true;
}

static lmx_status build_geometry(lmx_ctx* c) {
  int W = c->desc.width, H = c->desc.height;
  for (int l = 0; l < c->L; ++l) {
    if (l > 0) { W /= 2; H /= 2; }
    const int T = c->bank->T[l];
    if (T < 1 || T > 16) { set_error("T=%d at level %d unsupported (1..16)", T, l); return LMX_ERR_INVALID_ARG; }
    if (W <= 0 || H <= 0 || W % T != 0 || H % T != 0) {
      set_error("image size %dx%d at pyramid level %d is not a multiple of T=%d (upstream linearize CV_Assert)", W, H, l, T);
      return LMX_ERR_SHAPE;
    }
    if (l + 1 < c->L && (W < 4 || H < 4)) {  // the fused pyrDown reflects at most two pixels across a border
      set_error("image size %dx%d at pyramid level %d is too small to be downsampled again", W, H, l);
      return LMX_ERR_SHAPE;
    }
    if (((long)W * H) % 16 != 0) {
      set_error("rows*cols = %ld at level %d is not a multiple of 16 (upstream computeResponseMaps CV_Assert)", (long)W * H, l);
      return LMX_ERR_SHAPE;
    }
    LevelGeom& g = c->kp.geom[l];
    g.W = W; g.H = H; g.T = T; g.Wc = W / T; g.Hc = H / T;
    g.cells = (uint32_t)g.Wc * g.Hc;
    const uint32_t pad = g.cells + std::max<uint32_t>(16u * g.Wc + 64u, 2048u);
    g.ori_stride = round_up((uint32_t)T * T * g.cells + pad, 256);
    g.mod_stride = 8 * g.ori_stride + 8192;
    g.zero_off = (uint32_t)T * T * g.cells;
    const uint32_t nib_bytes = ((uint32_t)T * T * g.cells + 1) / 2;
    g.nib_ori_stride = round_up(nib_bytes + g.cells / 2 + 2048 + 64, 256);
    g.nib_mod_stride = 8 * g.nib_ori_stride + 8192;
    g.nib_zero_off = round_up(nib_bytes + 32, 4);
    g.ls_zero_off = (uint32_t)T * T * g.cells;
    g.ls_stride = round_up(g.ls_zero_off + pad, 256);
    g.ls_bands = 0; g.ls_band_stride = 0;
    const bool flat_only = getenv("LMX_LS_FLAT") != nullptr;    // A/B switch (scripts/ls_ab.sh, tests)
    if (l < c->L - 1 && g.Wc % 16 == 0 && g.Wc >= 32 && !flat_only) {
      const uint32_t rows = (uint32_t)T * T * g.Hc;
      if (rows < (1u << 17) && g.Wc < 4096 && level_features_pack(c->bank, l, c->L, c->M, T, g.Hc)) {
        g.ls_bands = (uint32_t)g.Wc / 16;
        g.ls_band_stride = (rows + 17) * 32;
        g.ls_zero_off = (rows + 1) * 32;          // band 0, the 16 never-written rows behind the image
        g.ls_stride = round_up(g.ls_bands * g.ls_band_stride, 256);
      }
    }
  }
  return LMX_OK;
}

static lmx_status build_device_bank(lmx_ctx* c) {
  const lmx_bank* b = c->bank;
  const int L = c->L, M = c->M, per = L * M;
  const int world = std::max(1, c->desc.shard_world), rank = c->desc.shard_rank;
  std::vector<TemplateInfo> info;
  std::vector<TemplateLevelInfo> linfo;
  std::vector<uint32_t> coarse, uni, blk;
  std::vector<ScoreInfo> sinfo;
  uint32_t pending_groups = 0;
  bool uni_ok = true;
  const uint32_t uni_block = (uint32_t)c->F * c->kp.geom[L - 1].nib_mod_stride;
  std::vector<std::vector<FeatEntry>> feat_l(L);
  std::vector<std::vector<uint8_t>> cnt_l(L);
  int ci = 0, nf_max = 0;
  c->class_names.clear();
  for (const auto& kv : b->classes) {
    const ClassData& cd = kv.second;
    c->class_names.push_back(kv.first);
    const long n = cd.n_pyramids;
    const int begin = (int)((rank * n) / world), end = (int)(((rank + 1) * n) / world);
    for (int t = begin; t < end; ++t) {
      TemplateInfo ti;
      ti.class_index = ci; ti.template_id = t; ti.class_slot = 0; ti.pad = 0;
      info.push_back(ti);
      for (int l = 0; l < L; ++l) {
        const LevelGeom& g = c->kp.geom[l];
        TemplateLevelInfo li{};
        const int32_t* t0 = &cd.templates[((size_t)t * per + (size_t)l * M) * 5];
        li.width = t0[0]; li.height = t0[1];
        int nf_total = 0;
        for (int m = 0; m < M; ++m) {
          const int32_t* tm = &cd.templates[((size_t)t * per + (size_t)l * M + m) * 5];
          const int fb = tm[3], fc = tm[4];
          nf_total += fc;
          std::vector<FeatEntry> ent(kFeatStride);
          std::vector<uint32_t> offs(kFeatStride, (g.nib_zero_off >> 2) << 3);  // (dword index << 3) | nibble shift 0
          for (int f = 0; f < fc; ++f) {
            const int32_t* ft = &cd.features[((size_t)fb + f) * 3];
            const int x = ft[0], y = ft[1], label = ft[2];
            // accessLinearMemory: flat element index inside one orientation's [T*T][cells] matrix
            const uint32_t e0 = (uint32_t)((y % g.T) * g.T + (x % g.T)) * g.cells + (uint32_t)(y / g.T) * g.Wc + (uint32_t)(x / g.T);
            // finer levels (refinement): label in the top 3 bits, element index into the linearised spread image below
            ent[f].off = ((uint32_t)label << 29) | e0;
            if (g.ls_bands)   // banded image: row and column of the matrix instead of the flat index (build_geometry checked the ranges)
              ent[f].off = ((uint32_t)label << 29) | ((uint32_t)((y % g.T) * g.T + (x % g.T)) * g.Hc + (uint32_t)(y / g.T)) << 12 | (uint32_t)(x / g.T);
            ent[f].x = (int16_t)x; ent[f].y = (int16_t)y;
            // coarsest level (scoring): (aligned dword index << 3) | (e0 & 7) into the nibble-packed memories;
            // upstream similarity() skips out-of-image features
            if (x < g.W && y < g.H) offs[f] = ((((uint32_t)label * g.nib_ori_stride) >> 2) + (e0 >> 3)) << 3 | (e0 & 7u);
          }
          for (int f = fc; f < kFeatStride; ++f) { ent[f].off = g.ls_zero_off; ent[f].x = 0; ent[f].y = 0; }
          ent[kFeatStride - 1].y = (int16_t)fc;   // entry 63 is always padding (<= 63 features): k_refine reads the row's feature count from it
          feat_l[l].insert(feat_l[l].end(), ent.begin(), ent.end());
          cnt_l[l].push_back((uint8_t)fc);
          if (l == L - 1) {
            coarse.insert(coarse.end(), offs.begin(), offs.end());
            nf_max = std::max(nf_max, fc);
          }
        }
        li.nf_total = nf_total;
        if (l == L - 1) {
          // unified table (see DeviceBankView): the last M rows of `coarse` are this template's.  Order: modalities interleaved
          // in groups of 3 (round robin), then regrouped by nibble shift (entry & 7): triples with ONE shift come first ("fast"
          // groups: the kernel sums the three dwords before the funnel shift), emitted round robin over the shift classes so
          // that the modalities stay mixed; the leftovers (< 3 per class) follow as mixed groups, the last one padded with
          // zero-run entries.  Entry 63 = fast groups | all groups << 8.
          std::vector<uint32_t> row(kFeatStride, (g.nib_zero_off >> 2) << 3);
          uint32_t row_groups = 0;
          if (nf_total <= kFeatStride - 1) {
            std::vector<int> next(M, 0), cnt(M);
            for (int m = 0; m < M; ++m) cnt[m] = cd.templates[((size_t)t * per + (size_t)l * M + m) * 5 + 4];
            std::vector<uint32_t> cls[8];
            for (bool any = true; any;) {
              any = false;
              for (int m = 0; m < M; ++m)
                for (int u = 0; u < 3 && next[m] < cnt[m]; ++u, any = true) {
                  const uint32_t e = coarse[coarse.size() - (size_t)(M - m) * kFeatStride + next[m]++] + ((((uint64_t)m * uni_block) >> 2) << 3);
                  cls[e & 7u].push_back(e);
                }
            }
            int n = 0, n_fast = 0;
            size_t taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (bool any = true; any;) {
              any = false;
              for (int k = 0; k < 8; ++k)
                if (cls[k].size() - taken[k] >= 3) {
                  for (int u = 0; u < 3; ++u) row[n++] = cls[k][taken[k]++];
                  ++n_fast;
                  any = true;
                }
            }
            for (int k = 0; k < 8; ++k)
              while (taken[k] < cls[k].size()) row[n++] = cls[k][taken[k]++];
            // final encoding of the unified table: byte offset (< 2^27, checked below) | funnel-shift bits (4 * nibble) << 27,
            // so that the kernel needs one scalar instruction for each
            for (int i = 0; i < kFeatStride - 1; ++i) row[i] = ((row[i] >> 3) << 2) | ((row[i] & 7u) * 4u) << 27;
            const int n_groups = (n + 2) / 3;
            row[kFeatStride - 1] = (uint32_t)n_fast | ((uint32_t)n_groups << 8);
            row_groups = row[kFeatStride - 1];
            // scalar-block row (k_score_coarse_sb): the same full triples in the same order, then one padded triple per shift class
            // that has leftovers; 5 groups per 16-dword block
            const uint32_t zero_entry = g.nib_zero_off & ~3u;   // byte offset of the zero run (modality 0's block; any shift reads zeros)
            std::vector<uint32_t> brow((size_t)SB_BLOCK * SB_MAX_BLOCKS, zero_entry);
            struct Grp { uint32_t off[3]; uint32_t shift; int real; };
            std::vector<Grp> grps;
            size_t tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (bool any = true; any;) {
              any = false;
              for (int k = 0; k < 8; ++k)
                if (cls[k].size() - tk[k] >= 3) {
                  Grp gr;
                  for (int u = 0; u < 3; ++u) gr.off[u] = (cls[k][tk[k]++] >> 3) << 2;
                  gr.shift = (uint32_t)k * 4u; gr.real = 3;
                  grps.push_back(gr);
                  any = true;
                }
            }
            for (int k = 0; k < 8; ++k)
              if (tk[k] < cls[k].size()) {
                Grp gr;
                gr.real = 0;
                for (int u = 0; u < 3; ++u) {
                  if (tk[k] < cls[k].size()) { gr.off[u] = (cls[k][tk[k]++] >> 3) << 2; gr.real += 1; }
                  else gr.off[u] = zero_entry;
                }
                gr.shift = (uint32_t)k * 4u;
                grps.push_back(gr);
              }
            const int n_blocks = ((int)grps.size() + SB_GROUPS - 1) / SB_GROUPS;
            if (n_blocks > SB_MAX_BLOCKS) uni_ok = false;   // cannot happen for <= 63 features (<= 21 full + 8 padded groups)
            else {
              int consumed = 0;
              for (int bi = 0; bi < n_blocks; ++bi) {
                uint32_t meta = 0;
                for (int q = 0; q < SB_GROUPS; ++q) {
                  const size_t gi = (size_t)bi * SB_GROUPS + q;
                  if (gi >= grps.size()) continue;
                  for (int u = 0; u < 3; ++u) brow[(size_t)bi * SB_BLOCK + 3 * q + u] = grps[gi].off[u];
                  meta |= grps[gi].shift << (5 * q);
                  consumed += grps[gi].real;
                }
                brow[(size_t)bi * SB_BLOCK + SB_BLOCK - 1] = meta | ((uint32_t)consumed << 25);
              }
              row_groups |= (uint32_t)n_blocks << 16;
              blk.insert(blk.end(), brow.begin(), brow.end());
            }
          } else {
            uni_ok = false;
          }
          if (blk.size() < (uni.size() / kFeatStride + 1) * (size_t)SB_BLOCK * SB_MAX_BLOCKS) blk.resize((uni.size() / kFeatStride + 1) * (size_t)SB_BLOCK * SB_MAX_BLOCKS, 0u);
          uni.insert(uni.end(), row.begin(), row.end());
          pending_groups = row_groups;
        }
        const int wf = (li.width - 1) / g.T + 1, hf = (li.height - 1) / g.T + 1;
        const long pos = (long)(g.Hc - hf) * g.Wc + (g.Wc - wf) + 1;
        li.positions = (int32_t)std::max<long>(0, std::min<long>(pos, (long)g.cells));
        linfo.push_back(li);
        if (l == L - 1) sinfo.push_back(ScoreInfo{li.positions, li.nf_total, ci, pending_groups});
      }
    }
    ++ci;
  }
  c->n_classes = ci;
  DeviceBankView& d = c->dbank;
  d.G = (int)info.size(); d.L = L; d.M = M; d.nf_max_coarse = nf_max;
  lmx_status st;
  if ((st = dev_upload(c, &d.info, info)) != LMX_OK) return st;
  if ((st = dev_upload(c, &d.linfo, linfo)) != LMX_OK) return st;
  if ((st = dev_upload(c, &d.coarse_off, coarse)) != LMX_OK) return st;
  if ((st = dev_upload(c, &d.coarse_uni, uni)) != LMX_OK) return st;
  if ((st = dev_upload(c, &d.coarse_blk, blk)) != LMX_OK) return st;
  if ((st = dev_upload(c, &d.sinfo, sinfo)) != LMX_OK) return st;
  d.uni_ok = (uni_ok && (uint64_t)M * uni_block + c->kp.geom[L - 1].nib_mod_stride < (1u << 27)) ? 1 : 0;  // byte offsets inside one frame's block
  d.uni_mod_block_bytes = uni_block;
  std::vector<FeatEntry> feat_all;
  std::vector<uint8_t> cnt_all;
  for (int l = 0; l < L; ++l) {
    feat_all.insert(feat_all.end(), feat_l[l].begin(), feat_l[l].end());
    cnt_all.insert(cnt_all.end(), cnt_l[l].begin(), cnt_l[l].end());
  }
  if ((st = dev_upload(c, &d.feat, feat_all)) != LMX_OK) return st;
  if ((st = dev_upload(c, &d.feat_count, cnt_all)) != LMX_OK) return st;
  return LMX_OK;
}

static bool get_events(lmx_ctx* c, hipEvent_t* a, hipEvent_t* b) {
  if (c->event_pool.empty()) {
    if (hipEventCreate(a) != hipSuccess) return false;
    if (hipEventCreate(b) != hipSuccess) return false;
    return true;
  }
  *a = c->event_pool.back().first; *b = c->event_pool.back().second;
  c->event_pool.pop_back();
  return true;
}

struct ScopedKernel {
  lmx_ctx* c; int id; hipEvent_t a{}, b{}; bool on = false;
  ScopedKernel(lmx_ctx* c_, int id_) : c(c_), id(id_) {
    if (((c->profiling >> id) & 1u) && get_events(c, &a, &b)) { on = true; (void)hipEventRecord(a, c->cur_stream); }
  }
  ~ScopedKernel() {
    if (on) { (void)hipEventRecord(b, c->cur_stream); c->pending.push_back({id, a, b}); }
  }
};

static void drain_profiling(lmx_ctx* c) {
  for (const ProfEvent& e : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) { c->k_ms[e.kernel] += ms; c->k_launches[e.kernel] += 1; }
    c->event_pool.emplace_back(e.start, e.stop);
  }
  c->pending.clear();
}

}  // namespace lmx

extern "C" {

const char* lmx_last_error(void) { return g_error.c_str(); }
const char* lmx_version(void) { return "lmx 0.1 (gfx950)"; }

// ---- bank -------------------------------------------------------------------------------------------------
lmx_status lmx_bank_create(const lmx_bank_desc* desc, lmx_bank** out) {
  return lmx::guarded("lmx_bank_create", [&]() -> lmx_status {
  if (!desc || !out || !desc->T || !desc->modalities) { set_error("lmx_bank_create: null argument"); return LMX_ERR_INVALID_ARG; }
  if (desc->pyramid_levels < 1 || desc->pyramid_levels > kMaxLevels) {
    set_error("pyramid_levels=%d unsupported (1..%d)", desc->pyramid_levels, kMaxLevels);
    return LMX_ERR_INVALID_ARG;
  }
  if (desc->n_modalities < 1 || desc->n_modalities > kMaxModalities) {
    set_error("n_modalities=%d unsupported (1..%d)", desc->n_modalities, kMaxModalities);
    return LMX_ERR_INVALID_ARG;
  }
  for (int m = 0; m < desc->n_modalities; ++m)
    if (desc->modalities[m].type != LMX_MOD_COLOR_GRADIENT && desc->modalities[m].type != LMX_MOD_DEPTH_NORMAL) {
      set_error("unknown modality type %d", desc->modalities[m].type);
      return LMX_ERR_INVALID_ARG;
    }
  lmx_bank* b = new lmx_bank();
  b->T.assign(desc->T, desc->T + desc->pyramid_levels);
  b->mods.assign(desc->modalities, desc->modalities + desc->n_modalities);
  b->normal_lut.resize(LMX_NORMAL_LUT_SIZE);
  default_normal_lut(b->normal_lut.data());
  b->normal_lut_origin = LMX_LUT_DEFAULT;
  *out = b;
  return LMX_OK;
  });
}

lmx_status lmx_default_normal_lut(uint8_t* out) {
  if (!out) { set_error("lmx_default_normal_lut: null argument"); return LMX_ERR_INVALID_ARG; }
  default_normal_lut(out);
  return LMX_OK;
}

lmx_status lmx_bank_set_normal_lut(lmx_bank* bank, const uint8_t* lut) {
  return lmx::guarded("lmx_bank_set_normal_lut", [&]() -> lmx_status {
  if (!bank) { set_error("lmx_bank_set_normal_lut: null bank"); return LMX_ERR_INVALID_ARG; }
  if (!lut) {
    default_normal_lut(bank->normal_lut.data());
    bank->normal_lut_origin = LMX_LUT_DEFAULT; bank->lut_epoch += 1;
    return LMX_OK;
  }
  std::vector<uint8_t> bins(LMX_NORMAL_LUT_SIZE);
  if (!normal_lut_to_bins(lut, bins.data())) { set_error("normal LUT entries must be 0 or a single bit (1, 2, 4, ..., 128)"); return LMX_ERR_INVALID_ARG; }
  bank->normal_lut.assign(lut, lut + LMX_NORMAL_LUT_SIZE);
  bank->normal_lut_origin = LMX_LUT_USER; bank->lut_epoch += 1;
  return LMX_OK;
  });
}

lmx_status lmx_bank_get_normal_lut(const lmx_bank* bank, uint8_t* out) {
  if (!bank || !out) { set_error("lmx_bank_get_normal_lut: null argument"); return LMX_ERR_INVALID_ARG; }
  std::memcpy(out, bank->normal_lut.data(), LMX_NORMAL_LUT_SIZE);
  return LMX_OK;
}

lmx_status lmx_bank_load_normal_lut(lmx_bank* bank, const char* path) {
  return lmx::guarded("lmx_bank_load_normal_lut", [&]() -> lmx_status {
  if (!bank || !path) { set_error("lmx_bank_load_normal_lut: null argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<uint8_t> lut;
  lmx_status st = normal_lut_from_file(path, lut);
  if (st != LMX_OK) return st;
  bank->normal_lut = lut;
  bank->normal_lut_origin = LMX_LUT_USER; bank->lut_epoch += 1;
  return LMX_OK;
  });
}

int32_t lmx_bank_normal_lut_origin(const lmx_bank* bank) { return bank ? bank->normal_lut_origin : -1; }

lmx_status lmx_bank_require_normal_lut(lmx_bank* bank) {
  if (!bank) { set_error("lmx_bank_require_normal_lut: null bank"); return LMX_ERR_INVALID_ARG; }
  if (bank->normal_lut_origin == LMX_LUT_USER || bank->normal_lut_origin == LMX_LUT_SIDECAR) return LMX_OK;
  const char* env = std::getenv("LMX_NORMAL_LUT");
  if (env && *env) {
    std::vector<uint8_t> lut;
    lmx_status st = normal_lut_from_file(env, lut);
    if (st != LMX_OK) return st;
    bank->normal_lut = lut;
    bank->normal_lut_origin = LMX_LUT_SIDECAR; bank->lut_epoch += 1;
    return LMX_OK;
  }
  bank->normal_lut_origin = LMX_LUT_UNKNOWN; bank->lut_epoch += 1;
  return LMX_OK;
}

lmx_status lmx_bank_add_class(lmx_bank* bank, const char* class_id, int32_t n_pyramids, const int32_t* templates,
                              const int32_t* features, int64_t n_features_total) {
  return lmx::guarded("lmx_bank_add_class", [&]() -> lmx_status {
  if (!bank || !class_id || n_pyramids < 0 || (n_pyramids > 0 && (!templates || !features))) {
    set_error("lmx_bank_add_class: invalid argument");
    return LMX_ERR_INVALID_ARG;
  }
  const int L = (int)bank->T.size(), M = (int)bank->mods.size(), per = L * M;
  // validate before touching the bank
  for (int64_t k = 0; k < (int64_t)n_pyramids * per; ++k) {
    const int32_t* t = templates + k * 5;
    const int l = (int)((k % per) / M);
    if (t[4] > 63) { set_error("template %ld has %d features; upstream similarity() asserts <= 63", (long)(k / per), t[4]); return LMX_ERR_SHAPE; }
    if (t[3] < 0 || t[4] < 0 || (int64_t)t[3] + t[4] > n_features_total) { set_error("template %ld: feature range out of bounds", (long)(k / per)); return LMX_ERR_INVALID_ARG; }
    if (t[2] != l) { set_error("template %ld entry %d: pyramid_level %d != %d", (long)(k / per), (int)(k % per), t[2], l); return LMX_ERR_INVALID_ARG; }
    if (t[0] < 0 || t[1] < 0 || t[0] > 65535 || t[1] > 65535) { set_error("template %ld: size %d x %d (cropTemplates yields 0 .. image size)", (long)(k / per), t[0], t[1]); return LMX_ERR_INVALID_ARG; }
    const int32_t* t0 = templates + (k - (k % M)) * 5;
    if (t[0] != t0[0] || t[1] != t0[1]) {
      set_error("template %ld level %d: modalities differ in width/height (cropTemplates gives one box per level)", (long)(k / per), l);
      return LMX_ERR_INVALID_ARG;
    }
    for (int f = 0; f < t[4]; ++f) {
      const int32_t* ft = features + ((int64_t)t[3] + f) * 3;
      if (ft[0] < 0 || ft[1] < 0 || ft[0] > 32767 || ft[1] > 32767 || ft[2] < 0 || ft[2] > 7) {
        set_error("template %ld: feature (%d,%d,%d) out of range", (long)(k / per), ft[0], ft[1], ft[2]);
        return LMX_ERR_INVALID_ARG;
      }
    }
  }
  ClassData& cd = bank->classes[class_id];
  cd.id = class_id;
  const int32_t fbase = (int32_t)(cd.features.size() / 3);
  for (int64_t k = 0; k < (int64_t)n_pyramids * per; ++k) {
    const int32_t* t = templates + k * 5;
    cd.templates.insert(cd.templates.end(), {t[0], t[1], t[2], t[3] + fbase, t[4]});
  }
  cd.features.insert(cd.features.end(), features, features + n_features_total * 3);
  cd.n_pyramids += n_pyramids;
  return LMX_OK;
  });
}

lmx_status lmx_bank_load_yaml(const char* path, lmx_bank** out) { return lmx::guarded("lmx_bank_load_yaml", [&]() -> lmx_status { return yaml_load(path, out); }); }
lmx_status lmx_bank_save_yaml(const lmx_bank* bank, const char* path) { return lmx::guarded("lmx_bank_save_yaml", [&]() -> lmx_status { return yaml_save(bank, path); }); }
void lmx_bank_destroy(lmx_bank* bank) { delete bank; }

int32_t lmx_bank_pyramid_levels(const lmx_bank* bank) { return bank ? (int32_t)bank->T.size() : 0; }
int32_t lmx_bank_T(const lmx_bank* bank, int32_t level) { return (bank && level >= 0 && level < (int)bank->T.size()) ? bank->T[level] : 0; }
int32_t lmx_bank_num_modalities(const lmx_bank* bank) { return bank ? (int32_t)bank->mods.size() : 0; }
lmx_status lmx_bank_modality(const lmx_bank* bank, int32_t index, lmx_modality_desc* out) {
  if (!bank || !out || index < 0 || index >= (int)bank->mods.size()) { set_error("lmx_bank_modality: bad index"); return LMX_ERR_INVALID_ARG; }
  *out = bank->mods[index];
  return LMX_OK;
}
int32_t lmx_bank_num_classes(const lmx_bank* bank) { return bank ? (int32_t)bank->classes.size() : 0; }
const char* lmx_bank_class_id(const lmx_bank* bank, int32_t class_index) {
  if (!bank || class_index < 0) return nullptr;
  int i = 0;
  for (const auto& kv : bank->classes)
    if (i++ == class_index) return kv.first.c_str();
  return nullptr;
}
int32_t lmx_bank_num_templates(const lmx_bank* bank, const char* class_id) {
  if (!bank) return 0;
  if (class_id) {
    auto it = bank->classes.find(class_id);
    return it == bank->classes.end() ? 0 : it->second.n_pyramids;
  }
  int32_t n = 0;
  for (const auto& kv : bank->classes) n += kv.second.n_pyramids;
  return n;
}
lmx_status lmx_bank_get_template(const lmx_bank* bank, const char* class_id, int32_t template_id, int32_t k, int32_t* width,
                                 int32_t* height, int32_t* pyramid_level, const int32_t** features, int32_t* n_features) {
  if (!bank || !class_id) { set_error("lmx_bank_get_template: null argument"); return LMX_ERR_INVALID_ARG; }
  auto it = bank->classes.find(class_id);
  if (it == bank->classes.end()) { set_error("class '%s' not in bank", class_id); return LMX_ERR_NOT_FOUND; }
  const ClassData& cd = it->second;
  const int per = (int)(bank->T.size() * bank->mods.size());
  if (template_id < 0 || template_id >= cd.n_pyramids || k < 0 || k >= per) { set_error("template index out of range"); return LMX_ERR_INVALID_ARG; }
  const int32_t* t = &cd.templates[((size_t)template_id * per + k) * 5];
  if (width) *width = t[0];
  if (height) *height = t[1];
  if (pyramid_level) *pyramid_level = t[2];
  if (features) *features = &cd.features[(size_t)t[3] * 3];
  if (n_features) *n_features = t[4];
  return LMX_OK;
}

// ---- context ----------------------------------------------------------------------------------------------
void lmx_ctx_destroy(lmx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int lane = 1; lane < lmx_ctx::kLanes; ++lane)
    if (c->lane_stream[lane]) { (void)hipStreamSynchronize(c->lane_stream[lane]); (void)hipStreamDestroy(c->lane_stream[lane]); }
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->pre_stream) { (void)hipStreamSynchronize(c->pre_stream); (void)hipStreamDestroy(c->pre_stream); }
  if (c->raw_dma_done) (void)hipEventDestroy(c->raw_dma_done);
  for (lmx_ctx::FrameSet& fs : c->sets) {
    if (fs.h_stage) (void)hipHostFree(fs.h_stage);
    if (fs.h_raw) (void)hipHostFree(fs.h_raw);
    if (fs.d_raw) (void)hipFree(fs.d_raw);
    if (fs.h_tab) (void)hipHostFree(fs.h_tab);
    if (fs.h2d_done) (void)hipEventDestroy(fs.h2d_done);
    for (hipEvent_t e : fs.read_done)
      if (e) (void)hipEventDestroy(e);
  }
  for (const ProfEvent& e : c->pending) { (void)hipEventDestroy(e.start); (void)hipEventDestroy(e.stop); }
  for (auto& pr : c->event_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& ge : c->graphs) (void)hipGraphExecDestroy(ge.exec);
  for (void* p : c->allocs) (void)hipFree(p);
  for (int i = 0; i < lmx_ctx::kSlots; ++i) {
    if (c->h_out_slot[i]) (void)hipHostFree(c->h_out_slot[i]);
    if (c->done[i]) (void)hipEventDestroy(c->done[i]);
  }
  if (c->d_f2_dists) (void)hipFree(c->d_f2_dists);
  if (c->d_f2_rects) (void)hipFree(c->d_f2_rects);
  if (c->f2_stream) { (void)hipStreamSynchronize(c->f2_stream); (void)hipStreamDestroy(c->f2_stream); }
  if (c->h_f2_out) (void)hipHostFree(c->h_f2_out);
  if (c->h_mask_stage) (void)hipHostFree(c->h_mask_stage);
  if (c->mask_h2d) (void)hipEventDestroy(c->mask_h2d);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// Points the context's working view (kp.fb, derived colour pyramid levels, candidate list) at one lane's buffers.
static void select_lane(lmx_ctx* c, int lane) {
  c->kp.fb = c->lane_fb[lane];
  for (int m = 0; m < c->M; ++m)
    for (int l = 1; l < c->L; ++l) c->mb[m].bgr[l] = c->lane_bgr[lane][m][l];
  c->d_cands = c->lane_cands[lane];
}

// Points the level-0 frame pointers at one frame set.
static void select_set(lmx_ctx* c, int set) {
  c->cur_set = set;
  const lmx_ctx::FrameSet& fs = c->sets[set];
  for (int m = 0; m < c->M; ++m) {
    c->mb[m].bgr[0] = fs.stored && fs.bgr[m] ? fs.store_buf[m] : fs.bgr[m];
    c->mb[m].depth = fs.stored && fs.depth[m] ? reinterpret_cast<uint16_t*>(fs.store_buf[m]) : fs.depth[m];
  }
}

// Host-side wait for everything queued on every lane and on the copy stream.
static lmx_status sync_lanes(lmx_ctx* c) {
  if (c->copy_stream) LMX_HIP(hipStreamSynchronize(c->copy_stream));
  if (c->pre_stream) LMX_HIP(hipStreamSynchronize(c->pre_stream));
  for (int lane = 0; lane < c->n_lanes; ++lane) LMX_HIP(hipStreamSynchronize(c->lane_stream[lane]));
  return LMX_OK;
}

static lmx_status ctx_create_impl(lmx_ctx* c) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error("no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return LMX_ERR_NO_DEVICE;
  }
  if (c->desc.device < 0 || c->desc.device >= ndev) { set_error("device %d out of range (%d devices)", c->desc.device, ndev); return LMX_ERR_NO_DEVICE; }
  c->device = c->desc.device;
  LMX_HIP(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  LMX_HIP(hipGetDeviceProperties(&prop, c->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; liblmx is built for gfx950 (MI355X) only", c->device, prop.gcnArchName);
    return LMX_ERR_NO_DEVICE;
  }
  if (c->desc.stream) c->stream = (hipStream_t)c->desc.stream;
  else { LMX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  c->lane_stream[0] = c->cur_stream = c->stream;
  c->n_lanes = (c->desc.flags & LMX_CTX_OVERLAP) ? lmx_ctx::kLanes : 1;
  c->n_slots = 2 * c->n_lanes;
  for (int lane = 1; lane < c->n_lanes; ++lane) LMX_HIP(hipStreamCreateWithFlags(&c->lane_stream[lane], hipStreamNonBlocking));
  LMX_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  c->n_sets = c->n_lanes + 1;

  lmx_status st = build_geometry(c);
  if (st != LMX_OK) return st;
  const int F = c->F;
  for (int m = 0; m < c->M; ++m) {
    const lmx_modality_desc& md = c->bank->mods[m];
    for (int set = 0; set < c->n_sets; ++set) {
      if (md.type == LMX_MOD_COLOR_GRADIENT) {
        if ((st = dev_alloc(c, &c->sets[set].bgr[m], (size_t)F * c->desc.width * c->desc.height * 3, false)) != LMX_OK) return st;
      } else {
        if ((st = dev_alloc(c, &c->sets[set].depth[m], (size_t)F * c->desc.width * c->desc.height, false)) != LMX_OK) return st;
      }
    }
    c->frame_bytes[m] = (size_t)c->desc.width * c->desc.height * (md.type == LMX_MOD_COLOR_GRADIENT ? 3 : 2);
    for (int lane = 0; lane < c->n_lanes; ++lane) {
      FrameBuffers& fb = c->lane_fb[lane];  // the uploaded frames belong to the frame sets, everything derived from them is per lane
      for (int l = 0; l < c->L; ++l) {
        const LevelGeom& g = c->kp.geom[l];
        if (md.type == LMX_MOD_COLOR_GRADIENT && l > 0 && (st = dev_alloc(c, &c->lane_bgr[lane][m][l], (size_t)F * g.W * g.H * 3, false)) != LMX_OK) return st;
        if ((st = dev_alloc(c, &fb.quant[l][m], (size_t)F * g.W * g.H, false)) != LMX_OK) return st;
        // pads must read as zero: clear once, kernels only ever write the matrices.  Byte + nibble-packed response memories
        // exist for the coarsest level only; finer levels keep the linearised spread image
        if (l == c->L - 1) {
          // the byte-wide memories are only an intermediate of the generic path (k_spread_linearize + k_pack_nibbles)
          if (!spread_writes_nibbles(g) && (st = dev_alloc(c, &fb.lm[l][m], (size_t)F * g.mod_stride + 8192, true)) != LMX_OK) return st;
          // the modalities' nibble memories of a lane are ONE allocation [M][F][nib_mod_stride] (+ zero tail): the u8 scoring
          // kernel addresses all of them from modality 0's base (DeviceBankView::coarse_uni)
          if (m == 0 && (st = dev_alloc(c, &fb.lmn[0], (size_t)c->M * F * g.nib_mod_stride + 8192, true)) != LMX_OK) return st;
          fb.lmn[m] = fb.lmn[0] + (size_t)m * F * g.nib_mod_stride;
        } else {
          if ((st = dev_alloc(c, &fb.ls[l][m], (size_t)F * g.ls_stride + 8192, true)) != LMX_OK) return st;
        }
      }
    }
  }
  if ((st = build_device_bank(c)) != LMX_OK) return st;
  {
    std::vector<uint8_t> bins(LMX_NORMAL_LUT_SIZE);
    if (!normal_lut_to_bins(c->bank->normal_lut.data(), bins.data())) { set_error("bank holds an invalid normal LUT"); return LMX_ERR_INVALID_ARG; }
    const uint8_t* d_bins = nullptr;
    if ((st = dev_upload(c, &d_bins, bins)) != LMX_OK) return st;
    c->d_normal_bins = const_cast<uint8_t*>(d_bins);
  }
  if ((st = dev_alloc(c, &c->d_class_slot, (size_t)std::max(1, c->n_classes), true)) != LMX_OK) return st;
  c->cur_slots.assign(c->n_classes, -2);
  const uint32_t per_frame = c->desc.max_candidates > 0 ? (uint32_t)c->desc.max_candidates : 16384u;
  c->cap_total = per_frame * (uint32_t)F;
  for (int lane = 0; lane < c->n_lanes; ++lane)
    if ((st = dev_alloc(c, &c->lane_cands[lane], lmx::cand_list_entries(c->cap_total), false)) != LMX_OK) return st;
  select_lane(c, 0);
  c->h_out_records = c->cap_total;
  for (int i = 0; i < c->n_slots; ++i) {
    // the candidate list's stripe counters live in front of the slot's header (lmx_internal.hpp): one reset clears both
    if ((st = dev_alloc(c, &c->d_out_slot[i], lmx::kStripeAreaBytes + 64 + (size_t)c->cap_total * sizeof(lmx_raw_match_t), true)) != LMX_OK) return st;
    c->d_out_slot[i] += lmx::kStripeAreaBytes;
    LMX_HIP(hipHostMalloc((void**)&c->h_out_slot[i], 64 + c->h_out_records * sizeof(lmx_raw_match_t), hipHostMallocMapped));
    LMX_HIP(hipHostGetDevicePointer((void**)&c->h_out_dev[i], c->h_out_slot[i], 0));
    std::memset(c->h_out_slot[i], 0, 64);
    LMX_HIP(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
  }
  if ((st = dev_alloc(c, &c->d_pub_counter, (size_t)lmx_ctx::kSlots, true)) != LMX_OK) return st;
  c->d_out = c->d_out_slot[0];
  c->h_out = c->h_out_slot[0];
  size_t stage = 0;
  for (int m = 0; m < c->M; ++m) stage += c->frame_bytes[m];
  c->h_stage_bytes = stage * F;
  for (int set = 0; set < c->n_sets; ++set) {
    lmx_ctx::FrameSet& fs = c->sets[set];
    if (!(c->desc.flags & lmx::LMX_CTX_EXTERNAL_STAGING)) LMX_HIP(hipHostMalloc((void**)&fs.h_stage, c->h_stage_bytes, hipHostMallocDefault));
    LMX_HIP(hipHostMalloc((void**)&fs.h_tab, sizeof(PullEntry) * (size_t)c->M * F, hipHostMallocMapped));
    LMX_HIP(hipHostGetDevicePointer((void**)&fs.d_tab, fs.h_tab, 0));
    LMX_HIP(hipEventCreateWithFlags(&fs.h2d_done, hipEventDisableTiming));
    for (int lane = 0; lane < c->n_lanes; ++lane) LMX_HIP(hipEventCreateWithFlags(&fs.read_done[lane], hipEventDisableTiming));
  }
  {
    // host-writable device buffers for the direct-store upload of small batches; graphs bake the frame pointers in, so not with them
    hipDeviceProp_t prop;
    c->store_ok = std::getenv("LMX_NO_STORE_UPLOAD") == nullptr && !(c->desc.flags & LMX_CTX_HIPGRAPH) &&
                  hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.isLargeBar;
    for (int set = 0; set < c->n_sets && c->store_ok; ++set)
      for (int m = 0; m < c->M && c->store_ok; ++m) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, c->frame_bytes[m] * (size_t)std::min(F, (int)lmx_ctx::kStoreFrames), hipDeviceMallocFinegrained) != hipSuccess) {
          (void)hipGetLastError();
          c->store_ok = false;
        } else {
          c->sets[set].store_buf[m] = static_cast<uint8_t*>(p);
          c->allocs.push_back(p);
        }
      }
  }
  select_set(c, 0);
  LMX_HIP(hipStreamSynchronize(c->stream));
  return LMX_OK;
}

lmx_status lmx_ctx_create(const lmx_bank* bank, const lmx_ctx_desc* desc, lmx_ctx** out) {
  return lmx::guarded("lmx_ctx_create", [&]() -> lmx_status {
  if (!bank || !desc || !out) { set_error("lmx_ctx_create: null argument"); return LMX_ERR_INVALID_ARG; }
  if (desc->max_batch < 1) { set_error("max_batch must be >= 1"); return LMX_ERR_INVALID_ARG; }
  if (bank->normal_lut_origin == LMX_LUT_UNKNOWN) {
    for (const lmx_modality_desc& md : bank->mods)
      if (md.type == LMX_MOD_DEPTH_NORMAL) {
        set_error("this bank has a DepthNormal modality and was read from a yml without a normal-LUT marker or side-car: it was trained against "
                  "OpenCV's NORMAL_LUT (normal_lut.i), which this library does not contain.  Supply it (lmx_bank_load_normal_lut / "
                  "lmx_bank_set_normal_lut / <yml>.normal_lut / LMX_NORMAL_LUT=<file>) or choose the default generator explicitly with "
                  "lmx_bank_set_normal_lut(bank, NULL)");
        return LMX_ERR_INVALID_ARG;
      }
  }
  if (desc->shard_world > 1 && (desc->shard_rank < 0 || desc->shard_rank >= desc->shard_world)) {
    set_error("shard_rank %d outside [0,%d)", desc->shard_rank, desc->shard_world);
    return LMX_ERR_INVALID_ARG;
  }
  lmx_ctx* c = new lmx_ctx();
  c->bank = bank; c->desc = *desc;
  if (c->desc.shard_world <= 1) { c->desc.shard_world = 1; c->desc.shard_rank = 0; }
  c->L = (int)bank->T.size(); c->M = (int)bank->mods.size(); c->F = desc->max_batch;
  c->trace_collect = std::getenv("LMX_COLLECT_TRACE") != nullptr;
  if (const char* pm = std::getenv("LMX_PINNED_MODE")) c->env_pinned_mode = std::strcmp(pm, "dma") == 0 ? 1 : (std::strcmp(pm, "stage") == 0 ? 2 : 0);
  c->env_no_small_chain = std::getenv("LMX_NO_SMALL_CHAIN") != nullptr;
  if (const char* e = std::getenv("LMX_CAND_STRIPES")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= lmx::kCandStripes && (v & (v - 1)) == 0) c->cand_stripes = v;
  }
  c->env_debug_collect = std::getenv("LMX_DEBUG_COLLECT") != nullptr;
  if (const char* e = std::getenv("LMX_UPLOAD_THREADS")) c->env_upload_threads = std::max(0, std::min(std::atoi(e), 64));
  {
    const char* e = std::getenv("LMX_SCORE_KERNEL");
    c->dbank.score_variant = (std::getenv("LMX_SCORE_GENERIC") != nullptr || (e && std::strcmp(e, "generic") == 0)) ? 0 : ((e && std::strcmp(e, "u8") == 0) ? 1 : 2);
  }
  lmx_status st = ctx_create_impl(c);
  if (st != LMX_OK) { std::string keep = g_error; lmx_ctx_destroy(c); g_error = keep; return st; }
  *out = c;
  return LMX_OK;
  });
}

static int upload_threads(const lmx_ctx* c) {
  if (c->env_upload_threads >= 1) return c->env_upload_threads;
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(8u, hw ? hw / 2 : 1u));
}

// pinned (page-locked, device-visible) host memory: the DMA engine can read it in place
static bool is_pinned_host(const void* p, const void** device_view) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }  // plain malloc memory: "invalid value"
  if (attr.type != hipMemoryTypeHost || attr.devicePointer == nullptr) return false;
  *device_view = attr.devicePointer;
  return true;
}

// Opens frame set `set` for a new upload (see lmx_ctx::FrameSet): host-side wait for the previous transfer out of its staging,
// device-side wait of the copy stream for the lanes that still read the set.
static lmx_status begin_set_upload(lmx_ctx* c, int set) {
  lmx_ctx::FrameSet& fs = c->sets[set];
  for (int m = 0; m < c->M; ++m) fs.masked[m] = false;   // masks belong to the frames they were uploaded for
  if (fs.h2d_recorded) LMX_HIP(hipEventSynchronize(fs.h2d_done));
  for (int lane = 0; lane < c->n_lanes; ++lane)
    if (fs.read_recorded[lane]) LMX_HIP(hipStreamWaitEvent(c->copy_stream, fs.read_done[lane], 0));
  return LMX_OK;
}
static lmx_status end_set_upload(lmx_ctx* c, int set, hipStream_t last = nullptr) {
  lmx_ctx::FrameSet& fs = c->sets[set];
  LMX_HIP(hipEventRecord(fs.h2d_done, last ? last : c->copy_stream));
  fs.h2d_recorded = true;
  select_set(c, set);
  return LMX_OK;
}

static void store_modality(lmx_ctx* c, lmx_ctx::FrameSet& fs, int m, int n_frames, const lmx_image* sources);

lmx_status lmx_ctx_upload(lmx_ctx* c, int32_t n_frames, const lmx_image* sources, int32_t n_sources) {
  return lmx::guarded("lmx_ctx_upload", [&]() -> lmx_status {
  if (!c || !sources) { set_error("lmx_ctx_upload: null argument"); return LMX_ERR_INVALID_ARG; }
  lmx_status st = lmx::ctx_check_sources(c, n_frames, sources, n_sources);
  if (st != LMX_OK) return st;
  LMX_HIP(hipSetDevice(c->device));
  const int W = c->desc.width, H = c->desc.height;
  // next frame set: (host) the previous transfer out of its staging buffer has finished; (device, copy stream) every enqueue
  // that reads the set's old frames is past its last kernel.  Neither waits for a lane to drain.
  const int set = (c->cur_set + 1) % c->n_sets;
  st = begin_set_upload(c, set);
  if (st != LMX_OK) return st;
  lmx_ctx::FrameSet& fs = c->sets[set];
  // staging: one task per (modality, frame); pinned sources (hipHostMalloc / hipHostRegister'ed caller memory) skip it
  struct Task { uint8_t* dst; const uint8_t* src; size_t row_bytes, src_stride; int rows; };
  std::vector<Task> tasks;
  fs.stored = false;
  if (c->store_ok && n_frames <= lmx_ctx::kStoreFrames && !(c->desc.flags & LMX_CTX_ASYNC_INPUT)) {
    // Direct store (see FrameSet::store_buf): the lanes that still read this set's previous frames are waited for on the HOST here
    // (with one frame per call they finished long ago), then the rows go straight into device memory.
    for (int lane = 0; lane < c->n_lanes; ++lane)
      if (fs.read_recorded[lane]) LMX_HIP(hipEventSynchronize(fs.read_done[lane]));
    if (c->deferred_frames == -1) {   // lmx_match_batch: the enqueue that follows writes the frames between its launches (issue_small)
      c->deferred_sources = sources;
      c->deferred_frames = n_frames;
    } else {
      for (int m = 0; m < c->M; ++m) store_modality(c, fs, m, n_frames, sources);
    }
    // nothing was queued on the copy stream and the stores are globally visible (sfence inside stream_copy; posted writes reach the
    // device before the doorbell of any later launch): the enqueue has no transfer event to wait for
    fs.stored = true;
    fs.h2d_recorded = false;
    fs.n_uploaded = n_frames;
    select_set(c, set);
    return LMX_OK;
  }
  std::vector<int> direct(c->M, 0);   // modality m: every frame is pinned caller memory -> DMA straight from it
  const bool async_input = (c->desc.flags & LMX_CTX_ASYNC_INPUT) != 0;
  size_t off = 0;
  for (int m = 0; m < c->M; ++m) {
    const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    const size_t row_bytes = (size_t)W * (cg ? 3 : 2);
    // LMX_PINNED_MODE (measurement switch): "pull" = one kernel pulls every pinned image, "dma" = one hipMemcpyAsync per image,
    // "stage" = treat pinned sources like pageable ones
    // Pinned caller memory is read in place only when the caller asked for it (LMX_CTX_ASYNC_INPUT: no host copy at all, the
    // transfer is a kernel pulling over PCIe).  Otherwise pinned sources are staged like pageable ones: measured, the staging copy
    // with non-temporal stores + one DMA per modality moves 54.5 GB/s end to end, the pull kernel 44 GB/s (it competes with the
    // compute kernels for CUs) and per-image DMA calls 32 GB/s (profiles/r02_host_frame_transfer_modes.txt).
    const int pinned_mode = c->env_pinned_mode < 0 ? (async_input ? 0 : 2) : c->env_pinned_mode;
    bool all_pinned = pinned_mode != 2;
    for (int f = 0; f < n_frames && all_pinned; ++f) {
      const lmx_image& im = sources[(size_t)f * c->M + m];
      const void* dv = nullptr;
      all_pinned = is_pinned_host(im.data, &dv);
      if (all_pinned) fs.h_tab[(size_t)m * c->F + f] = PullEntry{(uint64_t)(uintptr_t)dv, (uint64_t)im.row_stride_bytes};
    }
    direct[m] = all_pinned ? 1 : 0;
    if (!all_pinned && !fs.h_stage) { set_error("lmx_ctx_upload: this context is a member of a device group and is fed through lmx_group_upload"); return LMX_ERR_INVALID_ARG; }
    if (!all_pinned) {
      // one task per image for batches, row bands for a few frames (a single 640x480 RGB-D frame is still 1.5 MB: 60 us on one
      // thread, a third of the whole single-frame call)
      const int bands = n_frames >= 8 ? 1 : std::max(1, std::min(8, H / 64));
      for (int f = 0; f < n_frames; ++f) {
        const lmx_image& im = sources[(size_t)f * c->M + m];
        for (int b = 0; b < bands; ++b) {
          const int y0 = (int)((long)H * b / bands), y1 = (int)((long)H * (b + 1) / bands);
          tasks.push_back(Task{fs.h_stage + off + (size_t)f * c->frame_bytes[m] + (size_t)y0 * row_bytes, (const uint8_t*)im.data + (size_t)y0 * im.row_stride_bytes, row_bytes,
                               im.row_stride_bytes, y1 - y0});
        }
      }
    }
    off += c->frame_bytes[m] * c->F;
  }
  if (!tasks.empty()) {
    if (!c->pool) c->pool.reset(new CopyPool(upload_threads(c) - 1));
    c->pool->parallel_for((int)tasks.size(), [&](int i) {
      const Task& t = tasks[i];
      if (t.src_stride == t.row_bytes) stream_copy(t.dst, t.src, t.row_bytes * t.rows);
      else
        for (int y = 0; y < t.rows; ++y) stream_copy(t.dst + (size_t)y * t.row_bytes, t.src + (size_t)y * t.src_stride, t.row_bytes);
    });
  }
  off = 0;
  for (int m = 0; m < c->M; ++m) {
    const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    const size_t row_bytes = (size_t)W * (cg ? 3 : 2);
    uint8_t* dst = cg ? fs.bgr[m] : reinterpret_cast<uint8_t*>(fs.depth[m]);
    if (!direct[m]) {
      LMX_HIP(hipMemcpyAsync(dst, fs.h_stage + off, c->frame_bytes[m] * n_frames, hipMemcpyHostToDevice, c->copy_stream));
    } else {
      // caller-owned pinned images: one kernel pulls all frames of the modality over PCIe (per-image DMA calls were measured at
      // 32 GB/s against 57 GB/s for this form)
      if (c->env_pinned_mode == 1) {
        for (int f = 0; f < n_frames; ++f) {
          const lmx_image& im = sources[(size_t)f * c->M + m];
          if (im.row_stride_bytes == row_bytes)
            LMX_HIP(hipMemcpyAsync(dst + (size_t)f * c->frame_bytes[m], im.data, c->frame_bytes[m], hipMemcpyHostToDevice, c->copy_stream));
          else
            LMX_HIP(hipMemcpy2DAsync(dst + (size_t)f * c->frame_bytes[m], row_bytes, im.data, im.row_stride_bytes, row_bytes, H, hipMemcpyHostToDevice, c->copy_stream));
        }
      } else {
        launch_pull_frames(c->copy_stream, fs.d_tab + (size_t)m * c->F, dst, c->frame_bytes[m], H, (uint32_t)row_bytes, n_frames);
        LMX_HIP(hipGetLastError());
      }
    }
    off += c->frame_bytes[m] * c->F;
  }
  fs.n_uploaded = n_frames;
  st = end_set_upload(c, set);
  if (st != LMX_OK) return st;
  // pinned caller memory is read by the DMA engine after this call returns: only a caller that asked for it (LMX_CTX_ASYNC_INPUT)
  // gets that; by default the call keeps the "callee copies, never retains pointers" contract of the boundary
  bool any_direct = false;
  for (int m = 0; m < c->M; ++m) any_direct = any_direct || direct[m];
  if (any_direct && !async_input) LMX_HIP(hipEventSynchronize(fs.h2d_done));
  return LMX_OK;
  });
}

lmx_status lmx_ctx_upload_masks(lmx_ctx* c, int32_t n_frames, const lmx_image* masks, int32_t n_masks) {
  return lmx::guarded("lmx_ctx_upload_masks", [&]() -> lmx_status {
  if (!c || !masks) { set_error("lmx_ctx_upload_masks: null argument"); return LMX_ERR_INVALID_ARG; }
  if (n_masks != c->M) { set_error("masks.size()=%d != modalities.size()=%d (upstream CV_Assert in Detector::match)", n_masks, c->M); return LMX_ERR_SHAPE; }
  lmx_ctx::FrameSet& fs = c->sets[c->cur_set];
  if (n_frames < 1 || n_frames > fs.n_uploaded) { set_error("lmx_ctx_upload_masks: n_frames=%d but the most recent upload holds %d frame(s)", n_frames, fs.n_uploaded); return LMX_ERR_INVALID_ARG; }
  const int W = c->desc.width, H = c->desc.height;
  for (int f = 0; f < n_frames; ++f)
    for (int m = 0; m < c->M; ++m) {
      const lmx_image& im = masks[(size_t)f * c->M + m];
      if (!im.data) continue;   // an empty Mat: no mask for this source
      if (im.rows != H || im.cols != W) { set_error("frame %d mask %d: size %dx%d != source %dx%d (upstream CV_Assert)", f, m, im.cols, im.rows, W, H); return LMX_ERR_SHAPE; }
      if (im.channels != 1 || im.elem_size != 1 || im.row_stride_bytes < (size_t)W) { set_error("frame %d mask %d: masks are 8UC1", f, m); return LMX_ERR_SHAPE; }
    }
  LMX_HIP(hipSetDevice(c->device));
  const size_t frame_px = (size_t)W * H;
  if (!c->h_mask_stage) {
    LMX_HIP(hipHostMalloc((void**)&c->h_mask_stage, frame_px * (size_t)c->F, hipHostMallocDefault));
    LMX_HIP(hipEventCreateWithFlags(&c->mask_h2d, hipEventDisableTiming));
  }
  // An enqueue that follows may still be reading this set's PREVIOUS masks on a lane when this is a second upload_masks for the same
  // frames (upload -> masks -> enqueue -> masks -> enqueue): the copy stream waits for those readers like begin_set_upload does for the
  // frames (advisor finding, round 3)
  for (int lane = 0; lane < c->n_lanes; ++lane)
    if (fs.read_recorded[lane]) LMX_HIP(hipStreamWaitEvent(c->copy_stream, fs.read_done[lane], 0));
  for (int m = 0; m < c->M; ++m) {
    bool any = false;
    for (int f = 0; f < n_frames; ++f) any = any || masks[(size_t)f * c->M + m].data != nullptr;
    if (!any) { fs.masked[m] = false; continue; }
    if (!fs.mask[m]) {
      lmx_status st = dev_alloc(c, &fs.mask[m], frame_px * (size_t)c->F, false);
      if (st != LMX_OK) return st;
    }
    // one modality at a time through the single staging buffer: the previous modality's transfer has to have left it
    LMX_HIP(hipStreamSynchronize(c->copy_stream));
    // masked[m] holds for every frame of the set: frames the caller gave no mask for -- an empty Mat, or frames [n_frames, n_uploaded) --
    // get an all-pass mask, so that an enqueue of all uploaded frames never reads mask memory nobody wrote
    for (int f = 0; f < fs.n_uploaded; ++f) {
      uint8_t* dst = c->h_mask_stage + (size_t)f * frame_px;
      if (f >= n_frames || !masks[(size_t)f * c->M + m].data) { std::memset(dst, 255, frame_px); continue; }
      const lmx_image& im = masks[(size_t)f * c->M + m];
      for (int y = 0; y < H; ++y) std::memcpy(dst + (size_t)y * W, (const uint8_t*)im.data + (size_t)y * im.row_stride_bytes, (size_t)W);
    }
    LMX_HIP(hipMemcpyAsync(fs.mask[m], c->h_mask_stage, frame_px * (size_t)fs.n_uploaded, hipMemcpyHostToDevice, c->copy_stream));
    fs.masked[m] = true;
  }
  // an enqueue waits for the set's h2d_done: record it again behind the masks (direct-store uploads recorded nothing: now they do); what
  // the event stood for so far (lmx_ctx_upload_raw records it on its kernel stream) stays part of it
  if (fs.h2d_recorded) LMX_HIP(hipStreamWaitEvent(c->copy_stream, fs.h2d_done, 0));
  LMX_HIP(hipEventRecord(fs.h2d_done, c->copy_stream));
  fs.h2d_recorded = true;
  LMX_HIP(hipStreamSynchronize(c->copy_stream));   // the caller's masks and the staging buffer are free again when this returns
  return LMX_OK;
  });
}

lmx_status lmx_match_masked(lmx_ctx* c, const lmx_image* sources, const lmx_image* masks, int32_t n_sources, float threshold, const char* const* class_ids,
                            int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_match_masked", [&]() -> lmx_status {
  if (!c) { set_error("lmx_match_masked: null context"); return LMX_ERR_INVALID_ARG; }
  if (!masks) return lmx_match(c, sources, n_sources, threshold, class_ids, n_class_ids, out, cap, n_out);
  std::lock_guard<std::recursive_mutex> lk(c->call_mutex);
  lmx_status st = lmx_ctx_upload(c, 1, sources, n_sources);
  if (st == LMX_OK) st = lmx_ctx_upload_masks(c, 1, masks, n_sources);
  if (st == LMX_OK) st = lmx_ctx_enqueue(c, 1, threshold, class_ids, n_class_ids);
  if (st != LMX_OK) return st;
  return lmx_ctx_collect(c, 1, out, cap, n_out);
  });
}

lmx_status lmx_ctx_upload_wait(lmx_ctx* c) {
  if (!c) { set_error("lmx_ctx_upload_wait: null context"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  lmx_ctx::FrameSet& fs = c->sets[c->cur_set];
  if (fs.h2d_recorded) LMX_HIP(hipEventSynchronize(fs.h2d_done));
  return LMX_OK;
}

}  // extern "C"

// ---- hooks for device groups (lmx_internal.hpp) ---------------------------------------------------------------------------
namespace lmx {

int ctx_num_sets(const lmx_ctx* c) { return c->n_sets; }
int ctx_next_set(const lmx_ctx* c) { return (c->cur_set + 1) % c->n_sets; }
size_t ctx_stage_bytes(const lmx_ctx* c) { return c->h_stage_bytes; }

lmx_status ctx_check_sources(lmx_ctx* c, int n_frames, const lmx_image* sources, int n_sources) {
  if (n_sources != c->M) {
    set_error("sources.size()=%d != modalities.size()=%d (upstream CV_Assert in Detector::match)", n_sources, c->M);
    return LMX_ERR_SHAPE;
  }
  if (n_frames < 1 || n_frames > c->F) { set_error("n_frames=%d outside [1,%d]", n_frames, c->F); return LMX_ERR_INVALID_ARG; }
  const int W = c->desc.width, H = c->desc.height;
  for (int f = 0; f < n_frames; ++f)
    for (int m = 0; m < c->M; ++m) {
      const lmx_image& im = sources[(size_t)f * c->M + m];
      const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
      const int want_ch = cg ? 3 : 1, want_es = cg ? 1 : 2;
      if (!im.data || im.rows != H || im.cols != W) { set_error("frame %d source %d: size %dx%d != context %dx%d", f, m, im.cols, im.rows, W, H); return LMX_ERR_SHAPE; }
      if (im.channels != want_ch || im.elem_size != want_es) {
        set_error("frame %d source %d: %s wants %s", f, m, cg ? "ColorGradient" : "DepthNormal", cg ? "8UC3" : "16UC1");
        return LMX_ERR_SHAPE;
      }
      if (im.row_stride_bytes < (size_t)W * want_ch * want_es) { set_error("frame %d source %d: row stride too small", f, m); return LMX_ERR_INVALID_ARG; }
    }
  return LMX_OK;
}

// Copies every source of a batch into a pinned staging area with the layout of FrameSet::h_stage (modality m at offset
// sum_{m' < m} frame_bytes[m'] * max_batch, frames back to back, rows packed): one task per image for batches, row bands for a few
// frames (a single 640x480 RGB-D frame is still 1.5 MB: 60 us on one thread).
void ctx_stage_sources(lmx_ctx* c, CopyPool* pool, uint8_t* base, int n_frames, const lmx_image* sources) {
  struct Task { uint8_t* dst; const uint8_t* src; size_t row_bytes, src_stride; int rows; };
  std::vector<Task> tasks;
  const int W = c->desc.width, H = c->desc.height;
  size_t off = 0;
  for (int m = 0; m < c->M; ++m) {
    const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    const size_t row_bytes = (size_t)W * (cg ? 3 : 2);
    const int bands = n_frames >= 8 ? 1 : std::max(1, std::min(8, H / 64));
    for (int f = 0; f < n_frames; ++f) {
      const lmx_image& im = sources[(size_t)f * c->M + m];
      for (int b = 0; b < bands; ++b) {
        const int y0 = (int)((long)H * b / bands), y1 = (int)((long)H * (b + 1) / bands);
        tasks.push_back(Task{base + off + (size_t)f * c->frame_bytes[m] + (size_t)y0 * row_bytes, (const uint8_t*)im.data + (size_t)y0 * im.row_stride_bytes, row_bytes,
                             im.row_stride_bytes, y1 - y0});
      }
    }
    off += c->frame_bytes[m] * c->F;
  }
  auto run = [&](int i) {
    const Task& t = tasks[i];
    if (t.src_stride == t.row_bytes) stream_copy(t.dst, t.src, t.row_bytes * t.rows);
    else
      for (int y = 0; y < t.rows; ++y) stream_copy(t.dst + (size_t)y * t.row_bytes, t.src + (size_t)y * t.src_stride, t.row_bytes);
  };
  if (pool) pool->parallel_for((int)tasks.size(), run);
  else
    for (int i = 0; i < (int)tasks.size(); ++i) run(i);
}

lmx_status ctx_begin_staged_upload(lmx_ctx* c) {
  LMX_HIP(hipSetDevice(c->device));
  return begin_set_upload(c, (c->cur_set + 1) % c->n_sets);
}

lmx_status ctx_finish_staged_upload(lmx_ctx* c, int n_frames, const uint8_t* pinned) {
  LMX_HIP(hipSetDevice(c->device));
  const int set = (c->cur_set + 1) % c->n_sets;
  lmx_ctx::FrameSet& fs = c->sets[set];
  fs.stored = false;
  size_t off = 0;
  for (int m = 0; m < c->M; ++m) {
    const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    uint8_t* dst = cg ? fs.bgr[m] : reinterpret_cast<uint8_t*>(fs.depth[m]);
    LMX_HIP(hipMemcpyAsync(dst, pinned + off, c->frame_bytes[m] * n_frames, hipMemcpyHostToDevice, c->copy_stream));
    off += c->frame_bytes[m] * c->F;
  }
  fs.n_uploaded = n_frames;
  return end_set_upload(c, set);
}

lmx_status ctx_drop_newest(lmx_ctx* c) {
  if (c->outstanding < 1) { set_error("ctx_drop_newest: nothing enqueued"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  const int slot = (c->head + c->n_slots - 1) % c->n_slots;
  LMX_HIP(hipEventSynchronize(c->done[slot]));
  c->head = slot;
  c->outstanding -= 1;
  if (c->outstanding == 0) drain_profiling(c);
  return LMX_OK;
}

}  // namespace lmx

extern "C" {

lmx_status lmx_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) { set_error("lmx_host_alloc: invalid argument"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return LMX_OK;
}
void lmx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

lmx_status lmx_ctx_upload_raw(lmx_ctx* c, int32_t n_frames, const lmx_image* sources, int32_t n_sources, const lmx_pre_desc* pre) {
  return lmx::guarded("lmx_ctx_upload_raw", [&]() -> lmx_status {
  if (!c || !sources || !pre) { set_error("lmx_ctx_upload_raw: null argument"); return LMX_ERR_INVALID_ARG; }
  if (n_sources != c->M) {
    set_error("sources.size()=%d != modalities.size()=%d (upstream CV_Assert in Detector::match)", n_sources, c->M);
    return LMX_ERR_SHAPE;
  }
  if (n_frames < 1 || n_frames > c->F) { set_error("n_frames=%d outside [1,%d]", n_frames, c->F); return LMX_ERR_INVALID_ARG; }
  const int W = c->desc.width, H = c->desc.height;
  if (pre->src_width < W || pre->src_height < H || pre->crop_x < 0 || pre->crop_y < 0 || pre->crop_x + W > pre->src_width ||
      pre->crop_y + H > pre->src_height) {
    set_error("crop %dx%d at (%d,%d) does not fit the %dx%d source", W, H, pre->crop_x, pre->crop_y, pre->src_width, pre->src_height);
    return LMX_ERR_SHAPE;
  }
  LMX_HIP(hipSetDevice(c->device));
  // per modality: element layout of the raw source and whether it is full-size (cropped on device) or already frame-size
  struct Raw { int sh, sw, ch, es, cx, cy; size_t bytes; };
  std::vector<Raw> raw(c->M);
  size_t total = 0;
  for (int m = 0; m < c->M; ++m) {
    const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    const int ch = cg ? (pre->mono ? 1 : 3) : 1, es = cg ? 1 : (pre->depth_float_m ? 4 : 2);
    for (int f = 0; f < n_frames; ++f) {
      const lmx_image& im = sources[(size_t)f * c->M + m];
      const bool full = im.rows == pre->src_height && im.cols == pre->src_width;
      const bool fit = im.rows == H && im.cols == W;
      if (!im.data || !(full || fit)) { set_error("frame %d source %d: size %dx%d is neither the raw %dx%d nor the context %dx%d", f, m, im.cols, im.rows, pre->src_width, pre->src_height, W, H); return LMX_ERR_SHAPE; }
      if (im.channels != ch || im.elem_size != es) { set_error("frame %d source %d: expected %d channel(s) of %d byte(s)", f, m, ch, es); return LMX_ERR_SHAPE; }
      if (im.row_stride_bytes < (size_t)im.cols * ch * es) { set_error("frame %d source %d: row stride too small", f, m); return LMX_ERR_INVALID_ARG; }
      if (f == 0) raw[m] = Raw{im.rows, im.cols, ch, es, full ? pre->crop_x : 0, full ? pre->crop_y : 0, (size_t)im.rows * im.cols * ch * es};
      else if (im.rows != raw[m].sh || im.cols != raw[m].sw) { set_error("frame %d source %d: size differs from frame 0", f, m); return LMX_ERR_SHAPE; }
    }
    total += raw[m].bytes * n_frames;
  }
  // next frame set: its previous transfer AND the pre-processing kernels behind it have finished (h2d_done is recorded behind them), so
  // the set's raw staging may be overwritten; the lanes that still read the set's frames are waited for on the copy stream
  const int set = (c->cur_set + 1) % c->n_sets;
  lmx_status st = begin_set_upload(c, set);
  if (st != LMX_OK) return st;
  lmx_ctx::FrameSet& fs = c->sets[set];
  fs.stored = false;   // the pre-processing kernels write the regular frame buffers
  if (total > fs.raw_bytes) {
    if (fs.h_raw) (void)hipHostFree(fs.h_raw);
    if (fs.d_raw) (void)hipFree(fs.d_raw);
    fs.h_raw = nullptr; fs.d_raw = nullptr; fs.raw_bytes = 0;
    LMX_HIP(hipHostMalloc((void**)&fs.h_raw, total, hipHostMallocDefault));
    LMX_HIP(hipMalloc((void**)&fs.d_raw, total));
    fs.raw_bytes = total;
  }
  // staging with non-temporal stores on the upload threads (one task per image), like lmx_ctx_upload
  struct Task { uint8_t* dst; const uint8_t* src; size_t row_bytes, src_stride; int rows; };
  std::vector<Task> tasks;
  size_t off = 0;
  for (int m = 0; m < c->M; ++m) {
    const Raw& r = raw[m];
    const size_t row_bytes = (size_t)r.sw * r.ch * r.es;
    for (int f = 0; f < n_frames; ++f) {
      const lmx_image& im = sources[(size_t)f * c->M + m];
      tasks.push_back(Task{fs.h_raw + off + (size_t)f * r.bytes, (const uint8_t*)im.data, row_bytes, im.row_stride_bytes, r.sh});
    }
    off += r.bytes * n_frames;
  }
  auto run = [&](int i) {
    const Task& t = tasks[i];
    if (t.src_stride == t.row_bytes) stream_copy(t.dst, t.src, t.row_bytes * t.rows);
    else
      for (int y = 0; y < t.rows; ++y) stream_copy(t.dst + (size_t)y * t.row_bytes, t.src + (size_t)y * t.src_stride, t.row_bytes);
  };
  if (tasks.size() > 2) {
    if (!c->pool) c->pool.reset(new CopyPool(upload_threads(c) - 1));
    c->pool->parallel_for((int)tasks.size(), run);
  } else {
    for (int i = 0; i < (int)tasks.size(); ++i) run(i);
  }
  if (!c->pre_stream) {
    LMX_HIP(hipStreamCreateWithFlags(&c->pre_stream, hipStreamNonBlocking));
    LMX_HIP(hipEventCreateWithFlags(&c->raw_dma_done, hipEventDisableTiming));
  }
  off = 0;
  for (int m = 0; m < c->M; ++m) {
    LMX_HIP(hipMemcpyAsync(fs.d_raw + off, fs.h_raw + off, raw[m].bytes * n_frames, hipMemcpyHostToDevice, c->copy_stream));
    off += raw[m].bytes * n_frames;
  }
  // the kernels follow on their own stream: the copy stream is free for the next batch's transfer while they run.  (The copy stream waited
  // for the lanes that still read the set's frames before the transfer; the kernels, which overwrite those frames, come behind it.)
  LMX_HIP(hipEventRecord(c->raw_dma_done, c->copy_stream));
  LMX_HIP(hipStreamWaitEvent(c->pre_stream, c->raw_dma_done, 0));
  off = 0;
  for (int m = 0; m < c->M; ++m) {
    const Raw& r = raw[m];
    c->cur_stream = c->pre_stream;
    ScopedKernel k(c, K_PRE);
    if (c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT)
      launch_pre_color(c->pre_stream, fs.d_raw + off, fs.bgr[m], r.sh, r.sw, r.ch, H, W, r.cx, r.cy, pre->blur3 ? 1 : 0, n_frames);
    else
      launch_pre_depth(c->pre_stream, fs.d_raw + off, fs.depth[m], r.sh, r.sw, H, W, r.cx, r.cy, pre->depth_float_m ? 1 : 0, n_frames);
    off += r.bytes * n_frames;
  }
  LMX_HIP(hipGetLastError());
  fs.n_uploaded = n_frames;
  return end_set_upload(c, set, c->pre_stream);
  });
}

// The per-batch chain in two stages.  No host synchronisation and no allocation in either, so they can run eagerly or inside a
// stream capture (hipGraph).  Stage 1 (pre-processing): every level/modality -> quantised images, spread images, memories.
static lmx_status issue_pre(lmx_ctx* c, int32_t n_frames, hipStream_t s) {
  c->cur_stream = s;
  bool first = true;  // the chain's first kernel (level 0 of modality 0, whichever kind) also clears the output slot's header
  for (int l = 0; l < c->L; ++l) {
    const LevelGeom& g = c->kp.geom[l];
    for (int m = 0; m < c->M; ++m) {
      const lmx_modality_desc& md = c->bank->mods[m];
      if (md.type == LMX_MOD_COLOR_GRADIENT) {
        // the level-l kernel also writes the pyrDown'ed source of level l+1 (upstream: ColorGradientPyramid::pyrDown)
        ScopedKernel k(c, K_COLOR_QUANTIZE);
        launch_color_quantize(s, c->mb[m].bgr[l], c->kp.fb.quant[l][m], l + 1 < c->L ? c->mb[m].bgr[l + 1] : nullptr, g.H, g.W, n_frames,
                              md.weak_threshold, nullptr, first ? reinterpret_cast<uint32_t*>(c->d_out) : nullptr);
        first = false;
      } else {
        if (l == 0) {
          // also writes level 1's label image (a8: the quantised image is downsampled, not the depth)
          ScopedKernel k(c, K_DEPTH_QUANTIZE);
          launch_depth_quantize(s, c->mb[m].depth, c->kp.fb.quant[0][m], c->L > 1 ? c->kp.fb.quant[1][m] : nullptr, g.H, g.W, n_frames,
                                md.distance_threshold, md.difference_threshold, c->d_normal_bins, first ? reinterpret_cast<uint32_t*>(c->d_out) : nullptr);
          first = false;
        } else if (l == 1) {
          // done by the level-0 kernel
        } else {
          ScopedKernel k(c, K_NN_DOWN);
          launch_nn_down2(s, c->kp.fb.quant[l - 1][m], c->kp.fb.quant[l][m], g.H, g.W, n_frames);
        }
      }
    }
    // Detector::match(..., masks): labels outside a modality's mask are dropped before they are spread (upstream quantize(): copyTo(dst, mask))
    for (int m = 0; m < c->M; ++m)
      if (c->sets[c->cur_set].masked[m]) launch_apply_mask(s, c->kp.fb.quant[l][m], c->sets[c->cur_set].mask[m], g.H, g.W, c->desc.width, c->desc.height, l, n_frames);
    // spread + linearise of the level: all modalities in one launch when the level has a fast kernel
    SpreadBatch sb{};
    for (int m = 0; m < c->M; ++m) {
      sb.quant[m] = c->kp.fb.quant[l][m]; sb.lm[m] = c->kp.fb.lm[l][m]; sb.ls[m] = c->kp.fb.ls[l][m];
      sb.lmn[m] = l == c->L - 1 ? c->kp.fb.lmn[m] : nullptr;
    }
    bool batched;
    {
      ScopedKernel k(c, K_SPREAD_LINEARIZE);
      batched = launch_spread_linearize_all(s, sb, c->M, g, n_frames);
    }
    for (int m = 0; m < c->M; ++m) {
      if (!batched) {
        ScopedKernel k(c, K_SPREAD_LINEARIZE);
        launch_spread_linearize(s, sb.quant[m], sb.lm[m], sb.ls[m], sb.lmn[m], g, n_frames);
      }
      if (l == c->L - 1 && !spread_writes_nibbles(g)) {
        ScopedKernel k(c, K_PACK_NIBBLES);
        launch_pack_nibbles(s, c->kp.fb.lm[l][m], c->kp.fb.lmn[m], g, n_frames);
      }
    }
  }
  LMX_HIP(hipGetLastError());
  return LMX_OK;
}

// Stage 2 (matching): score, refine, queue the read-back (the slot header was cleared by the first kernel of stage 1).
static lmx_status issue_post(lmx_ctx* c, int slot, int32_t n_frames, float threshold, hipStream_t s) {
  c->cur_stream = s;
  {
    const uint8_t* lm_mod[kMaxModalities] = {nullptr, nullptr, nullptr, nullptr};
    for (int m = 0; m < c->M; ++m) lm_mod[m] = c->kp.fb.lmn[m];
    ScopedKernel k(c, K_SCORE_COARSE);
    launch_score_coarse(s, c->dbank, c->kp.geom[c->L - 1], lm_mod, n_frames, threshold, c->d_class_slot, c->d_cands, c->d_cand_count(), c->cap_total, c->stripes_for(n_frames));
  }
  bool published;
  {
    // the read-back of the header and a first slice of records is the last workgroup's job (k_refine's folded publish; collect() only
    // waits on the slot's event).  It is a kernel writing through the device mapping of the pinned slot, not a DMA copy: see
    // k_publish_records, which still serves shards without templates and the gather-block exports
    // Folded only for one or two frames: there a launch (~4 us) is a visible share of the call and a few dozen workgroups take a
    // ticket; at 64 frames ~2000 workgroups would each pay a release fence and an atomic on one address (measured: k_refine 0.021 ->
    // 0.075 ms per step, 138 k -> 122 k frames/s), far more than the launch they save.
    const bool fold = n_frames <= lmx_ctx::kStoreFrames;
    ScopedKernel k(c, K_REFINE);
    published = launch_refine(s, c->dbank, c->kp, n_frames, threshold, c->d_class_slot, c->d_cands, c->d_cand_count(), c->cap_total, c->stripes_for(n_frames), c->d_records(),
                              c->d_match_count(), fold ? c->h_out_dev[slot] : nullptr, c->d_out, c->d_pub_counter + slot,
                              (uint32_t)std::min<size_t>(c->h_out_records, lmx_ctx::kFirstSlice));
    published = published && fold;
  }
  if (!published) launch_publish_records(s, c->h_out_dev[slot], c->d_out, (uint32_t)std::min<size_t>(c->h_out_records, lmx_ctx::kFirstSlice), c->cap_total);
  LMX_HIP(hipGetLastError());
  return LMX_OK;
}

// One modality's frames written straight into the frame set's host-visible device buffers (see FrameSet::store_buf).
static void store_modality(lmx_ctx* c, lmx_ctx::FrameSet& fs, int m, int n_frames, const lmx_image* sources) {
  const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
  const size_t row_bytes = (size_t)c->desc.width * (cg ? 3 : 2);
  const int H = c->desc.height;
  // on the calling thread: one thread's non-temporal stores already fill the link's write direction (the microbenchmark: 45.7
  // GB/s with 1 thread, 44.5 with 8), and waking pool threads costs more than it could save
  for (int f = 0; f < n_frames; ++f) {
    const lmx_image& im = sources[(size_t)f * c->M + m];
    uint8_t* dst = fs.store_buf[m] + (size_t)f * c->frame_bytes[m];
    if (im.row_stride_bytes == row_bytes) stream_copy(dst, im.data, row_bytes * H);
    else
      for (int y = 0; y < H; ++y) stream_copy(dst + (size_t)y * row_bytes, (const uint8_t*)im.data + (size_t)y * im.row_stride_bytes, row_bytes);
  }
}

// The chain for one or two frames of the reference's own configuration (two pyramid levels; ColorGradient, or ColorGradient +
// DepthNormal): five launches instead of eight --
//   colour L0 | depth L0 + colour L1 | spread L0 + L1 | score | refine (+ read-back)
// -- and, when `sources` is given (lmx_match / lmx_match_batch with the direct-store upload), the frames are written between the
// launches: colour first, and the depth frames while the colour kernel of level 0 already runs.  Measured per call with a fresh
// 640x480 RGB-D host frame (3000 templates): see DESIGN.md section 6 / profiles/r03_single_frame_latency.txt.
static bool small_chain_ok(const lmx_ctx* c, int n_frames) {
  if (n_frames > lmx_ctx::kStoreFrames || c->L != 2 || c->M < 1 || c->M > 2) return false;
  if (c->bank->mods[0].type != LMX_MOD_COLOR_GRADIENT) return false;
  if (c->M == 2 && c->bank->mods[1].type != LMX_MOD_DEPTH_NORMAL) return false;
  return !c->env_no_small_chain;   // A/B switch (LMX_NO_SMALL_CHAIN, read when the context was created)
}

static lmx_status issue_small(lmx_ctx* c, int slot, int32_t n_frames, float threshold, hipStream_t s, lmx_ctx::FrameSet& fs, const lmx_image* sources) {
  c->cur_stream = s;
  const LevelGeom &g0 = c->kp.geom[0], &g1 = c->kp.geom[1];
  const lmx_modality_desc& cg = c->bank->mods[0];
  if (sources) store_modality(c, fs, 0, n_frames, sources);
  {
    ScopedKernel k(c, K_COLOR_QUANTIZE);
    launch_color_quantize(s, c->mb[0].bgr[0], c->kp.fb.quant[0][0], c->mb[0].bgr[1], g0.H, g0.W, n_frames, cg.weak_threshold, nullptr, reinterpret_cast<uint32_t*>(c->d_out));
  }
  if (c->M == 2) {
    if (sources) store_modality(c, fs, 1, n_frames, sources);   // lands while the colour kernel runs
    const lmx_modality_desc& dn = c->bank->mods[1];
    ScopedKernel k(c, K_DEPTH_QUANTIZE);
    launch_small_depth_color(s, c->mb[1].depth, c->kp.fb.quant[0][1], c->kp.fb.quant[1][1], g0.H, g0.W, dn.distance_threshold, dn.difference_threshold, c->d_normal_bins,
                             c->mb[0].bgr[1], c->kp.fb.quant[1][0], nullptr, g1.H, g1.W, cg.weak_threshold, n_frames);
  } else {
    ScopedKernel k(c, K_COLOR_QUANTIZE);
    launch_color_quantize(s, c->mb[0].bgr[1], c->kp.fb.quant[1][0], nullptr, g1.H, g1.W, n_frames, cg.weak_threshold, nullptr, nullptr);
  }
  SpreadBatch sb[2] = {};
  for (int l = 0; l < 2; ++l)
    for (int m = 0; m < c->M; ++m) {
      sb[l].quant[m] = c->kp.fb.quant[l][m]; sb[l].lm[m] = c->kp.fb.lm[l][m]; sb[l].ls[m] = c->kp.fb.ls[l][m];
      sb[l].lmn[m] = l == 1 ? c->kp.fb.lmn[m] : nullptr;
    }
  bool fused;
  {
    ScopedKernel k(c, K_SPREAD_LINEARIZE);
    fused = launch_small_spread(s, sb[0], g0, sb[1], g1, c->M, n_frames);
  }
  for (int l = 0; l < 2 && !fused; ++l) {   // no fused kernel for this pair of T / these widths: level by level, like issue_pre
    bool batched;
    {
      ScopedKernel k(c, K_SPREAD_LINEARIZE);
      batched = launch_spread_linearize_all(s, sb[l], c->M, c->kp.geom[l], n_frames);
    }
    for (int m = 0; m < c->M; ++m) {
      if (!batched) {
        ScopedKernel k(c, K_SPREAD_LINEARIZE);
        launch_spread_linearize(s, sb[l].quant[m], sb[l].lm[m], sb[l].ls[m], sb[l].lmn[m], c->kp.geom[l], n_frames);
      }
      if (l == 1 && !spread_writes_nibbles(g1)) {
        ScopedKernel k(c, K_PACK_NIBBLES);
        launch_pack_nibbles(s, c->kp.fb.lm[l][m], c->kp.fb.lmn[m], g1, n_frames);
      }
    }
  }
  LMX_HIP(hipGetLastError());
  return issue_post(c, slot, n_frames, threshold, s);
}

// Stream capture and other threads.  A device group drives its members from several host threads; the first enqueues of every
// member capture their chains at the same time, and on ROCm 7.2 a capture (thread-local mode) that overlaps another thread's capture
// or launches ends with "operation failed due to a previous error during capture".  Captures are rare (once per slot, frame set,
// batch size and threshold), so they simply run alone: every enqueue holds this lock shared, a capture holds it exclusively.
static std::shared_mutex g_capture_mutex;

// Stream capture of one stage (or of both, back to back) into an executable graph.
static lmx_status capture_graph(hipStream_t s, hipGraphExec_t* exec, const std::function<lmx_status()>& issue) {
  hipGraph_t graph = nullptr;
  LMX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  lmx_status st = issue();
  hipError_t e = hipStreamEndCapture(s, &graph);
  if (st != LMX_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
  if (e != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return LMX_ERR_HIP; }
  if (const char* dot = std::getenv("LMX_GRAPH_DOT")) {  // diagnostics: one .dot file per captured chain
    static int n_dot = 0;
    char path[512];
    snprintf(path, sizeof(path), "%s/lmx_graph_%d.dot", dot, n_dot++);
    (void)hipGraphDebugDotPrint(graph, path, hipGraphDebugDotFlagsVerbose);
  }
  LMX_HIP(hipGraphInstantiate(exec, graph, nullptr, nullptr, 0));
  (void)hipGraphDestroy(graph);
  return LMX_OK;
}

// The executable graph of the whole per-batch chain for (output slot, frame set, batch size, threshold); captured on first use.
// Expects the lane of `slot` selected and c->d_out pointing at the slot.
static lmx_status ensure_graph(lmx_ctx* c, int slot, int set, int32_t n_frames, float threshold, hipStream_t sa, hipGraphExec_t* out) {
  uint32_t tbits;
  std::memcpy(&tbits, &threshold, 4);
  for (const lmx_ctx::GraphEntry& ge : c->graphs)
    if (ge.slot == slot && ge.set == set && ge.n_frames == n_frames && ge.threshold_bits == tbits) { *out = ge.exec; return LMX_OK; }
  std::unique_lock<std::shared_mutex> capture_lock(g_capture_mutex);
  hipGraphExec_t exec = nullptr;
  lmx_status st = capture_graph(sa, &exec, [&]() {
    lmx_status r = issue_pre(c, n_frames, sa);
    return r != LMX_OK ? r : issue_post(c, slot, n_frames, threshold, sa);
  });
  if (st != LMX_OK) return st;
  if (c->graphs.size() >= 64) {
    if (sync_lanes(c) != LMX_OK) return LMX_ERR_HIP;  // the evicted graph may still be executing
    (void)hipGraphExecDestroy(c->graphs.front().exec);
    c->graphs.erase(c->graphs.begin());
  }
  c->graphs.push_back(lmx_ctx::GraphEntry{slot, set, n_frames, tbits, exec});
  *out = exec;
  return LMX_OK;
}

}  // extern "C"

// Device groups call this for every member from the calling thread before their host threads enqueue in parallel: the capture of a
// chain that is not cached yet then happens here, with no other thread of the group inside the HIP runtime (see g_capture_mutex).
lmx_status lmx::ctx_prepare_graph(lmx_ctx* c, int n_frames, float threshold) {
  if (!(c->desc.flags & LMX_CTX_HIPGRAPH) || c->profiling != 0) return LMX_OK;
  if (n_frames < 1 || n_frames > c->F || c->outstanding >= c->n_slots) return LMX_OK;   // the enqueue reports it
  for (int m = 0; m < c->M; ++m)
    if (c->sets[c->cur_set].masked[m]) return LMX_OK;   // masked batches take the plain chain (lmx_ctx_enqueue): a graph captured now would bake k_apply_mask in
  LMX_HIP(hipSetDevice(c->device));
  const int slot = c->head, lane = slot % c->n_lanes;
  select_lane(c, lane);
  c->d_out = c->d_out_slot[slot];
  hipGraphExec_t exec = nullptr;
  return ensure_graph(c, slot, c->cur_set, n_frames, threshold, c->lane_stream[lane], &exec);
}

extern "C" {

lmx_status lmx_ctx_enqueue(lmx_ctx* c, int32_t n_frames, float threshold, const char* const* class_ids, int32_t n_class_ids) {
  return lmx::guarded("lmx_ctx_enqueue", [&]() -> lmx_status {
  if (!c) { set_error("lmx_ctx_enqueue: null context"); return LMX_ERR_INVALID_ARG; }
  if (n_frames < 1 || n_frames > c->F) { set_error("n_frames=%d outside [1,%d]", n_frames, c->F); return LMX_ERR_INVALID_ARG; }
  std::shared_lock<std::shared_mutex> launch_lock(g_capture_mutex);
  LMX_HIP(hipSetDevice(c->device));
  // class filter -> insertion slot per class (upstream iterates the map when the filter is empty, else the list)
  std::vector<int32_t> slots(c->n_classes, -1);
  if (n_class_ids <= 0 || !class_ids) {
    for (int i = 0; i < c->n_classes; ++i) slots[i] = i;
  } else {
    int slot = 0;
    for (int i = 0; i < n_class_ids; ++i)
      for (int k = 0; k < c->n_classes; ++k)
        if (class_ids[i] && c->class_names[k] == class_ids[i] && slots[k] < 0) slots[k] = slot++;
  }
  if (slots != c->cur_slots && c->n_classes > 0) {
    if (sync_lanes(c) != LMX_OK) return LMX_ERR_HIP;
    LMX_HIP(hipMemcpy(c->d_class_slot, slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    c->cur_slots = slots;
  }
  if (c->outstanding >= c->n_slots) {
    set_error("lmx_ctx_enqueue: %d enqueues are already outstanding; collect one first", c->outstanding);
    return LMX_ERR_INVALID_ARG;
  }
  const int slot = c->head;
  const int lane = slot % c->n_lanes;
  select_lane(c, lane);
  hipStream_t sa = c->lane_stream[lane];
  c->d_out = c->d_out_slot[slot];
  c->last_slot = slot;
  // the chain starts behind the upload of the frame set it reads (queued on the copy stream), not behind other lanes' kernels
  const int set = c->cur_set;
  lmx_ctx::FrameSet& fset = c->sets[set];
  if (n_frames > fset.n_uploaded) {
    set_error("lmx_ctx_enqueue: n_frames=%d but the most recent upload holds %d frame(s); an enqueue reads the frames of the latest upload", n_frames, fset.n_uploaded);
    return LMX_ERR_INVALID_ARG;
  }
  if (fset.h2d_recorded) LMX_HIP(hipStreamWaitEvent(sa, fset.h2d_done, 0));
  // Buffer hazards: a lane's intermediates are rewritten by every enqueue on it, in stream order; outputs are per slot.
  bool masked = false;   // masks are rare: the batch then takes the plain chain (no graph, no fused small-batch launches)
  for (int m = 0; m < c->M; ++m) masked = masked || fset.masked[m];
  if ((c->desc.flags & LMX_CTX_HIPGRAPH) && c->profiling == 0 && !masked) {
    // the whole per-batch chain (memset, kernels, read-back) as ONE graph launch; captured once per (slot, n_frames, threshold)
    hipGraphExec_t exec = nullptr;
    launch_lock.unlock();
    lmx_status gst = ensure_graph(c, slot, set, n_frames, threshold, sa, &exec);
    if (gst != LMX_OK) return gst;
    launch_lock.lock();
    LMX_HIP(hipGraphLaunch(exec, sa));
    LMX_HIP(hipEventRecord(fset.read_done[lane], sa));   // a graph is one unit: the frames are free once it has finished
  } else if (small_chain_ok(c, n_frames) && !masked) {
    // one or two frames: five launches, the frames stored between them when this is lmx_match's deferred upload
    const lmx_image* src = c->deferred_frames == n_frames ? c->deferred_sources : nullptr;
    if (!src && c->deferred_sources)   // an enqueue for fewer frames than were handed over: store them all first
      for (int m = 0; m < c->M; ++m) store_modality(c, fset, m, c->deferred_frames, c->deferred_sources);
    c->deferred_sources = nullptr; c->deferred_frames = 0;
    lmx_status st = issue_small(c, slot, n_frames, threshold, sa, fset, src);
    if (st != LMX_OK) return st;
    // recorded behind the whole chain: an event between two kernels of one stream costs a 5-6 us bubble, a third of what a
    // kernel of this chain takes, and nothing waits to overwrite the set of a one-frame call
    LMX_HIP(hipEventRecord(fset.read_done[lane], sa));
  } else {
    if (c->deferred_sources) {
      for (int m = 0; m < c->M; ++m) store_modality(c, fset, m, c->deferred_frames, c->deferred_sources);
      c->deferred_sources = nullptr; c->deferred_frames = 0;
    }
    lmx_status st = issue_pre(c, n_frames, sa);
    // the level-0 quantisers are the only readers of the uploaded frames: the set may be overwritten from here on
    if (st == LMX_OK) LMX_HIP(hipEventRecord(fset.read_done[lane], sa));
    if (st == LMX_OK) st = issue_post(c, slot, n_frames, threshold, sa);
    if (st != LMX_OK) return st;
  }
  fset.read_recorded[lane] = true;
  hipStream_t s = sa;
  LMX_HIP(hipEventRecord(c->done[slot], s));
  c->last_threshold = threshold;
  c->slot_frames[slot] = n_frames;
  c->head = (slot + 1) % c->n_slots;
  c->outstanding += 1;
  return LMX_OK;
  });
}

// sync + read-back + per-frame finalisation shared by collect / collect_flat
static lmx_status collect_impl(lmx_ctx* c, int32_t n_frames, std::vector<std::vector<HostMatch>>& fin) {
  if (c->outstanding < 1) { set_error("lmx_ctx_collect: nothing enqueued"); return LMX_ERR_INVALID_ARG; }
  const int slot = (c->head + c->n_slots - c->outstanding) % c->n_slots;  // oldest outstanding enqueue
  if (n_frames != c->slot_frames[slot]) { set_error("lmx_ctx_collect: n_frames=%d but the enqueue had %d", n_frames, c->slot_frames[slot]); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  const size_t first = std::min<size_t>(c->h_out_records, lmx_ctx::kFirstSlice);
  using clk = std::chrono::steady_clock;
  const clk::time_point t0 = clk::now();
  LMX_HIP(hipEventSynchronize(c->done[slot]));
  const clk::time_point t1 = clk::now();
  c->h_out = c->h_out_slot[slot];
  uint8_t* const d_slot = c->d_out_slot[slot];
  c->outstanding -= 1;
  if (c->outstanding == 0) drain_profiling(c);  // every recorded event has completed
  const uint32_t n_cand = reinterpret_cast<uint32_t*>(c->h_out)[0];
  const uint32_t n_match = reinterpret_cast<uint32_t*>(c->h_out)[1];
  if (c->env_debug_collect) {
    // diagnostics: the device-side slot against its pinned host mirror once the slot's event has completed
    uint32_t dev[16];
    if (hipMemcpy(dev, d_slot, 64, hipMemcpyDeviceToHost) == hipSuccess && (dev[0] != n_cand || dev[1] != n_match))
      fprintf(stderr, "LMX_DEBUG_COLLECT: slot %d host mirror {cand %u, match %u} != device {cand %u, match %u}\n", slot, n_cand, n_match, dev[0], dev[1]);
  }
  c->stat_cands = n_cand; c->stat_matches = n_match;
  if (n_cand > c->cap_total || n_match > c->cap_total) {
    set_error("candidate list overflow: %u candidates / %u matches > capacity %u; raise lmx_ctx_desc.max_candidates", n_cand, n_match, c->cap_total);
    return LMX_ERR_OVERFLOW;
  }
  if (n_match > first) {
    // rare: more matches than the first slice; the slot's records are final (its event has completed)
    LMX_HIP(hipMemcpy(c->h_out + 64 + first * sizeof(lmx_raw_match_t), d_slot + 64 + first * sizeof(lmx_raw_match_t),
                      (n_match - first) * sizeof(lmx_raw_match_t), hipMemcpyDeviceToHost));
  }
  const clk::time_point t2 = clk::now();
  const lmx_raw_match_t* recs = reinterpret_cast<const lmx_raw_match_t*>(c->h_out + 64);
  std::vector<std::vector<const lmx_raw_match_t*>> per_frame(n_frames);
  for (uint32_t i = 0; i < n_match; ++i) {
    const int f = recs[i].frame;
    if (f >= 0 && f < n_frames) per_frame[f].push_back(&recs[i]);
  }
  const clk::time_point t3 = clk::now();
  fin.resize(n_frames);
  // frames are independent; worth the upload threads only in the explosive regime (threshold 50: 10^5 records per frame, where the two
  // sorts of a frame take tens of milliseconds: DESIGN.md section 8), never at the reference's thresholds
  if (n_match > (1u << 15) && n_frames > 1) {
    if (!c->pool) c->pool.reset(new CopyPool(upload_threads(c) - 1));
    c->pool->parallel_for(n_frames, [&](int f) { finalize_frame(per_frame[f], fin[f]); });
  } else {
    for (int f = 0; f < n_frames; ++f) finalize_frame(per_frame[f], fin[f]);
  }
  if (c->trace_collect) {
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    size_t n_final = 0;
    for (int f = 0; f < n_frames; ++f) n_final += fin[f].size();
    fprintf(stderr, "lmx collect: %d frames, %u candidates, %u records -> %zu matches | wait %.1f us, fetch beyond first slice %.1f, group %.1f, order + std::sort + std::unique %.1f\n",
            n_frames, n_cand, n_match, n_final, us(t0, t1), us(t1, t2), us(t2, t3), us(t3, clk::now()));
  }
  return LMX_OK;
}

lmx_status lmx_ctx_collect(lmx_ctx* c, int32_t n_frames, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_ctx_collect", [&]() -> lmx_status {
  if (!c || !n_out || (cap > 0 && !out)) { set_error("lmx_ctx_collect: null argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<std::vector<HostMatch>> fin;
  lmx_status st = collect_impl(c, n_frames, fin);
  if (st != LMX_OK) { for (int f = 0; f < n_frames; ++f) n_out[f] = 0; return st; }
  for (int f = 0; f < n_frames; ++f) {
    n_out[f] = fin[f].size();
    const size_t n = std::min(cap, fin[f].size());
    for (size_t i = 0; i < n; ++i) out[(size_t)f * cap + i] = fin[f][i].m;
    if (fin[f].size() > cap) { set_error("frame %d: %zu matches > output capacity %zu", f, fin[f].size(), cap); st = LMX_ERR_OVERFLOW; }
  }
  return st;
  });
}

lmx_status lmx_ctx_collect_flat(lmx_ctx* c, int32_t n_frames, lmx_match_t* out, size_t cap_total, size_t* offsets) {
  return lmx::guarded("lmx_ctx_collect_flat", [&]() -> lmx_status {
  if (!c || !offsets || (cap_total > 0 && !out)) { set_error("lmx_ctx_collect_flat: null argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<std::vector<HostMatch>> fin;
  lmx_status st = collect_impl(c, n_frames, fin);
  if (st != LMX_OK) { for (int f = 0; f <= n_frames; ++f) offsets[f] = 0; return st; }
  size_t pos = 0;
  offsets[0] = 0;
  for (int f = 0; f < n_frames; ++f) {
    for (size_t i = 0; i < fin[f].size(); ++i, ++pos)
      if (pos < cap_total) out[pos] = fin[f][i].m;
    offsets[f + 1] = pos;
  }
  if (pos > cap_total) { set_error("%zu matches > output capacity %zu", pos, cap_total); return LMX_ERR_OVERFLOW; }
  return LMX_OK;
  });
}

// The depth ring of a vote is `(int)((dist - renderer_radius_min) / renderer_radius_step)` in float, as the reference computes it
// (src/rgbdDetector.cpp:48-56).  A step that is not positive and finite, a distance that is not finite or a quotient an int cannot hold
// make that conversion undefined in the reference and different between x86 and the GPU here: refused up front for the whole side-car.
static lmx_status check_vote_rings(const double* dists, size_t n, const lmx_cluster_params* pp) {
  const float step = (float)pp->renderer_radius_step;
  if (!(step > 0.0f) || !std::isfinite(step) || !std::isfinite((float)pp->renderer_radius_min)) {
    set_error("renderer_radius_step must be positive and finite, renderer_radius_min finite");
    return LMX_ERR_INVALID_ARG;
  }
  for (size_t i = 0; i < n; ++i) {
    const float q = ((float)dists[i] - pp->renderer_radius_min) / step;
    if (!(q > -1.0e9f && q < 1.0e9f)) { set_error("template %zu: origin distance %g gives no usable depth ring", i, dists[i]); return LMX_ERR_INVALID_ARG; }
  }
  return LMX_OK;
}

lmx_status lmx_ctx_set_cluster_sidecar(lmx_ctx* c, const double* obj_origin_dists, const int32_t* rects, size_t n_templates, const lmx_cluster_params* params) {
  if (!c || !obj_origin_dists || !rects || !params || n_templates == 0) { set_error("lmx_ctx_set_cluster_sidecar: invalid argument"); return LMX_ERR_INVALID_ARG; }
  if (params->vote_row_col_step <= 0) { set_error("vote_row_col_step must be positive"); return LMX_ERR_INVALID_ARG; }
  // the reference compares `size() <= thresh` with the int converted to size_t (src/rgbdDetector.cpp:72-85): a negative threshold would drop
  // every cluster there, and its erase-while-iterating is undefined anyway; refused so that the host and device chains cannot diverge
  if (params->cluster_size_thresh < 0) { set_error("cluster_size_thresh must not be negative"); return LMX_ERR_INVALID_ARG; }
  if (lmx_status vs = check_vote_rings(obj_origin_dists, n_templates, params)) return vs;
  LMX_HIP(hipSetDevice(c->device));
  if (sync_lanes(c) != LMX_OK) return LMX_ERR_HIP;   // a kernel may still read the previous side-car
  if (c->d_f2_dists) (void)hipFree(c->d_f2_dists);
  if (c->d_f2_rects) (void)hipFree(c->d_f2_rects);
  c->d_f2_dists = nullptr; c->d_f2_rects = nullptr; c->f2_sidecar = false;
  LMX_HIP(hipMalloc((void**)&c->d_f2_dists, n_templates * sizeof(double)));
  LMX_HIP(hipMalloc((void**)&c->d_f2_rects, n_templates * 4 * sizeof(int32_t)));
  LMX_HIP(hipMemcpy(c->d_f2_dists, obj_origin_dists, n_templates * sizeof(double), hipMemcpyHostToDevice));
  LMX_HIP(hipMemcpy(c->d_f2_rects, rects, n_templates * 4 * sizeof(int32_t), hipMemcpyHostToDevice));
  c->f2_templates = n_templates; c->f2_params = *params; c->f2_sidecar = true;
  c->f2_host_dists.assign(obj_origin_dists, obj_origin_dists + n_templates);
  c->f2_host_rects.assign(rects, rects + n_templates * 4);
  return LMX_OK;
}

lmx_status lmx_ctx_collect_clusters(lmx_ctx* c, int32_t n_frames, lmx_match_t* matches, size_t cap_matches, size_t* match_offsets, lmx_cluster_t* clusters,
                                    size_t cap_clusters, size_t* cluster_offsets, int32_t* members, size_t cap_members) {
  return lmx::guarded("lmx_ctx_collect_clusters", [&]() -> lmx_status {
  if (!c || !match_offsets || !cluster_offsets || (cap_matches > 0 && !matches) || (cap_clusters > 0 && !clusters) || (cap_members > 0 && !members)) {
    set_error("lmx_ctx_collect_clusters: null argument");
    return LMX_ERR_INVALID_ARG;
  }
  if (!c->f2_sidecar) { set_error("lmx_ctx_collect_clusters: call lmx_ctx_set_cluster_sidecar first"); return LMX_ERR_INVALID_ARG; }
  if (c->outstanding < 1) { set_error("lmx_ctx_collect_clusters: nothing enqueued"); return LMX_ERR_INVALID_ARG; }
  const int slot = (c->head + c->n_slots - c->outstanding) % c->n_slots;  // oldest outstanding enqueue
  if (n_frames != c->slot_frames[slot]) { set_error("lmx_ctx_collect_clusters: n_frames=%d but the enqueue had %d", n_frames, c->slot_frames[slot]); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  for (int f = 0; f <= n_frames; ++f) match_offsets[f] = cluster_offsets[f] = 0;
  const size_t F = (size_t)c->F;
  const size_t off_counts = F * F2_MAX * sizeof(lmx_match_t), off_clusters = off_counts + ((F * 4 * sizeof(uint32_t) + 63) & ~(size_t)63),
               off_members = off_clusters + F * F2_MAX * sizeof(lmx_cluster_t), out_bytes = off_members + F * F2_MAX * sizeof(int32_t);
  if (!c->h_f2_out) {
    lmx_status st;
    uint8_t* dv = nullptr;
    LMX_HIP(hipHostMalloc((void**)&c->h_f2_out, out_bytes, hipHostMallocMapped));
    LMX_HIP(hipHostGetDevicePointer((void**)&dv, c->h_f2_out, 0));
    std::memset(c->h_f2_out + off_counts, 0, F * 4 * sizeof(uint32_t));
    c->d_f2_matches = reinterpret_cast<lmx_match_t*>(dv); c->d_f2_counts = reinterpret_cast<uint32_t*>(dv + off_counts);
    c->d_f2_clusters = reinterpret_cast<lmx_cluster_t*>(dv + off_clusters); c->d_f2_members = reinterpret_cast<int32_t*>(dv + off_members);
    if ((st = dev_alloc(c, &c->d_f2_scratch, F * F2_MAX * 32, false)) != LMX_OK) return st;
    LMX_HIP(hipStreamCreateWithFlags(&c->f2_stream, hipStreamNonBlocking));
    LMX_HIP(hipStreamSynchronize(c->stream));
  }
  LMX_HIP(hipEventSynchronize(c->done[slot]));
  c->outstanding -= 1;
  if (c->outstanding == 0) drain_profiling(c);
  const uint32_t* h_hdr = reinterpret_cast<const uint32_t*>(c->h_out_slot[slot]);
  const uint32_t n_cand = h_hdr[0], n_match = h_hdr[1];
  c->stat_cands = n_cand; c->stat_matches = n_match;
  if (n_cand > c->cap_total || n_match > c->cap_total) {
    set_error("candidate list overflow: %u candidates / %u matches > capacity %u; raise lmx_ctx_desc.max_candidates", n_cand, n_match, c->cap_total);
    return LMX_ERR_OVERFLOW;
  }
  // on its own stream: the slot's kernels have finished (its event was waited for above), and the lane's stream may already carry later
  // batches that this collect must not wait for
  hipStream_t s = c->f2_stream;
  F2Params p{};
  p.recs = reinterpret_cast<const lmx_raw_match_t*>(c->d_out_slot[slot] + 64);
  p.hdr = reinterpret_cast<const uint32_t*>(c->d_out_slot[slot]);
  p.cap = c->cap_total; p.n_frames = n_frames;
  p.out_matches = c->d_f2_matches; p.out_counts = c->d_f2_counts; p.out_clusters = c->d_f2_clusters; p.out_members = c->d_f2_members; p.scratch = c->d_f2_scratch;
  p.dists = c->d_f2_dists; p.rects = c->d_f2_rects; p.n_templates = (uint32_t)c->f2_templates;
  p.step = c->f2_params.vote_row_col_step; p.size_thresh = c->f2_params.cluster_size_thresh; p.do_clusters = 1;
  p.radius_min = c->f2_params.renderer_radius_min; p.radius_step = c->f2_params.renderer_radius_step;
  launch_f2(s, p);
  LMX_HIP(hipGetLastError());
  LMX_HIP(hipStreamSynchronize(s));
  const uint32_t* counts = reinterpret_cast<const uint32_t*>(c->h_f2_out + off_counts);
  const lmx_match_t* all_m = reinterpret_cast<const lmx_match_t*>(c->h_f2_out);
  const lmx_cluster_t* all_c = reinterpret_cast<const lmx_cluster_t*>(c->h_f2_out + off_clusters);
  const int32_t* all_mem = reinterpret_cast<const int32_t*>(c->h_f2_out + off_members);
  // frames the device could not take (too many records, bins outside the packed range): the host path on the slot's records
  std::vector<lmx_raw_match_t> host_recs;
  bool any_host = false;
  for (int f = 0; f < n_frames; ++f) any_host = any_host || counts[(size_t)f * 4 + 3] != 0;
  if (any_host) {
    host_recs.resize(n_match);
    if (n_match) LMX_HIP(hipMemcpy(host_recs.data(), c->d_out_slot[slot] + 64, (size_t)n_match * sizeof(lmx_raw_match_t), hipMemcpyDeviceToHost));
  }
  lmx_status st = LMX_OK;
  size_t mpos = 0, cpos = 0, mempos = 0;
  std::vector<HostMatch> fin;
  std::vector<lmx_match_t> fm;
  std::vector<lmx_cluster_t> fc;
  std::vector<int32_t> fmem;
  for (int f = 0; f < n_frames; ++f) {
    size_t nm = 0, nc = 0, nmem = 0;
    if (counts[(size_t)f * 4 + 3] == 0) {
      nm = counts[(size_t)f * 4 + 0]; nc = counts[(size_t)f * 4 + 1]; nmem = counts[(size_t)f * 4 + 2];
      if (cap_matches) fm.assign(all_m + (size_t)F2_MAX * f, all_m + (size_t)F2_MAX * f + nm);
      else fm.clear();
      fc.assign(all_c + (size_t)F2_MAX * f, all_c + (size_t)F2_MAX * f + nc);
      fmem.assign(all_mem + (size_t)F2_MAX * f, all_mem + (size_t)F2_MAX * f + nmem);
    } else {
      std::vector<const lmx_raw_match_t*> recs;
      for (const lmx_raw_match_t& r : host_recs)
        if (r.frame == f) recs.push_back(&r);
      finalize_frame(recs, fin);
      nm = fin.size();
      fm.resize(nm);
      for (size_t i = 0; i < nm; ++i) fm[i] = fin[i].m;
      fc.resize(std::max<size_t>(nm, 1)); fmem.resize(std::max<size_t>(nm, 1));
      size_t got = 0;
      lmx_status hs = lmx_cluster_matches(fm.data(), nm, c->f2_host_dists.data(), c->f2_host_rects.data(), c->f2_templates, &c->f2_params, fc.data(), fc.size(), &got,
                                          fmem.data(), fmem.size());
      if (hs != LMX_OK) return hs;
      nc = got; nmem = 0;
      for (size_t i = 0; i < nc; ++i) nmem += (size_t)fc[i].member_count;
    }
    if (cap_matches) {
      if (mpos + nm > cap_matches) st = LMX_ERR_OVERFLOW;
      else if (nm) std::memcpy(matches + mpos, fm.data(), nm * sizeof(lmx_match_t));   // (an empty vector's data() may be null)
    }
    if (cpos + nc <= cap_clusters && mempos + nmem <= cap_members) {
      for (size_t i = 0; i < nc; ++i) { clusters[cpos + i] = fc[i]; clusters[cpos + i].member_begin += (int32_t)mempos; }
      if (nmem) std::memcpy(members + mempos, fmem.data(), nmem * sizeof(int32_t));
    } else {
      st = LMX_ERR_OVERFLOW;
    }
    mpos += nm; cpos += nc; mempos += nmem;
    match_offsets[f + 1] = mpos; cluster_offsets[f + 1] = cpos;
  }
  if (st != LMX_OK) set_error("%zu matches / %zu clusters / %zu members exceed the output capacity", mpos, cpos, mempos);
  return st;
  });
}

lmx_status lmx_match_batch(lmx_ctx* c, int32_t n_frames, const lmx_image* sources, int32_t n_sources, float threshold,
                           const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_match_batch", [&]() -> lmx_status {
  if (!c) { set_error("lmx_match: null context"); return LMX_ERR_INVALID_ARG; }
  std::lock_guard<std::recursive_mutex> lk(c->call_mutex);   // contexts handed out by lmx_ctx_acquire may be shared between threads
  c->deferred_sources = nullptr;
  c->deferred_frames = -1;   // "upload may leave the direct stores of a small batch to the enqueue below" (the sources outlive both calls)
  lmx_status st = lmx_ctx_upload(c, n_frames, sources, n_sources);
  if (c->deferred_frames == -1) c->deferred_frames = 0;
  if (st == LMX_OK) st = lmx_ctx_enqueue(c, n_frames, threshold, class_ids, n_class_ids);
  if (c->deferred_sources) {   // the enqueue failed before it consumed them: the set must still hold what upload promised
    lmx_ctx::FrameSet& fs = c->sets[c->cur_set];
    for (int m = 0; m < c->M; ++m) store_modality(c, fs, m, c->deferred_frames, c->deferred_sources);
    c->deferred_sources = nullptr;
  }
  c->deferred_frames = 0;
  if (st != LMX_OK) return st;
  return lmx_ctx_collect(c, n_frames, out, cap, n_out);
  });
}

void lmx_ctx_lock(lmx_ctx* c) { if (c) c->call_mutex.lock(); }
void lmx_ctx_unlock(lmx_ctx* c) { if (c) c->call_mutex.unlock(); }

lmx_status lmx_match(lmx_ctx* c, const lmx_image* sources, int32_t n_sources, float threshold, const char* const* class_ids,
                     int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_match", [&]() -> lmx_status {
  return lmx_match_batch(c, 1, sources, n_sources, threshold, class_ids, n_class_ids, out, cap, n_out);
  });
}

lmx_status lmx_ctx_raw_matches(lmx_ctx* c, void** d_records, void** d_counts, size_t* capacity) {
  if (!c) { set_error("lmx_ctx_raw_matches: null context"); return LMX_ERR_INVALID_ARG; }
  if (d_records) *d_records = c->d_records();
  if (d_counts) *d_counts = c->d_out;  // uint32[16] header: [0] = candidates, [1] = matches
  if (capacity) *capacity = c->cap_total;
  return LMX_OK;
}

lmx_status lmx_ctx_export_raw_on(lmx_ctx* c, void* d_block, size_t capacity_records, void* stream) {
  if (!c || !d_block) { set_error("lmx_ctx_export_raw: null argument"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  const size_t n = std::min<size_t>(capacity_records, c->cap_total);
  // d_out already has the gather-block layout: [64-byte header][records].  The copy is ordered behind the enqueue that
  // produced the records, whichever lane it ran on
  LMX_HIP(hipStreamWaitEvent(s, c->done[c->last_slot], 0));
  // header + as many records as it counts (<= n), by kernel (see k_publish_records); the rest of the block is don't-care
  launch_publish_records(s, d_block, c->d_out, (uint32_t)n, c->cap_total);
  LMX_HIP(hipGetLastError());
  return LMX_OK;
}

lmx_status lmx_ctx_export_oldest_on(lmx_ctx* c, void* d_block, size_t capacity_records, void* stream) {
  if (!c || !d_block) { set_error("lmx_ctx_export_oldest_on: null argument"); return LMX_ERR_INVALID_ARG; }
  if (c->outstanding < 1) { set_error("lmx_ctx_export_oldest_on: nothing enqueued"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  const int slot = (c->head + c->n_slots - c->outstanding) % c->n_slots;
  LMX_HIP(hipStreamWaitEvent(s, c->done[slot], 0));
  launch_publish_records(s, d_block, c->d_out_slot[slot], (uint32_t)std::min<size_t>(capacity_records, c->cap_total), c->cap_total);
  LMX_HIP(hipGetLastError());
  return LMX_OK;
}

lmx_status lmx_ctx_export_raw(lmx_ctx* c, void* d_block, size_t capacity_records) {
  return lmx_ctx_export_raw_on(c, d_block, capacity_records, c ? (void*)c->stream : nullptr);
}

int32_t lmx_ctx_max_outstanding(const lmx_ctx* c) { return c ? c->n_slots : 0; }

lmx_status lmx_ctx_release(lmx_ctx* c) {
  if (!c) { set_error("lmx_ctx_release: null context"); return LMX_ERR_INVALID_ARG; }
  if (c->outstanding < 1) { set_error("lmx_ctx_release: nothing enqueued"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  const int slot = (c->head + c->n_slots - c->outstanding) % c->n_slots;
  LMX_HIP(hipEventSynchronize(c->done[slot]));
  c->outstanding -= 1;
  if (c->outstanding == 0) drain_profiling(c);
  return LMX_OK;
}

lmx_status lmx_stream_copy(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src) { set_error("lmx_stream_copy: null argument"); return LMX_ERR_INVALID_ARG; }
  if (((uintptr_t)dst | (uintptr_t)src | bytes) & 15u) { set_error("lmx_stream_copy: pointers and size must be multiples of 16 bytes"); return LMX_ERR_INVALID_ARG; }
  launch_copy_bytes((hipStream_t)stream, dst, src, bytes);
  LMX_HIP(hipGetLastError());
  return LMX_OK;
}

lmx_status lmx_stream_copy_blocks(void* dst, const void* src, int32_t n_blocks, size_t block_stride_bytes, size_t capacity_records, void* stream) {
  if (!dst || !src || n_blocks < 1) { set_error("lmx_stream_copy_blocks: invalid argument"); return LMX_ERR_INVALID_ARG; }
  if ((((uintptr_t)dst | (uintptr_t)src | block_stride_bytes) & 15u) || block_stride_bytes < LMX_GATHER_HEADER_BYTES + capacity_records * sizeof(lmx_raw_match_t)) {
    set_error("lmx_stream_copy_blocks: pointers and stride must be multiples of 16 bytes and a block must hold its capacity");
    return LMX_ERR_INVALID_ARG;
  }
  launch_publish_blocks((hipStream_t)stream, dst, src, n_blocks, block_stride_bytes, (uint32_t)std::min<size_t>(capacity_records, 0xffffffffu));
  LMX_HIP(hipGetLastError());
  return LMX_OK;
}

lmx_status lmx_merge_gathered(const void* blocks, int32_t n_ranks, size_t block_stride_bytes, size_t capacity_records, int32_t n_frames,
                              lmx_match_t* out, size_t cap_total, size_t* offsets) {
  return lmx::guarded("lmx_merge_gathered", [&]() -> lmx_status {
  if (!blocks || !offsets || n_ranks < 1 || n_frames < 1 || (cap_total > 0 && !out)) { set_error("lmx_merge_gathered: invalid argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<std::vector<const lmx_raw_match_t*>> per_frame(n_frames);
  for (int r = 0; r < n_ranks; ++r) {
    const uint8_t* blk = (const uint8_t*)blocks + (size_t)r * block_stride_bytes;
    const uint32_t n = reinterpret_cast<const uint32_t*>(blk)[1];
    const uint32_t n_cand = reinterpret_cast<const uint32_t*>(blk)[0], cand_cap = reinterpret_cast<const uint32_t*>(blk)[2];
    if (cand_cap != 0 && n_cand > cand_cap) {
      // the rank's scoring kernel dropped candidates (which ones is not deterministic): its matches are incomplete
      for (int f = 0; f <= n_frames; ++f) offsets[f] = 0;
      set_error("rank %d: candidate list overflow (%u candidates > capacity %u); raise lmx_ctx_desc.max_candidates", r, n_cand, cand_cap);
      return LMX_ERR_OVERFLOW;
    }
    if (n > capacity_records) {
      for (int f = 0; f <= n_frames; ++f) offsets[f] = 0;
      set_error("rank %d wrote %u records > gather capacity %zu", r, n, capacity_records);
      return LMX_ERR_OVERFLOW;
    }
    const lmx_raw_match_t* recs = reinterpret_cast<const lmx_raw_match_t*>(blk + LMX_GATHER_HEADER_BYTES);
    for (uint32_t i = 0; i < n; ++i)
      if (recs[i].frame >= 0 && recs[i].frame < n_frames) per_frame[recs[i].frame].push_back(&recs[i]);
  }
  size_t pos = 0;
  offsets[0] = 0;
  std::vector<HostMatch> fin;
  for (int f = 0; f < n_frames; ++f) {
    finalize_frame(per_frame[f], fin);
    for (size_t i = 0; i < fin.size(); ++i, ++pos)
      if (pos < cap_total) out[pos] = fin[i].m;
    offsets[f + 1] = pos;
  }
  if (pos > cap_total) { set_error("%zu matches > output capacity %zu", pos, cap_total); return LMX_ERR_OVERFLOW; }
  return LMX_OK;
  });
}

lmx_status lmx_ctx_sync(lmx_ctx* c) {
  if (!c) { set_error("lmx_ctx_sync: null context"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  if (sync_lanes(c) != LMX_OK) return LMX_ERR_HIP;
  drain_profiling(c);
  c->outstanding = 0;  // abandons enqueues that were not collected (their results stay readable via export_raw)
  return LMX_OK;
}

lmx_status lmx_merge_raw(const lmx_raw_match_t* records, size_t n_records, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_merge_raw", [&]() -> lmx_status {
  if ((n_records > 0 && !records) || !n_out || (cap > 0 && !out)) { set_error("lmx_merge_raw: null argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<const lmx_raw_match_t*> recs(n_records);
  for (size_t i = 0; i < n_records; ++i) recs[i] = &records[i];
  std::vector<HostMatch> fin;
  finalize_frame(recs, fin);
  *n_out = fin.size();
  const size_t n = std::min(cap, fin.size());
  for (size_t i = 0; i < n; ++i) out[i] = fin[i].m;
  if (fin.size() > cap) { set_error("%zu matches > output capacity %zu", fin.size(), cap); return LMX_ERR_OVERFLOW; }
  return LMX_OK;
  });
}

lmx_status lmx_ctx_debug_read(lmx_ctx* c, int32_t frame, int32_t what, int32_t level, int32_t modality, void* out, size_t out_bytes) {
  if (!c || !out) { set_error("lmx_ctx_debug_read: null argument"); return LMX_ERR_INVALID_ARG; }
  if (frame < 0 || frame >= c->F || level < 0 || level >= c->L || modality < 0 || modality >= c->M) { set_error("debug_read: index out of range"); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  if (sync_lanes(c) != LMX_OK) return LMX_ERR_HIP;
  const LevelGeom& g = c->kp.geom[level];
  if (what == LMX_DBG_QUANTIZED) {
    const size_t n = (size_t)g.W * g.H;
    if (out_bytes < n) { set_error("debug_read: buffer too small"); return LMX_ERR_INVALID_ARG; }
    LMX_HIP(hipMemcpy(out, c->kp.fb.quant[level][modality] + (size_t)frame * n, n, hipMemcpyDeviceToHost));
  } else if (what == LMX_DBG_LINEAR_MEMORY) {
    const size_t n = (size_t)g.T * g.T * g.cells;
    if (out_bytes < 8 * n) { set_error("debug_read: buffer too small"); return LMX_ERR_INVALID_ARG; }
    if (level == c->L - 1) {
      // the coarsest level lives nibble-packed on the device; unpack to upstream's byte-wide linear memories
      const size_t nb = (n + 1) / 2;
      std::vector<uint8_t> nib(nb);
      for (int o = 0; o < 8; ++o) {
        LMX_HIP(hipMemcpy(nib.data(), c->kp.fb.lmn[modality] + (size_t)frame * g.nib_mod_stride + (size_t)o * g.nib_ori_stride, nb,
                          hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) ((uint8_t*)out)[o * n + i] = (uint8_t)((nib[i >> 1] >> (4 * (i & 1))) & 0xf);
      }
    } else {
      // finer levels hold the linearised spread image only; expand it to upstream's eight linear memories for the caller
      static const uint32_t masks[8] = {0x0103070fu, 0x02070f1fu, 0x040e1f3fu, 0x081c3e7fu, 0x10387cfeu, 0x2070f8fdu, 0x40e0f1fbu, 0x80c1e3f7u};
      std::vector<uint8_t> sp(n);
      if (g.ls_bands) {   // banded image: take the first 16 columns of every band row
        std::vector<uint8_t> banded(g.ls_stride);
        LMX_HIP(hipMemcpy(banded.data(), c->kp.fb.ls[level][modality] + (size_t)frame * g.ls_stride, g.ls_stride, hipMemcpyDeviceToHost));
        const size_t rows = n / g.Wc;
        for (size_t r = 0; r < rows; ++r)
          for (uint32_t k = 0; k < g.ls_bands; ++k)
            memcpy(&sp[r * g.Wc + 16 * k], &banded[(size_t)k * g.ls_band_stride + (r + 1) * 32], 16);
      } else
        LMX_HIP(hipMemcpy(sp.data(), c->kp.fb.ls[level][modality] + (size_t)frame * g.ls_stride, n, hipMemcpyDeviceToHost));
      for (int o = 0; o < 8; ++o)
        for (size_t i = 0; i < n; ++i) {
          int r = 0;
          for (int k = 0; k < 4; ++k) r += (sp[i] & ((masks[o] >> (8 * k)) & 0xffu)) != 0;
          ((uint8_t*)out)[o * n + i] = (uint8_t)r;
        }
    }
  } else if (what == LMX_DBG_PYRAMID_BGR) {
    if (c->bank->mods[modality].type != LMX_MOD_COLOR_GRADIENT) { set_error("debug_read: modality %d has no colour pyramid", modality); return LMX_ERR_INVALID_ARG; }
    const size_t n = (size_t)g.W * g.H * 3;
    if (out_bytes < n) { set_error("debug_read: buffer too small"); return LMX_ERR_INVALID_ARG; }
    LMX_HIP(hipMemcpy(out, c->mb[modality].bgr[level] + (size_t)frame * n, n, hipMemcpyDeviceToHost));
  } else if (what == LMX_DBG_DEPTH) {
    if (c->bank->mods[modality].type != LMX_MOD_DEPTH_NORMAL) { set_error("debug_read: modality %d has no depth source", modality); return LMX_ERR_INVALID_ARG; }
    const size_t n = (size_t)c->desc.width * c->desc.height * 2;
    if (out_bytes < n) { set_error("debug_read: buffer too small"); return LMX_ERR_INVALID_ARG; }
    LMX_HIP(hipMemcpy(out, (const uint8_t*)c->mb[modality].depth + (size_t)frame * n, n, hipMemcpyDeviceToHost));
  } else {
    set_error("debug_read: unknown item %d", what);
    return LMX_ERR_INVALID_ARG;
  }
  return LMX_OK;
}

lmx_status lmx_debug_orientation_labels(int32_t device, const int16_t* dx, const int16_t* dy, size_t n, uint8_t* out) {
  if (!dx || !dy || !out) { set_error("lmx_debug_orientation_labels: null argument"); return LMX_ERR_INVALID_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available; this library has no CPU path"); return LMX_ERR_NO_DEVICE; }
  LMX_HIP(hipSetDevice(device));
  short *d_dx = nullptr, *d_dy = nullptr;
  uint8_t* d_out = nullptr;
  lmx_status st = LMX_OK;
  auto run = [&]() -> lmx_status {
    LMX_HIP(hipMalloc((void**)&d_dx, n * 2));
    LMX_HIP(hipMalloc((void**)&d_dy, n * 2));
    LMX_HIP(hipMalloc((void**)&d_out, n));
    LMX_HIP(hipMemcpy(d_dx, dx, n * 2, hipMemcpyHostToDevice));
    LMX_HIP(hipMemcpy(d_dy, dy, n * 2, hipMemcpyHostToDevice));
    launch_debug_orientation_label(nullptr, d_dx, d_dy, d_out, n);
    LMX_HIP(hipDeviceSynchronize());
    LMX_HIP(hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    return LMX_OK;
  };
  st = run();
  (void)hipFree(d_dx); (void)hipFree(d_dy); (void)hipFree(d_out);
  return st;
}

lmx_status lmx_ctx_stats(lmx_ctx* c, int64_t* n_candidates, int64_t* n_raw_matches) {
  if (!c) { set_error("lmx_ctx_stats: null context"); return LMX_ERR_INVALID_ARG; }
  if (n_candidates) *n_candidates = c->stat_cands;
  if (n_raw_matches) *n_raw_matches = c->stat_matches;
  return LMX_OK;
}

int32_t lmx_num_kernels(void) { return K_COUNT; }
const char* lmx_kernel_name(int32_t id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : nullptr; }
const char* lmx_ctx_device_kernel_name(lmx_ctx* c, int32_t id) {
  if (!c || id < 0 || id >= K_COUNT) return nullptr;
  if (id == K_SCORE_COARSE) {
    const int v = score_kernel_variant(c->dbank);
    return v == 2 ? "k_score_coarse_sb" : (v == 1 ? "k_score_coarse_u8" : "k_score_coarse");
  }
  if (id == K_SPREAD_LINEARIZE) return "k_spread_linearize_t";
  return kKernelNames[id];
}
lmx_status lmx_ctx_set_profiling(lmx_ctx* c, int32_t enabled) {
  if (!c) { set_error("null context"); return LMX_ERR_INVALID_ARG; }
  c->profiling = (uint32_t)enabled;
  return LMX_OK;
}
lmx_status lmx_ctx_kernel_time(lmx_ctx* c, int32_t id, double* total_ms, int64_t* launches) {
  if (!c || id < 0 || id >= K_COUNT) { set_error("lmx_ctx_kernel_time: bad argument"); return LMX_ERR_INVALID_ARG; }
  if (total_ms) *total_ms = c->k_ms[id];
  if (launches) *launches = c->k_launches[id];
  return LMX_OK;
}
lmx_status lmx_ctx_reset_profiling(lmx_ctx* c) {
  if (!c) { set_error("null context"); return LMX_ERR_INVALID_ARG; }
  for (int i = 0; i < K_COUNT; ++i) { c->k_ms[i] = 0; c->k_launches[i] = 0; }
  return LMX_OK;
}

// Algorithmic bytes per enqueue for one kernel: what the stage must read and write if every
// byte moved exactly once.  For k_score_coarse: sum over templates and modalities of nf * template_positions
// (one linear-memory byte per feature per placement) + the u8 map per modality and the u16 total per placement.
lmx_status lmx_ctx_algorithmic_bytes(lmx_ctx* c, int32_t id, int32_t n_frames, double* bytes) {
  if (!c || !bytes || id < 0 || id >= K_COUNT) { set_error("lmx_ctx_algorithmic_bytes: bad argument"); return LMX_ERR_INVALID_ARG; }
  const lmx_bank* b = c->bank;
  const int L = c->L, M = c->M, per = L * M;
  double v = 0;
  const LevelGeom& g0 = c->kp.geom[0];
  int n_cg = 0, n_dn = 0;
  for (int m = 0; m < M; ++m) (b->mods[m].type == LMX_MOD_COLOR_GRADIENT ? n_cg : n_dn)++;
  switch (id) {
    case K_PRE: v = 0; break;  // depends on the raw frame size passed to upload_raw
    case K_COLOR_QUANTIZE:  // 3 B in + 1 B out per pixel, + 3/4 B for the pyrDown output of the next level
      for (int l = 0; l < L; ++l) v += n_cg * (4.0 + (l + 1 < L ? 0.75 : 0.0)) * c->kp.geom[l].W * c->kp.geom[l].H;
      break;
    case K_DEPTH_QUANTIZE: v = n_dn * 3.0 * g0.W * g0.H; break;
    case K_NN_DOWN:
      for (int l = 1; l < L; ++l) v += n_dn * 2.0 * c->kp.geom[l].W * c->kp.geom[l].H;
      break;
    case K_SPREAD_LINEARIZE:  // 1 B in; at the coarsest level eight response maps out (4 B/px nibble-packed, 8 B/px in the generic
                              // byte path), 1 B out (spread byte) at finer levels
      for (int l = 0; l < L; ++l)
        v += M * (l == L - 1 ? (spread_writes_nibbles(c->kp.geom[l]) ? 5.0 : 9.0) : 2.0) * c->kp.geom[l].W * c->kp.geom[l].H;
      break;
    case K_PACK_NIBBLES: v = spread_writes_nibbles(c->kp.geom[L - 1]) ? 0.0 : M * 12.0 * c->kp.geom[L - 1].W * c->kp.geom[L - 1].H; break;  // generic path only
    case K_SCORE_COARSE: {
      const LevelGeom& g = c->kp.geom[L - 1];
      const int world = c->desc.shard_world, rank = c->desc.shard_rank;
      for (const auto& kv : b->classes) {
        const ClassData& cd = kv.second;
        const long n = cd.n_pyramids;
        for (long t = (rank * n) / world; t < ((rank + 1) * n) / world; ++t)
          for (int m = 0; m < M; ++m) {
            const int32_t* tm = &cd.templates[((size_t)t * per + (size_t)(L - 1) * M + m) * 5];
            const int wf = (tm[0] - 1) / g.T + 1, hf = (tm[1] - 1) / g.T + 1;
            const double pos = std::max<long>(0, (long)(g.Hc - hf) * g.Wc + (g.Wc - wf) + 1);
            v += tm[4] * pos + 3.0 * g.cells;
          }
      }
      break;
    }
    case K_REFINE: v = 0; break;  // depends on the candidate count of the frame; reported from stats by the caller
  }
  *bytes = v * n_frames;
  return LMX_OK;
}

}  // extern "C"

// ---- rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU (host) ----------------
namespace {
struct HostCluster {
  std::vector<int> index;
  double score = 0;
  bool suppressed = false;
  int rect[4] = {0, 0, 0, 0};
  std::vector<int32_t> members;  // indices into the caller's match array, in the order they were voted in
};
bool by_score_desc(const HostCluster& a, const HostCluster& b) { return a.score > b.score; }  // the comparator of rgbdDetector.h:127-130

// Overlap of two boxes {x, y, w, h} the way the reference's NMS measures it (rgbdDetector.cpp:532-574): inclusive pixel extents,
// the intersection area as an int product converted to float, the union in float, float division.  Same arithmetic as the
// device version in lmx_f2.hip (box_iou): the int/float mix is part of the observable result.
// The reference does this in plain `int`; with the rects its size_t division produces for clusters left of / above the origin
// (coordinates near 2^32 / n) those sums and products overflow, which on its platform wraps.  Here the wrap is spelled out
// (unsigned arithmetic, then back to int): the same values without undefined behaviour (found by UBSan on the host build).
inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
float box_overlap_ratio(const int* p, const int* q) {
  const int p_x0 = p[0], p_x1 = wsub(wadd(p[0], p[2]), 1), p_y0 = p[1], p_y1 = wsub(wadd(p[1], p[3]), 1);
  const int q_x0 = q[0], q_x1 = wsub(wadd(q[0], q[2]), 1), q_y0 = q[1], q_y1 = wsub(wadd(q[1], q[3]), 1);
  const int lo_x = std::max(p_x0, q_x0), hi_x = std::min(p_x1, q_x1), lo_y = std::max(p_y0, q_y0), hi_y = std::min(p_y1, q_y1);
  const bool overlap_x = (lo_x >= p_x0 && lo_x <= p_x1) || (lo_x >= q_x0 && lo_x <= q_x1);
  const bool overlap_y = (lo_y >= p_y0 && lo_y <= p_y1) || (lo_y >= q_y0 && lo_y <= q_y1);
  const float shared = (overlap_x && overlap_y) ? (float)wmul(wadd(wsub(hi_x, lo_x), 1), wadd(wsub(hi_y, lo_y), 1)) : 0.0f;
  const float total = (float)wadd(wmul(p[2], p[3]), wmul(q[2], q[3])) - shared;
  return shared / total;
}
}  // namespace

extern "C" lmx_status lmx_cluster_matches(const lmx_match_t* matches, size_t n_matches, const double* obj_origin_dists, const int32_t* rects,
                                          size_t n_templates, const lmx_cluster_params* pp, lmx_cluster_t* clusters, size_t cap_clusters,
                                          size_t* n_clusters, int32_t* members, size_t cap_members) {
  return lmx::guarded("lmx_cluster_matches", [&]() -> lmx_status {
  if ((n_matches && !matches) || !obj_origin_dists || !rects || !pp || !n_clusters || (cap_clusters && !clusters) || (cap_members && !members)) {
    lmx::set_error("lmx_cluster_matches: null argument");
    return LMX_ERR_INVALID_ARG;
  }
  if (pp->vote_row_col_step <= 0) { lmx::set_error("vote_row_col_step must be positive"); return LMX_ERR_INVALID_ARG; }
  if (pp->cluster_size_thresh < 0) { lmx::set_error("cluster_size_thresh must not be negative"); return LMX_ERR_INVALID_ARG; }   // see lmx_ctx_set_cluster_sidecar
  if (lmx_status vs = check_vote_rings(obj_origin_dists, n_templates, pp)) return vs;
  // rcd_voting: bins keyed by {y/step, x/step, depth ring}; std::map keeps them in lexicographic order like upstream
  std::map<std::vector<int>, std::vector<int32_t>> map_match;
  const float voting_depth_step = (float)pp->renderer_radius_step;
  for (size_t i = 0; i < n_matches; ++i) {
    const lmx_match_t& m = matches[i];
    if (m.template_id < 0 || (size_t)m.template_id >= n_templates) { lmx::set_error("match %zu: template_id %d outside the side-car arrays", i, m.template_id); return LMX_ERR_INVALID_ARG; }
    const float depth = (float)obj_origin_dists[m.template_id];
    std::vector<int> index(3);
    index[0] = m.y / pp->vote_row_col_step;
    index[1] = m.x / pp->vote_row_col_step;
    index[2] = (int)((depth - pp->renderer_radius_min) / voting_depth_step);
    map_match[index].push_back((int32_t)i);
  }
  // cluster_filter(map, thresh) -- intended semantics (see header) -- and cluster_scoring (similarity_score_calc)
  std::vector<HostCluster> cd;
  for (auto it = map_match.begin(); it != map_match.end(); ++it) {
    if ((long)it->second.size() <= (long)pp->cluster_size_thresh) continue;
    HostCluster c;
    c.index = it->first;
    double sum_score = 0.0;
    int num = 0;
    for (int32_t mi : it->second) { sum_score += matches[mi].similarity; num++; }
    c.score = sum_score / num;
    c.members = it->second;
    cd.push_back(c);
  }
  if (!cd.empty()) {
    // nonMaximaSuppressionUsingIOU: mean rect, sort by score (std::sort, like upstream), greedy suppression at IoU > 0.4
    for (HostCluster& c : cd) {
      int sum_x = 0, sum_y = 0, sum_w = 0, sum_h = 0;   // integer sums, like the reference
      for (int32_t mi : c.members) {
        const int32_t* r = rects + (size_t)matches[mi].template_id * 4;
        sum_x = wadd(sum_x, matches[mi].x); sum_y = wadd(sum_y, matches[mi].y); sum_w = wadd(sum_w, r[2]); sum_h = wadd(sum_h, r[3]);   // int, wrapping
      }
      // `X /= it1->matches.size();` in the reference divides by a size_t: the int sum is converted to size_t first, so a negative
      // sum (matches left of / above the origin) divides as 2^64 + X; the quotient goes back to int
      const size_t n = c.members.size();
      auto div_by_size = [n](int v) { return (int)(unsigned)((unsigned long long)(long long)v / (unsigned long long)n); };
      c.rect[0] = div_by_size(sum_x); c.rect[1] = div_by_size(sum_y); c.rect[2] = div_by_size(sum_w); c.rect[3] = div_by_size(sum_h);
    }
    std::sort(cd.begin(), cd.end(), by_score_desc);
    for (size_t a = 0; a < cd.size(); ++a) {
      if (cd[a].suppressed) continue;
      for (size_t b = a + 1; b < cd.size(); ++b)
        if (!cd[b].suppressed) {
          const double ratio = box_overlap_ratio(cd[a].rect, cd[b].rect);
          if (ratio > 0.4) cd[b].suppressed = true;
        }
    }
  }
  size_t nc = 0, nm = 0;
  lmx_status st = LMX_OK;
  for (const HostCluster& c : cd) {
    if (c.suppressed) continue;
    if (nc < cap_clusters && nm + c.members.size() <= cap_members) {
      lmx_cluster_t& o = clusters[nc];
      o.index[0] = c.index[0]; o.index[1] = c.index[1]; o.index[2] = c.index[2];
      for (int k = 0; k < 4; ++k) o.rect[k] = c.rect[k];
      o.score = c.score;
      o.member_begin = (int32_t)nm; o.member_count = (int32_t)c.members.size();
      std::memcpy(members + nm, c.members.data(), c.members.size() * sizeof(int32_t));
    } else {
      st = LMX_ERR_OVERFLOW;
    }
    nc += 1; nm += c.members.size();
  }
  *n_clusters = nc;
  if (st != LMX_OK) lmx::set_error("%zu clusters / %zu members exceed the output capacity", nc, nm);
  return st;
  });
}

// ---- caches for the per-request detector rebuild of the reference's service node (include/lmx.h) -----------------------------
namespace {

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

// Eight bytes per step for the bank fingerprint (an in-process cache key, not a file format: the binary bank's checksum stays the
// byte-wise FNV-1a above).  A freshly built bank of 3000 RGB-D templates (7 MB of templates and features) hashes in ~0.7 instead of 5.5 ms,
// which is what a detector rebuilt from FileNodes on every request pays before lmx_ctx_acquire can find its context.
uint64_t hash_words(uint64_t h, const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    std::memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x9e3779b97f4a7c15ull;
    h ^= h >> 29;
  }
  return fnv1a(h, p + i, n - i);
}

struct BankCacheEntry { std::string path; long long mtime_ns; long long size; uint64_t lut_key; lmx_bank* bank; int refs; };
struct CtxCacheEntry { uint64_t fingerprint; lmx_ctx_desc desc; lmx_bank* bank; lmx_ctx* ctx; int refs; uint64_t last_use; };

std::mutex g_cache_mutex;
std::vector<BankCacheEntry> g_bank_cache;
std::vector<CtxCacheEntry> g_ctx_cache;
uint64_t g_cache_clock = 0;
constexpr size_t kMaxIdleContexts = 8;

bool same_desc(const lmx_ctx_desc& a, const lmx_ctx_desc& b) {
  return a.device == b.device && a.width == b.width && a.height == b.height && a.max_batch == b.max_batch && a.max_candidates == b.max_candidates &&
         a.shard_rank == b.shard_rank && a.shard_world == b.shard_world && a.stream == b.stream && a.flags == b.flags;
}

}  // namespace

extern "C" {

lmx_status lmx_bank_clone(const lmx_bank* bank, lmx_bank** out) {
  return lmx::guarded("lmx_bank_clone", [&]() -> lmx_status {
  if (!bank || !out) { set_error("lmx_bank_clone: null argument"); return LMX_ERR_INVALID_ARG; }
  *out = new lmx_bank(*bank);
  return LMX_OK;
  });
}

uint64_t lmx_bank_fingerprint(const lmx_bank* bank) {
  if (!bank) return 0;
  // what the remembered value covered: counts of everything hashed below + the table's epoch (never 0, so that an empty cache misses)
  uint64_t sig = 0x9e3779b97f4a7c15ull ^ bank->T.size() ^ (bank->mods.size() << 8) ^ ((uint64_t)bank->lut_epoch << 16) ^ ((uint64_t)bank->normal_lut.size() << 40) ^
                 ((uint64_t)(uint32_t)bank->normal_lut_origin << 56);
  for (const auto& kv : bank->classes)
    sig = (sig * 0x100000001b3ull) ^ (kv.second.templates.size() * 0x9e3779b1ull) ^ (kv.second.features.size() << 20) ^ kv.first.size();
  sig |= 1ull;
  if (bank->fp_cache.signature.load(std::memory_order_acquire) == sig) return bank->fp_cache.value.load(std::memory_order_relaxed);
  uint64_t h = 0xcbf29ce484222325ull;
  h = fnv1a(h, bank->T.data(), bank->T.size() * sizeof(int32_t));
  for (const lmx_modality_desc& m : bank->mods) {
    // field by field: the struct has no padding today, but the hash must not depend on that
    h = fnv1a(h, &m.type, sizeof(m.type)); h = fnv1a(h, &m.weak_threshold, sizeof(float)); h = fnv1a(h, &m.strong_threshold, sizeof(float));
    h = fnv1a(h, &m.num_features, 4); h = fnv1a(h, &m.distance_threshold, 4); h = fnv1a(h, &m.difference_threshold, 4); h = fnv1a(h, &m.extract_threshold, 4);
  }
  h = hash_words(h, bank->normal_lut.data(), bank->normal_lut.size());
  for (const auto& kv : bank->classes) {
    h = fnv1a(h, kv.first.data(), kv.first.size() + 1);
    h = fnv1a(h, &kv.second.n_pyramids, 4);
    h = hash_words(h, kv.second.templates.data(), kv.second.templates.size() * sizeof(int32_t));
    h = hash_words(h, kv.second.features.data(), kv.second.features.size() * sizeof(int32_t));
  }
  bank->fp_cache.value.store(h, std::memory_order_relaxed);
  bank->fp_cache.signature.store(sig, std::memory_order_release);
  return h;
}

// ---- compact binary form of a bank ("+ a compact binary cache") ---------------------------------------------
// Layout (little endian): "LMXBANK1", then u64 fields and raw int32 arrays as written below, then the FNV-1a hash of everything
// before it.  A 3000-template RGB-D bank is 22.7 MB of FileStorage YAML (0.5 s to parse here, seconds in OpenCV) and 7 MB /
// 14 ms in this form.
namespace {
struct Writer {
  std::vector<uint8_t> buf;
  void raw(const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); buf.insert(buf.end(), b, b + n); }
  void u64(uint64_t v) { raw(&v, 8); }
};
struct Reader {
  const uint8_t* p; size_t n, pos = 0; bool ok = true;
  bool raw(void* dst, size_t k) { if (!ok || pos + k > n) { ok = false; return false; } std::memcpy(dst, p + pos, k); pos += k; return true; }
  uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
};

void serialize_bank(const lmx_bank* b, Writer& w) {
  w.raw("LMXBANK1", 8);
  w.u64(b->T.size()); w.raw(b->T.data(), b->T.size() * 4);
  w.u64(b->mods.size());
  for (const lmx_modality_desc& m : b->mods) {
    const int32_t ints[5] = {m.type, m.num_features, m.distance_threshold, m.difference_threshold, m.extract_threshold};
    w.raw(ints, sizeof(ints)); w.raw(&m.weak_threshold, 4); w.raw(&m.strong_threshold, 4);
  }
  w.u64((uint64_t)b->normal_lut_origin); w.raw(b->normal_lut.data(), LMX_NORMAL_LUT_SIZE);
  w.u64(b->classes.size());
  for (const auto& kv : b->classes) {
    w.u64(kv.first.size()); w.raw(kv.first.data(), kv.first.size());
    w.u64((uint64_t)kv.second.n_pyramids);
    w.u64(kv.second.templates.size()); w.raw(kv.second.templates.data(), kv.second.templates.size() * 4);
    w.u64(kv.second.features.size()); w.raw(kv.second.features.data(), kv.second.features.size() * 4);
  }
  w.u64(fnv1a(0xcbf29ce484222325ull, w.buf.data(), w.buf.size()));
}

lmx_status deserialize_bank(const uint8_t* data, size_t n, lmx_bank** out, const char* what) {
  if (n < 16 || std::memcmp(data, "LMXBANK1", 8) != 0) { set_error("'%s' is not a liblmx binary bank", what); return LMX_ERR_PARSE; }
  uint64_t stored = 0;
  std::memcpy(&stored, data + n - 8, 8);
  if (stored != fnv1a(0xcbf29ce484222325ull, data, n - 8)) { set_error("'%s': checksum mismatch (truncated or corrupted)", what); return LMX_ERR_PARSE; }
  Reader r{data, n - 8};
  r.pos = 8;
  std::unique_ptr<lmx_bank> b(new lmx_bank());
  const uint64_t L = r.u64();
  if (!r.ok || L < 1 || L > (uint64_t)kMaxLevels) { set_error("'%s': bad header", what); return LMX_ERR_PARSE; }
  b->T.resize(L); r.raw(b->T.data(), L * 4);
  for (uint64_t l = 0; r.ok && l < L; ++l)
    if (b->T[l] < 1 || b->T[l] > 16) { set_error("'%s': T=%d at level %d outside 1..16", what, b->T[l], (int)l); return LMX_ERR_PARSE; }
  const uint64_t M = r.u64();
  if (!r.ok || M < 1 || M > (uint64_t)kMaxModalities) { set_error("'%s': bad header", what); return LMX_ERR_PARSE; }
  for (uint64_t m = 0; m < M; ++m) {
    int32_t ints[5];
    lmx_modality_desc d{};
    r.raw(ints, sizeof(ints)); r.raw(&d.weak_threshold, 4); r.raw(&d.strong_threshold, 4);
    d.type = ints[0]; d.num_features = ints[1]; d.distance_threshold = ints[2]; d.difference_threshold = ints[3]; d.extract_threshold = ints[4];
    if (r.ok && d.type != LMX_MOD_COLOR_GRADIENT && d.type != LMX_MOD_DEPTH_NORMAL) { set_error("'%s': unknown modality type %d", what, d.type); return LMX_ERR_PARSE; }
    b->mods.push_back(d);
  }
  b->normal_lut_origin = (int32_t)r.u64();
  b->normal_lut.resize(LMX_NORMAL_LUT_SIZE); r.raw(b->normal_lut.data(), LMX_NORMAL_LUT_SIZE);
  if (r.ok) {
    std::vector<uint8_t> bins(LMX_NORMAL_LUT_SIZE);
    if (b->normal_lut_origin < LMX_LUT_DEFAULT || b->normal_lut_origin > LMX_LUT_UNKNOWN) { set_error("'%s': bad normal-LUT origin %d", what, b->normal_lut_origin); return LMX_ERR_PARSE; }
    if (!normal_lut_to_bins(b->normal_lut.data(), bins.data())) { set_error("'%s': normal LUT entries must be 0 or a single bit", what); return LMX_ERR_PARSE; }
  }
  const uint64_t nc = r.u64();
  for (uint64_t c = 0; r.ok && c < nc; ++c) {
    const uint64_t len = r.u64();
    if (!r.ok || len > 4096) { r.ok = false; break; }
    std::string name(len, '\0');
    r.raw(&name[0], len);
    const int64_t n_pyr = (int64_t)r.u64();
    const uint64_t nt = r.u64();
    if (!r.ok || nt > (n / 4) || n_pyr < 0 || n_pyr > (int64_t)(n / 20)) { r.ok = false; break; }
    std::vector<int32_t> templates(nt);
    r.raw(templates.data(), nt * 4);
    const uint64_t nf = r.u64();
    if (!r.ok || nf > (n / 4) || nf % 3 != 0) { r.ok = false; break; }
    std::vector<int32_t> features(nf);
    r.raw(features.data(), nf * 4);
    if (!r.ok || templates.size() != (size_t)n_pyr * L * M * 5 || b->classes.count(name)) { r.ok = false; break; }
    // the same validation every other way into a bank goes through (feature counts <= 63, ranges inside `features`, pyramid
    // levels, coordinates, labels 0..7): a stale-format, damaged-but-rehashed or crafted file must not reach build_device_bank
    static const int32_t none[5] = {0, 0, 0, 0, 0};
    const lmx_status vs = lmx_bank_add_class(b.get(), name.c_str(), (int32_t)n_pyr, templates.empty() ? none : templates.data(), features.empty() ? none : features.data(),
                                             (int64_t)(nf / 3));
    if (vs != LMX_OK) { const std::string why = g_error; set_error("'%s': class '%s' is invalid: %s", what, name.c_str(), why.c_str()); return LMX_ERR_PARSE; }
  }
  if (!r.ok || r.pos != n - 8) { set_error("'%s': malformed binary bank", what); return LMX_ERR_PARSE; }
  *out = b.release();
  return LMX_OK;
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  uint8_t tmp[1 << 16];
  size_t k;
  while ((k = std::fread(tmp, 1, sizeof(tmp), f)) > 0) out.insert(out.end(), tmp, tmp + k);
  std::fclose(f);
  return true;
}
}  // namespace

lmx_status lmx_bank_save_binary(const lmx_bank* bank, const char* path) {
  return lmx::guarded("lmx_bank_save_binary", [&]() -> lmx_status {
  if (!bank || !path) { set_error("lmx_bank_save_binary: null argument"); return LMX_ERR_INVALID_ARG; }
  Writer w;
  serialize_bank(bank, w);
  FILE* f = std::fopen(path, "wb");
  if (!f) { set_error("cannot open '%s' for writing", path); return LMX_ERR_IO; }
  const bool ok = std::fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
  if (std::fclose(f) != 0 || !ok) { set_error("write error on '%s'", path); return LMX_ERR_IO; }
  return LMX_OK;
  });
}

lmx_status lmx_bank_load_binary(const char* path, lmx_bank** out) {
  return lmx::guarded("lmx_bank_load_binary", [&]() -> lmx_status {
  if (!path || !out) { set_error("lmx_bank_load_binary: null argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<uint8_t> data;
  if (!read_file(path, data)) { set_error("cannot open '%s'", path); return LMX_ERR_IO; }
  return deserialize_bank(data.data(), data.size(), out, path);
  });
}

// Everything outside the yml that yaml_load folds into the bank: the side-car table `<yml>.normal_lut` and the file the environment
// variable LMX_NORMAL_LUT names (existence, mtime, size, and the variable's value).  Part of both cache keys: a table that appears
// or changes later must not be masked by a bank cached without it (advisor finding, round 2).
static uint64_t lut_inputs_key(const char* yml_path) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix_file = [&](const char* p) {
    struct stat sb;
    long long v[3] = {0, 0, 0};
    if (stat(p, &sb) == 0) { v[0] = 1; v[1] = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec; v[2] = (long long)sb.st_size; }
    h = fnv1a(h, v, sizeof(v));
  };
  mix_file((std::string(yml_path) + ".normal_lut").c_str());
  const char* env = std::getenv("LMX_NORMAL_LUT");
  if (env && *env) { h = fnv1a(h, env, std::strlen(env) + 1); mix_file(env); }
  return h;
}

lmx_status lmx_bank_load_yaml_cached(const char* path, const lmx_bank** out) {
  return lmx::guarded("lmx_bank_load_yaml_cached", [&]() -> lmx_status {
  if (!path || !out) { set_error("lmx_bank_load_yaml_cached: null argument"); return LMX_ERR_INVALID_ARG; }
  struct stat sb;
  if (stat(path, &sb) != 0) { set_error("cannot open '%s'", path); return LMX_ERR_IO; }
  const long long mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec, size = (long long)sb.st_size;
  const uint64_t lut_key = lut_inputs_key(path);
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (size_t i = 0; i < g_bank_cache.size(); ++i) {
    BankCacheEntry& e = g_bank_cache[i];
    if (e.path != path) continue;
    if (e.mtime_ns == mtime_ns && e.size == size && e.lut_key == lut_key) { e.refs += 1; *out = e.bank; return LMX_OK; }
    if (e.refs == 0) { delete e.bank; g_bank_cache.erase(g_bank_cache.begin() + (long)i); --i; }  // stale and unused
    // a stale entry that is still referenced stays until released; the new version gets its own entry
  }
  // second level: "<path>.lmxcache" next to the yml = {"LMXCACH2", mtime and size of the yml it was made from, key of the table inputs,
  // binary bank}; written on a miss when the directory allows it, ignored when stale, of another format or unreadable -- the yml is then
  // parsed again (LMX_NO_DISK_CACHE=1 turns it off)
  lmx_bank* b = nullptr;
  static const bool disk = std::getenv("LMX_NO_DISK_CACHE") == nullptr;
  const std::string cache_path = std::string(path) + ".lmxcache";
  if (disk) {
    std::vector<uint8_t> data;
    if (read_file(cache_path.c_str(), data) && data.size() > 32 && std::memcmp(data.data(), "LMXCACH2", 8) == 0) {
      long long c_mtime = 0, c_size = 0;
      uint64_t c_lut = 0;
      std::memcpy(&c_mtime, data.data() + 8, 8); std::memcpy(&c_size, data.data() + 16, 8); std::memcpy(&c_lut, data.data() + 24, 8);
      if (c_mtime == mtime_ns && c_size == size && c_lut == lut_key && deserialize_bank(data.data() + 32, data.size() - 32, &b, cache_path.c_str()) != LMX_OK) b = nullptr;
    }
  }
  if (!b) {
    lmx_status st = yaml_load(path, &b);
    if (st != LMX_OK) return st;
    if (disk) {
      Writer w;
      w.raw("LMXCACH2", 8); w.raw(&mtime_ns, 8); w.raw(&size, 8); w.raw(&lut_key, 8);
      Writer body;
      serialize_bank(b, body);
      w.raw(body.buf.data(), body.buf.size());
      const std::string tmp = cache_path + ".tmp";
      FILE* f = std::fopen(tmp.c_str(), "wb");
      if (f) {
        const bool ok = std::fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
        if (std::fclose(f) == 0 && ok) (void)std::rename(tmp.c_str(), cache_path.c_str());
        else (void)std::remove(tmp.c_str());
      }
    }
  }
  g_bank_cache.push_back(BankCacheEntry{path, mtime_ns, size, lut_key, b, 1});
  *out = b;
  return LMX_OK;
  });
}

void lmx_bank_release(const lmx_bank* bank) {
  if (!bank) return;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (BankCacheEntry& e : g_bank_cache)
    if (e.bank == bank && e.refs > 0) { e.refs -= 1; return; }   // stays cached for the next request
}

lmx_status lmx_ctx_acquire(const lmx_bank* bank, const lmx_ctx_desc* desc, lmx_ctx** out, int32_t* cache_hit) {
  return lmx::guarded("lmx_ctx_acquire", [&]() -> lmx_status {
  if (!bank || !desc || !out) { set_error("lmx_ctx_acquire: null argument"); return LMX_ERR_INVALID_ARG; }
  const uint64_t fp = lmx_bank_fingerprint(bank);
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (CtxCacheEntry& e : g_ctx_cache)
    if (e.fingerprint == fp && same_desc(e.desc, *desc) && e.bank->normal_lut_origin == bank->normal_lut_origin) {
      e.refs += 1; e.last_use = ++g_cache_clock;
      *out = e.ctx;
      if (cache_hit) *cache_hit = 1;
      return LMX_OK;
    }
  lmx_bank* own = new lmx_bank(*bank);
  lmx_ctx* ctx = nullptr;
  lmx_status st = lmx_ctx_create(own, desc, &ctx);
  if (st != LMX_OK) { delete own; return st; }
  // evict idle contexts beyond the limit, least recently used first
  for (;;) {
    size_t idle = 0, victim = g_ctx_cache.size();
    for (size_t i = 0; i < g_ctx_cache.size(); ++i)
      if (g_ctx_cache[i].refs == 0) { ++idle; if (victim == g_ctx_cache.size() || g_ctx_cache[i].last_use < g_ctx_cache[victim].last_use) victim = i; }
    if (idle < kMaxIdleContexts) break;
    lmx_ctx_destroy(g_ctx_cache[victim].ctx);
    delete g_ctx_cache[victim].bank;
    g_ctx_cache.erase(g_ctx_cache.begin() + (long)victim);
  }
  g_ctx_cache.push_back(CtxCacheEntry{fp, *desc, own, ctx, 1, ++g_cache_clock});
  *out = ctx;
  if (cache_hit) *cache_hit = 0;
  return LMX_OK;
  });
}

void lmx_ctx_unref(lmx_ctx* ctx) {
  if (!ctx) return;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (CtxCacheEntry& e : g_ctx_cache)
    if (e.ctx == ctx && e.refs > 0) { e.refs -= 1; e.last_use = ++g_cache_clock; return; }
}

void lmx_cache_trim(void) {
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (size_t i = 0; i < g_ctx_cache.size();) {
    if (g_ctx_cache[i].refs == 0) {
      lmx_ctx_destroy(g_ctx_cache[i].ctx);
      delete g_ctx_cache[i].bank;
      g_ctx_cache.erase(g_ctx_cache.begin() + (long)i);
    } else {
      ++i;
    }
  }
  for (size_t i = 0; i < g_bank_cache.size();) {
    if (g_bank_cache[i].refs == 0) { delete g_bank_cache[i].bank; g_bank_cache.erase(g_bank_cache.begin() + (long)i); }
    else ++i;
  }
}

}  // extern "C"

// Test hooks for csrc/lmx_sort_emul.hpp (host build of the code the device runs): the permutation the restated introsort
// produces for Match::operator< on (similarity, template_id) and for the cluster comparator score-descending.
extern "C" {
lmx_status lmx_debug_introsort_perm(const float* similarity, const int32_t* template_id, int32_t n, int32_t* perm) {
  if (n < 0 || (n > 0 && (!similarity || !template_id || !perm))) { set_error("lmx_debug_introsort_perm: invalid argument"); return LMX_ERR_INVALID_ARG; }
  for (int32_t i = 0; i < n; ++i) perm[i] = i;
  lmx::sortemu::sort(perm, n, [&](int32_t a, int32_t b) { return similarity[a] != similarity[b] ? similarity[a] > similarity[b] : template_id[a] < template_id[b]; });
  return LMX_OK;
}
lmx_status lmx_debug_device_sort_perm(int32_t device, const float* similarity, const int32_t* template_id, int32_t n, int32_t* perm) {
  if (n < 0 || n > F2_MAX || (n > 0 && (!similarity || !template_id || !perm))) { set_error("lmx_debug_device_sort_perm: invalid argument (n <= %d)", F2_MAX); return LMX_ERR_INVALID_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available; this library has no CPU path"); return LMX_ERR_NO_DEVICE; }
  if (n == 0) return LMX_OK;
  LMX_HIP(hipSetDevice(device));
  float* d_sim = nullptr; int* d_tid = nullptr; int* d_perm = nullptr; unsigned long long* d_spill = nullptr;
  auto run = [&]() -> lmx_status {
    LMX_HIP(hipMalloc((void**)&d_sim, (size_t)n * 4)); LMX_HIP(hipMalloc((void**)&d_tid, (size_t)n * 4)); LMX_HIP(hipMalloc((void**)&d_perm, (size_t)n * 4));
    LMX_HIP(hipMalloc((void**)&d_spill, (size_t)F2_MAX * 8));
    LMX_HIP(hipMemcpy(d_sim, similarity, (size_t)n * 4, hipMemcpyHostToDevice));
    LMX_HIP(hipMemcpy(d_tid, template_id, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_debug_block_sort(nullptr, d_sim, d_tid, n, d_perm, d_spill);
    LMX_HIP(hipGetLastError());
    LMX_HIP(hipDeviceSynchronize());
    LMX_HIP(hipMemcpy(perm, d_perm, (size_t)n * 4, hipMemcpyDeviceToHost));
    return LMX_OK;
  };
  const lmx_status st = run();
  (void)hipFree(d_sim); (void)hipFree(d_tid); (void)hipFree(d_perm); (void)hipFree(d_spill);
  return st;
}
lmx_status lmx_debug_introsort_perm_score(const double* score, int32_t n, int32_t* perm) {
  if (n < 0 || (n > 0 && (!score || !perm))) { set_error("lmx_debug_introsort_perm_score: invalid argument"); return LMX_ERR_INVALID_ARG; }
  for (int32_t i = 0; i < n; ++i) perm[i] = i;
  lmx::sortemu::sort(perm, n, [&](int32_t a, int32_t b) { return score[a] > score[b]; });
  return LMX_OK;
}
}  // extern "C"
