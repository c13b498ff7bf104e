// Internal declarations shared by the host side (lmx_ctx.hpp lists its translation units) and the HIP kernels
// (lmx_kernels.hip) of liblmx.so.  Nothing here crosses the C ABI (include/lmx.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <new>
#include <exception>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lmx.h"

namespace lmx {

// ---- host bank (mirrors cv::linemod::Detector's template state) ---------------------------
struct ClassData {
  std::string id;
  int32_t n_pyramids = 0;
  std::vector<int32_t> templates;  // [n_pyramids * L*M][5] {width, height, pyramid_level, feat_begin, feat_count}
  std::vector<int32_t> features;   // [n][3] {x, y, label}
};

}  // namespace lmx

struct lmx_bank {
  std::vector<int32_t> T;
  std::vector<lmx_modality_desc> mods;
  std::map<std::string, lmx::ClassData> classes;  // std::map: upstream iterates classes in key order (A.10)
  std::vector<uint8_t> normal_lut;                // DepthNormal NORMAL_LUT[20][20][20] (one-hot labels), see include/lmx.h
  int32_t normal_lut_origin = 0;                  // LMX_LUT_*
  uint32_t lut_epoch = 0;                         // bumped by everything that replaces normal_lut (same size, new content)
  // lmx_bank_fingerprint hashes every template and feature (3.5 MB for 3000 templates: 2.8 ms), and lmx_ctx_acquire asks for it on every
  // request: remembered together with a signature of what it covered (element counts and lut_epoch: the API only appends templates or
  // replaces the table).  Copy-constructible on purpose (lmx_ctx_acquire keeps a private copy of the caller's bank).
  struct FingerprintCache {
    std::atomic<uint64_t> value{0}, signature{0};
    FingerprintCache() = default;
    FingerprintCache(const FingerprintCache& o) : value(o.value.load(std::memory_order_relaxed)), signature(o.signature.load(std::memory_order_relaxed)) {}
    FingerprintCache& operator=(const FingerprintCache& o) {
      value.store(o.value.load(std::memory_order_relaxed), std::memory_order_relaxed);
      signature.store(o.signature.load(std::memory_order_relaxed), std::memory_order_relaxed);
      return *this;
    }
  };
  mutable FingerprintCache fp_cache;
};

namespace lmx {

void set_error(const char* fmt, ...);
void stream_copy(void* dst, const void* src, size_t n);   // lmx_hostcopy.cpp: copy into pinned staging with non-temporal stores
void stream_store_flag(uint32_t* flag, uint32_t value);    // lmx_hostcopy.cpp: the progress word of a streamed frame store (see StreamWait)
lmx_status yaml_load(const char* path, lmx_bank** out);
lmx_status yaml_save(const lmx_bank* bank, const char* path);
void default_normal_lut(uint8_t* out /* [LMX_NORMAL_LUT_SIZE] */);
// labels -> median bins (0 for "no label", k + 1 for 1 << k): the form the depth kernel reads; false if an entry is not one-hot/0
bool normal_lut_to_bins(const uint8_t* lut, uint8_t* bins);
lmx_status normal_lut_from_file(const char* path, std::vector<uint8_t>& out);

// ---- host thread pool ------------------------------------------------------------------------------------
// Host threads for the staging copies of lmx_ctx_upload (pageable caller memory -> pinned staging): one batch of 64 RGB-D
// frames is 98 MB, which a single thread copies slower than PCIe moves it.  parallel_for hands out task indices through an
// atomic counter; the calling thread works too.  Created on first use (LMX_UPLOAD_THREADS overrides the thread count).
class CopyPool {
 public:
  explicit CopyPool(int n_threads) {
    for (int i = 0; i < n_threads; ++i) workers_.emplace_back([this]() { run(); });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  void parallel_for(int n_tasks, const std::function<void(int)>& fn) {
    if (n_tasks <= 0) return;
    if (workers_.empty() || n_tasks == 1) { for (int i = 0; i < n_tasks; ++i) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; n_tasks_ = n_tasks; next_.store(0); busy_ = (int)workers_.size(); ++generation_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [this]() { return busy_ == 0; });
    fn_ = nullptr;
  }
  int threads() const { return (int)workers_.size() + 1; }

 private:
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_tasks_) break;
      (*fn_)(i);
    }
  }
  void run() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--busy_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_tasks_ = 0, busy_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};


// One persistent helper thread for the one-frame call (lmx_enqueue.cpp issue_small): the caller hands it a job (queue the rest of the kernel
// chain, then store the depth frame) and stores the colour frame itself.  A job arrives every ~100 us when frames are matched back to back and
// a condition-variable wake-up costs 10-50 us, so the helper SPINS for a bounded time after a job (300 us) before it goes to sleep: a caller in
// a tight loop finds it awake, a 1 Hz caller (the reference's demo node) pays a wake-up it will not notice and no core is kept busy.
class LaunchHelper {
 public:
  LaunchHelper() : th_([this]() { run(); }) {}
  ~LaunchHelper() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    th_.join();
  }
  void submit(const std::function<void()>* job) {
    job_ = job;
    done_.store(false, std::memory_order_relaxed);
    gen_.fetch_add(1, std::memory_order_release);
    if (sleeping_.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> lk(m_); cv_.notify_one(); }
  }
  void wait() const {
    while (!done_.load(std::memory_order_acquire)) __builtin_ia32_pause();
  }

 private:
  void run() {
    uint64_t seen = 0;
    for (;;) {
      const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      int spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        __builtin_ia32_pause();
        if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
          std::unique_lock<std::mutex> lk(m_);
          sleeping_.store(true, std::memory_order_release);
          cv_.wait(lk, [&]() { return gen_.load(std::memory_order_acquire) != seen; });
          sleeping_.store(false, std::memory_order_release);
        }
      }
      seen = gen_.load(std::memory_order_acquire);
      { std::lock_guard<std::mutex> lk(m_); if (stop_) return; }
      (*job_)();
      done_.store(true, std::memory_order_release);
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<bool> done_{true}, sleeping_{false};
  const std::function<void()>* job_ = nullptr;
  bool stop_ = false;
  std::thread th_;   // last: the thread starts with every other member constructed
};

// ---- hooks for device groups (lmx_group.cpp): several member contexts fed from ONE pinned staging area ------------------
// A group stages a batch of host frames once (layout: modality m at offset sum_{m' < m} frame_bytes[m'] * max_batch, frames
// back to back) and every member DMAs it into its own next frame set.  The members' frame sets advance in lock step, so staging
// area j is only ever read by transfers into set j of each member.
constexpr int32_t LMX_CTX_EXTERNAL_STAGING = 1 << 30;   // lmx_ctx_desc.flags, internal: no per-set pinned staging of its own
int ctx_num_sets(const lmx_ctx* c);
int ctx_next_set(const lmx_ctx* c);
size_t ctx_stage_bytes(const lmx_ctx* c);
size_t ctx_bytes_per_frame(const lmx_ctx* c);                                      // one frame of every modality in the staging layout
lmx_status ctx_check_sources(lmx_ctx* c, int n_frames, const lmx_image* sources, int n_sources, int max_frames /* 0 = the context's max_batch */);
// stride_frames = frames per modality block of the staging area (0 = the context's max_batch; a group with frame groups stages the WHOLE batch
// once and every member transfers its slice [first_frame, first_frame + n_frames) of it, n_frames may be 0)
void ctx_stage_sources(lmx_ctx* c, CopyPool* pool, uint8_t* base, int n_frames, const lmx_image* sources, int stride_frames);
lmx_status ctx_begin_staged_upload(lmx_ctx* c);                                    // host: the next set's previous transfer has left the staging area
lmx_status ctx_finish_staged_upload(lmx_ctx* c, int n_frames, const uint8_t* pinned, int first_frame, int stride_frames);  // queue the DMAs out of `pinned`, make the set current
lmx_status ctx_prepare_graph(lmx_ctx* c, int n_frames, float threshold);           // LMX_CTX_HIPGRAPH: capture the next enqueue's chain now if it is not cached
lmx_status ctx_drop_newest(lmx_ctx* c);                                            // undo the most recent enqueue (waits for it, frees its slot)

// ---- device-side geometry --------------------------------------------------------------------------------
constexpr int kMaxLevels = 4;
constexpr int kMaxModalities = 4;
constexpr int SB_GROUPS = 5, SB_BLOCK = 16, SB_MAX_BLOCKS = 6;   // scalar-block table of k_score_coarse_sb: <= 30 groups per template
constexpr int kFeatStride = 64;  // feature-table entries per (template, modality, level); upstream caps features at 63

struct LevelGeom {
  int32_t W, H;          // image size at this level
  int32_t T;             // sampling step
  int32_t Wc, Hc;        // W/T, H/T  (linear-memory "width"/"height")
  uint32_t cells;        // Wc*Hc      (length of one linear memory)
  uint32_t ori_stride;   // bytes per orientation block: T*T*cells + zero pad, multiple of 256
  uint32_t mod_stride;   // bytes per (frame, modality) at this level: 8*ori_stride + tail pad
  uint32_t zero_off;     // offset (within a modality block) of a run of >= cells+4096 zero bytes
  // nibble-packed memories of the coarsest level (two responses per byte), read by k_score_coarse:
  //   orientation o, byte i  =  elem(2i) | elem(2i + 1) << 4,  elem = the orientation's flat T*T*cells array, zero past its end.
  //   A feature whose first element index e0 is not a multiple of 8 starts in the middle of a dword: the kernel loads aligned
  //   dwords and funnel-shifts by 4 * (e0 & 7) bits with the neighbour lane's dword (v_alignbit_b32), so one copy serves
  //   every alignment.  Table entry = (dword index << 3) | (e0 & 7).
  uint32_t nib_ori_stride;    // bytes per orientation block incl. zero pad, multiple of 256
  uint32_t nib_mod_stride;    // bytes per (frame, modality): 8 * nib_ori_stride + tail pad
  uint32_t nib_zero_off;      // byte offset (multiple of 4) of a zero run (>= cells/2 + 2048 bytes) inside the block
  // finer levels keep NO response maps: only the spread image in linearize() order, one byte per cell.  k_refine derives the
  // 0..4 response of a feature's orientation from the spread byte with four nested bit masks (it touches a few hundred bytes
  // per candidate, so 8x fewer bytes are written and kept per frame than with materialised linear memories).
  uint32_t ls_stride;         // bytes per (frame, modality), multiple of 256
  uint32_t ls_zero_off;       // start of a zero run large enough for one patch
  // Banded form of that image (ls_bands > 0; Wc % 16 == 0, Wc >= 32).  View upstream's linear memories of one (frame, modality)
  // as ONE matrix of R = T*T*Hc rows x Wc columns (row = grid * Hc + cell row; flat index = row * Wc + column, a column past
  // Wc continues in the next row exactly as in the flat array).  Band k keeps columns 16k .. 16k+31 of every row in 32 bytes:
  //     byte (k * ls_band_stride + (row + 1) * 32 + c)  =  flat element row * Wc + 16k + c,      0 <= c < 32,
  // so every cell is stored twice and the 16 x 16 patch k_refine gathers for a feature (any origin) is 16 consecutive 32-byte
  // rows of one band: 4-5 cache lines instead of 16-17 (the gathers miss L2; round 2 measured the flat form at 5.4 TB/s of HBM
  // fetches, 730 lines per candidate).  Row 0 of a band is slack for the writer, rows R+1 .. R+16 stay zero.
  uint32_t ls_bands;          // 0: flat form
  uint32_t ls_band_stride;    // bytes per band: (R + 17) * 32
};

// Coarse candidate written by k_score_coarse, consumed by k_refine.
struct Candidate {
  uint32_t g;      // shard-local template index (all classes concatenated)
  uint32_t pos;    // raster index r*Wc + c at the coarsest level
  uint32_t raw;    // raw similarity (sum of responses)
  uint32_t frame;  // frame of the batch (candidates of all frames share one list)
};

// The candidate list of one output slot, filled by the scoring kernel and read by k_refine.  Appending through ONE counter costs
// 11.3 ns per reservation however many waves there are: device-scope atomics on one 128-byte line serialise (and counters a few bytes
// apart share the line; profiles/r03_atomic_append_microbench.txt), which on rendered banks -- thousands of candidates per frame -- was
// most of the scoring kernel's time.  So the list is striped: kCandStripes counters, each on a line of its own, in front of the slot's
// 64-byte header; a wave reserves all candidates of a chunk with one atomic on the stripe (its id + its append number) mod
// kCandStripes.  Stripe s owns entries [s * SC, (s + 1) * SC), SC = cap / kCandStripes; what does not fit its stripe goes to a spill
// region of `cap` entries behind the stripes through counter kCandStripes.  A stripe counter counts every candidate sent to it, fitted
// or spilled, so  sum of the stripe counters = the number of candidates, and when that is <= cap nothing was dropped (the spill region
// alone holds cap): the overflow rule "candidates > cap" is the one a single list had.  k_refine walks stripes and spill in order and
// writes the total to header word 0, where publish / export / the host find it.  The first workgroup of a batch's first kernel zeroes
// the counters together with the header (`clear16`).
constexpr int kCandStripes = 64;
constexpr int kStripeWords = 32;   // dwords between two counters: one 128-byte line each
constexpr size_t kStripeAreaBytes = (size_t)(kCandStripes + 1) * kStripeWords * 4;
inline size_t cand_list_entries(uint32_t cap) { return (size_t)cap * 2; }   // stripes (n * (cap / n) <= cap for any stripe count n) + spill
inline uint32_t* stripes_of_header(uint32_t* header) { return header - (size_t)(kCandStripes + 1) * kStripeWords; }
inline const uint32_t* stripes_of_header(const uint32_t* header) { return header - (size_t)(kCandStripes + 1) * kStripeWords; }

// Exception barrier of the C ABI: nothing may unwind into a C caller.  The readers size containers from what a file says, so a damaged
// or hostile file can make them throw (std::bad_alloc, std::length_error); the entry points that parse files, build banks and contexts
// or size vectors from device counters run their bodies through this (found by scripts/fuzz_files.py: a mutant of a renderer-params file
// ended the process with std::terminate).
template <typename F>
inline lmx_status guarded(const char* what, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    set_error("%s: out of memory (a size in the input is implausibly large?)", what);
    return LMX_ERR_INVALID_ARG;
  } catch (const std::exception& e) {
    set_error("%s: %s", what, e.what());
    return LMX_ERR_INVALID_ARG;
  } catch (...) {
    set_error("%s: unknown exception", what);
    return LMX_ERR_INVALID_ARG;
  }
}

// Fine-level feature table entry (refinement needs x,y for upstream's out-of-bounds skip).
struct FeatEntry {
  uint32_t off;    // finer levels: label << 29 | (grid_row*cells + lm_index) into the linearised spread image
  int16_t x, y;
};

struct TemplateInfo {    // per shard-local template g
  int32_t class_index;
  int32_t template_id;   // id within its class (global, not shard-local)
  int32_t class_slot;    // unused on device; slot is taken from the per-call class_slot table
  int32_t pad;
};

struct TemplateLevelInfo {  // per (g, level)
  int32_t width, height;    // of template l*M+0 (upstream uses tp[start] for the refinement clamp)
  int32_t nf_total;         // sum over modalities of features.size() at this level
  int32_t positions;        // template_positions at this level (only the coarsest is used)
};

struct ScoreInfo {           // per shard-local template: everything k_score_coarse_u8 needs, one 16-byte scalar load
  int32_t positions;        // template_positions at the coarsest level
  int32_t nf_total;         // features at the coarsest level, all modalities
  int32_t class_index;
  uint32_t groups;          // fast groups | all groups << 8 of the unified table row | blocks of the scalar-block row << 16
};

struct DeviceBankView {
  int32_t G;                        // templates in this shard
  int32_t L, M;
  const TemplateInfo* info;         // [G]
  const TemplateLevelInfo* linfo;   // [G][L]
  const uint32_t* coarse_off;       // [G][M][kFeatStride] offsets at level L-1, padded with zero_off
  const FeatEntry* feat;            // [L][G][M][kFeatStride] (levels 0..L-2 used by refine)
  const uint8_t* feat_count;        // [L][G][M]
  int32_t nf_max_coarse;            // max features over (g,m) at level L-1
  // Unified table of the coarsest level for the u8 scoring kernel: [G][kFeatStride] entries of ALL modalities of a template,
  // interleaved in groups of 3 (round robin over the modalities), each entry = (dword index << 3 | nibble shift) relative to
  // modality 0's memories of the frame (modality m's memories lie m * uni_mod_block_bytes behind them).  Valid (uni_ok) when
  // every template has at most 63 features at that level in total, so that a placement's sum (<= 252) fits a byte.
  const uint32_t* coarse_uni;
  // The same features as 16-dword blocks for the scalar side (k_score_coarse_sb): [G][SB_MAX_BLOCKS][SB_BLOCK]; per block 15 byte
  // offsets (5 groups x 3 features that share their funnel shift; < 3 leftovers of a shift class are padded with zero-run
  // entries), then one dword with the five shifts (5 bits each) and, from bit 25, the real features consumed so far.
  const uint32_t* coarse_blk;
  const ScoreInfo* sinfo;           // [G]
  int32_t uni_ok;
  int32_t score_variant;            // 0 generic, 1 u8, 2 sb: chosen when the context is created (LMX_SCORE_KERNEL), used when uni_ok
  int32_t score_no_prune;           // LMX_SCORE_NO_PRUNE (read when the context is created): the scoring kernel skips its exact early exits and does
                                    // similarity()'s full work -- identical candidates, data-independent cost (bench.py extra.score_full_work)
  uint32_t uni_mod_block_bytes;     // distance between consecutive modalities' nibble memories (max_batch * nib_mod_stride)
};

struct FrameBuffers {               // device pointers, frame-major with fixed per-frame strides
  uint8_t* lm[kMaxLevels][kMaxModalities];     // byte linear memories (coarsest level only), stride geom[l].mod_stride per frame
  uint8_t* ls[kMaxLevels][kMaxModalities];     // linearised spread images (finer levels), stride geom[l].ls_stride per frame
  uint8_t* quant[kMaxLevels][kMaxModalities];  // quantized label images, stride W_l*H_l per frame
  uint8_t* lmn[kMaxModalities];                // nibble-packed memories of the coarsest level, stride nib_mod_stride
};

struct KernelParams {
  LevelGeom geom[kMaxLevels];
  FrameBuffers fb;
};

// kernel ids for profiling
enum KernelId {
  K_PRE = 0,  // node-side pre-processing (k_pre_color / k_pre_depth), only with lmx_ctx_upload_raw
  K_COLOR_QUANTIZE,
  K_DEPTH_QUANTIZE,
  K_NN_DOWN,
  K_SPREAD_LINEARIZE,
  K_PACK_NIBBLES,
  K_SCORE_COARSE,
  K_REFINE,
  K_COUNT
};

// The depth quantiser's table on the device: the bank's NORMAL_LUT as median bins plus ONE trailing zero entry, the address of every pixel whose
// bin is 0 without a look-up (far, no valid neighbours, index past the table) -- the kernel's look-up is then an unconditional load.
constexpr size_t kNormalBinsDeviceBytes = (size_t)LMX_NORMAL_LUT_SIZE + 1;

// Streamed input of the one-frame call (lmx_match with a fresh host frame, the reference's own pattern: ..._service.cpp:339-344).  The
// quantisers of level 0 are launched BEFORE the host has written the frame into the frame set's host-visible device buffer; the calling thread
// then stores the rows band by band (non-temporal stores through the PCIe BAR) and publishes, after each band, how many rows have landed in a
// flag word that travels the same posted-write path (so it cannot overtake the rows).  A workgroup waits for the rows its tile reads -- thread 0
// polls the flag with system-scope acquire loads, the rest of the workgroup sits at a barrier -- and the kernel's launch latency and the
// transfer overlap instead of adding up.  The wait is BOUNDED (wall clock): if the rows never arrive the workgroup sets *fail and leaves without
// touching its tile, the chain behind it runs on whatever the buffers hold, and collect() reports the batch as failed instead of hanging.
//   flag word = seq << 20 | rows stored so far (all frames of the batch counted through); seq tells this call's stores from the previous call's
// Two threads may store one modality from both ends (round 4: the colour frame first, by both, then the depth frame): the second word
// `flag_hi` = seq << 20 | first row of the part stored from the bottom.  A tile reading rows [lo, hi) may start when hi <= rows from the top,
// or lo >= first row from the bottom, or the two fronts have met.
struct StreamWait {
  const uint32_t* flag = nullptr;   // device-visible (the frame set's fine-grained buffer); null = the frames are already there
  const uint32_t* flag_hi = nullptr;  // null = stored from the top only
  uint32_t seq = 0;
  uint32_t timeout_ticks = 0;       // of the 100 MHz wall clock
  uint32_t* fail = nullptr;         // set to 1 on a timeout: word 6 of the output slot's header
};

// ---- launchers (lmx_kernels.hip) -------------------------------------------------------------------------
void launch_color_quantize(hipStream_t s, const uint8_t* bgr, uint8_t* quant, uint8_t* pyr_next /* may be null */, int H, int W,
                           int n_frames, float weak_threshold, float* mag_out = nullptr /* trainer: squared magnitude per pixel */,
                           uint32_t* clear16 = nullptr /* 16 dwords zeroed by the first workgroup: the output slot's header */,
                           const StreamWait* wait = nullptr /* small batches: the frame is still being stored by the host */);
lmx_status train_add_template(lmx_bank* bank, int device, const lmx_image* sources, int n_sources, const char* class_id,
                              const lmx_image* object_mask, int32_t* template_id, int32_t* bounding_box);
void launch_depth_quantize(hipStream_t s, const uint16_t* depth, uint8_t* quant, uint8_t* quant_half, int H, int W, int n_frames, int distance_threshold,
                           int difference_threshold, const uint8_t* lut_bins /* device, [LMX_NORMAL_LUT_SIZE] median bins */,
                           uint32_t* clear16 = nullptr);
// device form of "finalise + cluster" (lmx_f2.hip)
constexpr int F2_MAX = 2048;   // records per frame the LDS path takes; larger frames are finished by the host
struct F2Params {
  const lmx_raw_match_t* recs;   // the slot's records (all frames of the batch)
  const uint32_t* hdr;           // the slot's header: [1] = records written
  uint32_t cap;
  int32_t n_frames;
  lmx_match_t* out_matches;      // [n_frames][F2_MAX] final matches (std::sort + std::unique applied)
  uint32_t* out_counts;          // [n_frames][4]: final matches, clusters, members, status (0 ok, 1 too many records, 2 side-car/range)
  lmx_cluster_t* out_clusters;   // [n_frames][F2_MAX]
  int32_t* out_members;          // [n_frames][F2_MAX]
  uint8_t* scratch;              // [n_frames][F2_MAX] x 32 bytes: per-cluster score, range, rect
  const double* dists;           // side-car: obj_origin_dists[n_templates]
  const int32_t* rects;          // side-car: rects[n_templates][4]
  uint32_t n_templates;
  int32_t step, size_thresh, do_clusters;
  double radius_min, radius_step;
};
void launch_f2(hipStream_t s, const F2Params& p);
void launch_debug_block_sort(hipStream_t s, const float* sim, const int* tid, int n, int* perm, unsigned long long* spill);

struct PullEntry { uint64_t src; uint64_t row_stride; };  // one caller-owned pinned image (device-visible address)
void launch_pull_frames(hipStream_t s, const PullEntry* tab /* device-visible */, uint8_t* dst, size_t frame_bytes, int rows, uint32_t row_bytes,
                        int n_frames);
void launch_publish_records(hipStream_t s, void* dst, const void* src, uint32_t max_records, uint32_t cand_cap);
void launch_copy_bytes(hipStream_t s, void* dst, const void* src, size_t bytes);
void launch_publish_blocks(hipStream_t s, void* dst, const void* src, int n_blocks, size_t block_bytes, uint32_t max_records);
constexpr int kPullMax = 16;
struct PullSources { const void* src[kPullMax]; };   // passed by value: the source blocks of k_pull_blocks
void launch_pull_blocks(hipStream_t s, void* dst, const PullSources& srcs, int n_blocks, size_t block_bytes, uint32_t max_records);
void launch_nn_down2(hipStream_t s, const uint8_t* src, uint8_t* dst, int Hd, int Wd, int n_frames);
void launch_apply_mask(hipStream_t s, uint8_t* quant, const uint8_t* mask0, int Hl, int Wl, int W0, int H0, int level, int n_frames);
bool spread_writes_nibbles(const LevelGeom& g);
int score_kernel_variant(const DeviceBankView& bank);
void launch_spread_linearize(hipStream_t s, const uint8_t* quant, uint8_t* lm /* coarsest level, byte form */, uint8_t* ls /* finer levels */,
                             uint8_t* lmn /* coarsest level, nibble form */, const LevelGeom& g, int n_frames);
// All modalities of one level in ONE launch (the same kernels, blockIdx.y = modality).  Returns false when the level has no
// fast kernel (generic T / widths): the caller then launches per modality with launch_spread_linearize.
struct SpreadBatch {
  const uint8_t* quant[kMaxModalities];
  uint8_t* lm[kMaxModalities];
  uint8_t* ls[kMaxModalities];
  uint8_t* lmn[kMaxModalities];
};
bool launch_spread_linearize_all(hipStream_t s, const SpreadBatch& b, int n_modalities, const LevelGeom& g, int n_frames);
void launch_pre_color(hipStream_t s, const uint8_t* src, uint8_t* dst, int SH, int SW, int SC, int H, int W, int crop_x, int crop_y, int blur3,
                      int n_frames);
void launch_pre_depth(hipStream_t s, const void* src, uint16_t* dst, int SH, int SW, int H, int W, int crop_x, int crop_y, int is_float,
                      int n_frames);
void launch_debug_orientation_label(hipStream_t s, const short* dx, const short* dy, uint8_t* out, size_t n);
void launch_pack_nibbles(hipStream_t s, const uint8_t* lm, uint8_t* lmn, const LevelGeom& g, int n_frames);
void launch_score_coarse(hipStream_t s, const DeviceBankView& bank, const LevelGeom& g, const uint8_t* const* lm_mod /*[M] device ptrs*/,
                         int n_frames, float threshold, const int32_t* class_slot, Candidate* cands /* cand_list_entries(cap) */,
                         uint32_t* header /* the slot's; its stripe counters lie in front of it (stripes_of_header) */, uint32_t cap,
                         int n_stripes /* power of two <= kCandStripes, the same for launch_refine */);
// Returns false when nothing was launched (empty shard).  pub_dst != null: the kernel also publishes the slot (header + counted
// records, <= pub_max) to pub_dst when its last workgroup finishes; pub_counter is that slot's zero-initialised ticket counter.
bool launch_refine(hipStream_t s, const DeviceBankView& bank, const KernelParams& kp, int n_frames, float threshold,
                   const int32_t* class_slot, const Candidate* cands, uint32_t* header, uint32_t cap, int n_stripes,
                   lmx_raw_match_t* matches, uint32_t* match_count, void* pub_dst = nullptr, const void* pub_src = nullptr, uint32_t* pub_counter = nullptr,
                   uint32_t pub_max = 0, uint32_t pub_seq = 0 /* written to word 7 of the published header LAST: the host may poll it instead of the slot's event */);
// Fused launches of the small-batch chain (lmx_enqueue.cpp issue_small): depth quantiser of level 0 + colour quantiser of level 1, and the
// spread of both levels of a two-level bank.
bool launch_small_depth_color(hipStream_t s, const uint16_t* depth, uint8_t* dq, uint8_t* dq_half, int H, int W, int distance_threshold, int difference_threshold,
                              const uint8_t* lut_bins, const uint8_t* bgr1, uint8_t* cq1, uint8_t* pyr2, int H1, int W1, float weak_threshold, int n_frames,
                              const StreamWait* wait = nullptr /* the depth frame is still being stored by the host */);
bool launch_small_spread(hipStream_t s, const SpreadBatch& b0, const LevelGeom& g0, const SpreadBatch& b1, const LevelGeom& g1, int n_mod, int n_frames);

}  // namespace lmx
