// liblmx.so, the per-batch kernel chain of Detector::match (/root/reference/src/rgbdDetector.cpp:33): the plain chain in two stages,
// the five-launch chain for one or two frames per call, hipGraph capture / replay, lmx_ctx_enqueue, and the synchronous composites
// lmx_match / lmx_match_batch / lmx_match_masked (the drop-in calls).

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <cctype>
#include <cmath>
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

#include "lmx_ctx.hpp"

using namespace lmx;

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
    c->pub_seq[slot] = 0;
    uint32_t seq = 0;
    if (fold && !(c->desc.flags & LMX_CTX_HIPGRAPH)) {   // a captured chain would replay the sequence number of its capture: graph contexts wait on the event
      c->pub_seq_counter = c->pub_seq_counter == 0xffffffffu ? 1u : c->pub_seq_counter + 1u;
      seq = c->pub_seq_counter;
    }
    ScopedKernel k(c, K_REFINE);
    published = launch_refine(s, c->dbank, c->kp, n_frames, threshold, c->d_class_slot, c->d_cands, c->d_cand_count(), c->cap_total, c->stripes_for(n_frames), c->d_records(),
                              c->d_match_count(), fold ? c->h_out_dev[slot] : nullptr, c->d_out, c->d_pub_counter + slot,
                              (uint32_t)std::min<size_t>(c->h_out_records, lmx_ctx::kFirstSlice), seq);
    published = published && fold;
    if (published) c->pub_seq[slot] = seq;
  }
  if (!published) launch_publish_records(s, c->h_out_dev[slot], c->d_out, (uint32_t)std::min<size_t>(c->h_out_records, lmx_ctx::kFirstSlice), c->cap_total);
  LMX_HIP(hipGetLastError());
  return LMX_OK;
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
  // Streamed stores (StreamWait, lmx_internal.hpp): the level-0 quantiser of a modality is launched first and its workgroups wait for the rows
  // the calling thread stores behind the launch, so the launch latency and the transfer overlap; without it the frame is stored, then the kernel
  // launched (colour), respectively stored while the colour kernel runs (depth).
  using clk = std::chrono::steady_clock;
  clk::time_point tp = clk::now();
  auto lap = [&](int phase) {
    if (!c->trace_match) return;
    const clk::time_point now = clk::now();
    c->tm_acc[phase] += std::chrono::duration<double>(now - tp).count();
    tp = now;
  };
  const bool stream = sources != nullptr && c->stream_ok && (size_t)n_frames * c->desc.height < (1u << 20);
  StreamWait wc, wd;
  const bool use_helper = stream && !c->env_no_launch_thread && c->profiling == 0;
  const bool two_ended = use_helper && !c->env_one_store_thread;   // both threads store, each modality from both ends
  if (stream) {
    c->stream_seq = (c->stream_seq % 4095u) + 1u;   // 1 .. 4095: never the value the flag words were initialised with
    wc.flag = fs.store_flag; wc.seq = c->stream_seq; wc.timeout_ticks = c->stream_timeout_ticks; wc.fail = reinterpret_cast<uint32_t*>(c->d_out) + 6;
    wd = wc;
    wd.flag = fs.store_flag + 32;
    if (two_ended) {
      // "nothing from the bottom yet", written (and fenced) by this thread before the helper exists for this call: it cannot overtake the helper's
      // first real update
      wc.flag_hi = wc.flag + 16; wd.flag_hi = wd.flag + 16;
      for (int m = 0; m < c->M; ++m) c->stream_reset_hi(fs, m, n_frames, c->stream_seq);
    }
  }
  // the second launch: depth L0 + colour L1 in one grid, or colour L1 alone
  auto launch_second = [&]() {
    if (c->M == 2) {
      const lmx_modality_desc& dn = c->bank->mods[1];
      ScopedKernel k(c, K_DEPTH_QUANTIZE);
      launch_small_depth_color(s, c->mb[1].depth, c->kp.fb.quant[0][1], c->kp.fb.quant[1][1], g0.H, g0.W, dn.distance_threshold, dn.difference_threshold, c->d_normal_bins,
                               c->mb[0].bgr[1], c->kp.fb.quant[1][0], nullptr, g1.H, g1.W, cg.weak_threshold, n_frames, stream ? &wd : nullptr);
    } else {
      ScopedKernel k(c, K_COLOR_QUANTIZE);
      launch_color_quantize(s, c->mb[0].bgr[1], c->kp.fb.quant[1][0], nullptr, g1.H, g1.W, n_frames, cg.weak_threshold, nullptr, nullptr);
    }
  };
  // spread of both levels, score, refine (+ read-back)
  auto launch_rest = [&]() -> lmx_status {
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
  };

  if (sources && !stream) { store_modality(c, fs, 0, n_frames, sources); lap(lmx_ctx::TM_STORE_COLOR); }
  auto launch_first = [&]() {
    ScopedKernel k(c, K_COLOR_QUANTIZE);
    launch_color_quantize(s, c->mb[0].bgr[0], c->kp.fb.quant[0][0], c->mb[0].bgr[1], g0.H, g0.W, n_frames, cg.weak_threshold, nullptr, reinterpret_cast<uint32_t*>(c->d_out),
                          stream ? &wc : nullptr);
  };
  // With a helper thread the FIRST launch is its job too and this thread starts storing at once: a launch costs ~7 us (~120 us when the call finds
  // the device idle after a pause), the first tiles need the first band of rows anyway.  In a loop: 84.9 against 88.4 us per call; after a
  // one-second pause the call stays at 190-270 us either way (profiles/r04_single_frame_latency.txt, r04M).  LMX_NO_DELEGATE_FIRST_LAUNCH=1: as before.
  const bool delegate_first = use_helper && !c->env_no_delegate_first;
  if (!delegate_first) launch_first();
  lap(lmx_ctx::TM_LAUNCH_COLOR);
  if (use_helper) {
    // With streamed stores the call is HOST-bound: ~45 us of stores and ~15 us of launches on one thread, the device waiting for both.  The rest
    // of the chain is queued by a helper thread (one persistent worker, woken here) while this thread stores the frame: the waiting kernels
    // make the order of "launch" and "store" irrelevant, the stream keeps the kernels in order (the helper's launches all come behind the
    // colour kernel's, which is already queued).
    if (!c->launch_helper) c->launch_helper.reset(new lmx::LaunchHelper());
    lmx_status rest_st = LMX_OK;
    std::string rest_msg;
    // ... and then it helps with the stores: two cores' write-combining buffers fill the PCIe link better than one (one thread moves a frame at
    // ~36 GB/s out of the caller's memory, the link takes ~45).  Both threads take the COLOUR frame first -- this thread from the top, the helper,
    // once its launches are out, from the bottom -- and then the depth frame the same way: the colour chain is one kernel longer (level 1 is
    // quantised from level 0's pyrDown), so it should not be the one that ends with the last byte of the call.
    const bool store_depth = c->M == 2 && !c->env_test_drop_stream;
    const int end = two_ended ? 1 : 0;
    const std::function<void()> job = [&]() {
      if (hipSetDevice(c->device) != hipSuccess) { rest_st = LMX_ERR_HIP; rest_msg = "hipSetDevice failed on the launch thread"; return; }
      if (delegate_first) launch_first();
      launch_second();
      rest_st = launch_rest();
      if (rest_st != LMX_OK) rest_msg = lmx_last_error();   // thread-local on the helper
      if (two_ended) {
        c->store_modality_streamed(fs, 0, n_frames, sources, wc.seq, -1);
        if (store_depth) c->store_modality_streamed(fs, 1, n_frames, sources, wd.seq, -1);
      }
    };
    c->launch_helper->submit(&job);
    c->store_modality_streamed(fs, 0, n_frames, sources, wc.seq, end);
    if (store_depth) c->store_modality_streamed(fs, 1, n_frames, sources, wd.seq, end);
    c->launch_helper->wait();
    lap(lmx_ctx::TM_STORE_COLOR);
    if (rest_st != LMX_OK) { set_error("%s", rest_msg.c_str()); return rest_st; }
    return LMX_OK;
  }
  if (stream) { c->store_modality_streamed(fs, 0, n_frames, sources, wc.seq); lap(lmx_ctx::TM_STORE_COLOR); }
  if (c->M == 2 && sources && !stream) { store_modality(c, fs, 1, n_frames, sources); lap(lmx_ctx::TM_STORE_DEPTH); }   // lands while the colour kernel runs
  launch_second();
  lap(lmx_ctx::TM_LAUNCH_DEPTH);
  if (c->M == 2 && stream && !c->env_test_drop_stream) { c->store_modality_streamed(fs, 1, n_frames, sources, wd.seq); lap(lmx_ctx::TM_STORE_DEPTH); }
  const lmx_status pst = launch_rest();
  lap(lmx_ctx::TM_LAUNCH_REST);
  return pst;
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

lmx_status lmx_match_batch(lmx_ctx* c, int32_t n_frames, const lmx_image* sources, int32_t n_sources, float threshold,
                           const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out) {
  return lmx::guarded("lmx_match_batch", [&]() -> lmx_status {
  if (!c) { set_error("lmx_match: null context"); return LMX_ERR_INVALID_ARG; }
  std::lock_guard<std::recursive_mutex> lk(c->call_mutex);   // contexts handed out by lmx_ctx_acquire may be shared between threads
  c->deferred_sources = nullptr;
  c->deferred_frames = -1;   // "upload may leave the direct stores of a small batch to the enqueue below" (the sources outlive both calls)
  const std::chrono::steady_clock::time_point t_call = std::chrono::steady_clock::now();
  lmx_status st = lmx_ctx_upload(c, n_frames, sources, n_sources);
  if (c->trace_match) { c->tm_acc[lmx_ctx::TM_UPLOAD] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count(); c->tm_n += 1; }
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

}  // extern "C"
