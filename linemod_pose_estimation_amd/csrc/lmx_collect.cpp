// liblmx.so, behind the kernel chain: read-back of an output slot and the host finalisation (restore upstream insertion order, then the
// very std::sort / std::unique Detector::match applies), the device-side consumer chain (lmx_ctx_collect_clusters), and the
// multi-GPU plumbing on one context's side: gather-block export, copies by kernel, the host merge of gathered blocks.

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

namespace {

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

}  // namespace

extern "C" {

// sync + read-back + per-frame finalisation shared by collect / collect_flat
static lmx_status collect_impl(lmx_ctx* c, int32_t n_frames, std::vector<std::vector<HostMatch>>& fin) {
  if (c->outstanding < 1) { set_error("lmx_ctx_collect: nothing enqueued"); return LMX_ERR_INVALID_ARG; }
  const int slot = (c->head + c->n_slots - c->outstanding) % c->n_slots;  // oldest outstanding enqueue
  if (n_frames != c->slot_frames[slot]) { set_error("lmx_ctx_collect: n_frames=%d but the enqueue had %d", n_frames, c->slot_frames[slot]); return LMX_ERR_INVALID_ARG; }
  LMX_HIP(hipSetDevice(c->device));
  const size_t first = std::min<size_t>(c->h_out_records, lmx_ctx::kFirstSlice);
  using clk = std::chrono::steady_clock;
  const clk::time_point t0 = clk::now();
  bool seen = false;
  if (c->pub_seq[slot] != 0 && !c->env_no_header_poll && c->profiling == 0) {
    // one or two frames per call: k_refine's last workgroup publishes the slot and then its sequence number (word 7 of the pinned header);
    // spinning on that word sees the result a few microseconds before the event's wake-up does.  Bounded: 200 us, then the event as always
    const volatile uint32_t* word = reinterpret_cast<const volatile uint32_t*>(c->h_out_slot[slot]) + 7;
    const uint32_t want = c->pub_seq[slot];
    for (;;) {
      if (*word == want) { seen = true; break; }
      if (clk::now() - t0 > std::chrono::microseconds(200)) break;
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!seen) LMX_HIP(hipEventSynchronize(c->done[slot]));
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
  if (reinterpret_cast<uint32_t*>(c->h_out)[6] != 0) {
    // a workgroup of a level-0 quantiser gave up waiting for rows of a streamed frame store (StreamWait): the batch ran on incomplete input
    set_error("the input frame never reached the device: a quantiser waited %u us for the host's stores (streamed upload of lmx_match); nothing was matched",
              c->stream_timeout_ticks / 100u);
    return LMX_ERR_HIP;
  }
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
  if (c->trace_match) {
    c->tm_acc[lmx_ctx::TM_WAIT] += std::chrono::duration<double>(t1 - t0).count();
    c->tm_acc[lmx_ctx::TM_FINALIZE] += std::chrono::duration<double>(clk::now() - t1).count();
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
  // the counters lmx_ctx_stats reports, as collect leaves them: the slot's pinned mirror holds the published header once its event has
  // completed (the sharded callers consume the records on the device and never collect; round 3 reported 0 candidates for them)
  const uint32_t* h_hdr = reinterpret_cast<const uint32_t*>(c->h_out_slot[slot]);
  c->stat_cands = h_hdr[0]; c->stat_matches = h_hdr[1];
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

lmx_status lmx_merge_gathered_groups(const void* blocks, int32_t n_ranks, size_t block_stride_bytes, size_t capacity_records, int32_t n_frames,
                                     int32_t frame_groups, lmx_match_t* out, size_t cap_total, size_t* offsets) {
  return lmx::guarded("lmx_merge_gathered", [&]() -> lmx_status {
  if (!blocks || !offsets || n_ranks < 1 || n_frames < 1 || (cap_total > 0 && !out)) { set_error("lmx_merge_gathered: invalid argument"); return LMX_ERR_INVALID_ARG; }
  const int G = frame_groups < 1 ? 1 : frame_groups;
  if (n_ranks % G != 0) { set_error("lmx_merge_gathered_groups: %d ranks do not form %d frame groups", n_ranks, G); return LMX_ERR_INVALID_ARG; }
  const int R = n_ranks / G;
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
    // the rank's frames within the batch: its frame group's slice (the whole batch when G == 1); record frames are local to it
    const int g = r / R;
    const int f_begin = (int)((long)g * n_frames / G), f_count = (int)((long)(g + 1) * n_frames / G) - f_begin;
    const lmx_raw_match_t* recs = reinterpret_cast<const lmx_raw_match_t*>(blk + LMX_GATHER_HEADER_BYTES);
    for (uint32_t i = 0; i < n; ++i)
      if (recs[i].frame >= 0 && recs[i].frame < f_count) per_frame[f_begin + recs[i].frame].push_back(&recs[i]);
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

lmx_status lmx_merge_gathered(const void* blocks, int32_t n_ranks, size_t block_stride_bytes, size_t capacity_records, int32_t n_frames,
                              lmx_match_t* out, size_t cap_total, size_t* offsets) {
  return lmx_merge_gathered_groups(blocks, n_ranks, block_stride_bytes, capacity_records, n_frames, 1, out, cap_total, offsets);
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

}  // extern "C"
