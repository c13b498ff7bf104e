// liblmx.so, introspection: stage-level debug reads for the parity tests, counters, per-kernel HIP-event timing, the algorithmic-bytes
// figure of SURVEY.md 8(d), and the test hooks of the std::sort restatement (csrc/lmx_sort_emul.hpp, lmx_sort_block.hpp).

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
#include "lmx_sort_emul.hpp"

using namespace lmx;

static const char* kKernelNames[K_COUNT] = {"k_pre",            "k_color_quantize", "k_depth_quantize", "k_nn_down2",   "k_spread_linearize",
                                            "k_pack_nibbles",   "k_score_coarse",   "k_refine"};

extern "C" {

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
