// liblmx.so, device context: creation (geometry, the device-resident bank, per-lane workspaces, output slots), destruction, and the
// upload paths in front of the kernel chain (host frames, masks, raw camera frames); the hooks device groups use.
// Mirrors the call surface of cv::linemod::Detector as the reference uses it (/root/reference/src/rgbdDetector.cpp:31-34); see
// include/lmx.h for the per-function mapping.  There is no CPU compute path: without a usable HIP device every compute call fails.

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

namespace lmx {

static uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

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

bool get_events(lmx_ctx* c, hipEvent_t* a, hipEvent_t* b) {
  if (c->event_pool.empty()) {
    if (hipEventCreate(a) != hipSuccess) return false;
    if (hipEventCreate(b) != hipSuccess) return false;
    return true;
  }
  *a = c->event_pool.back().first; *b = c->event_pool.back().second;
  c->event_pool.pop_back();
  return true;
}

void drain_profiling(lmx_ctx* c) {
  for (const ProfEvent& e : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) { c->k_ms[e.kernel] += ms; c->k_launches[e.kernel] += 1; }
    c->event_pool.emplace_back(e.start, e.stop);
  }
  c->pending.clear();
}

// Points the context's working view (kp.fb, derived colour pyramid levels, candidate list) at one lane's buffers.
void select_lane(lmx_ctx* c, int lane) {
  c->kp.fb = c->lane_fb[lane];
  for (int m = 0; m < c->M; ++m)
    for (int l = 1; l < c->L; ++l) c->mb[m].bgr[l] = c->lane_bgr[lane][m][l];
  c->d_cands = c->lane_cands[lane];
}

// Points the level-0 frame pointers at one frame set.
void select_set(lmx_ctx* c, int set) {
  c->cur_set = set;
  const lmx_ctx::FrameSet& fs = c->sets[set];
  for (int m = 0; m < c->M; ++m) {
    c->mb[m].bgr[0] = fs.stored && fs.bgr[m] ? fs.store_buf[m] : fs.bgr[m];
    c->mb[m].depth = fs.stored && fs.depth[m] ? reinterpret_cast<uint16_t*>(fs.store_buf[m]) : fs.depth[m];
  }
}

// Host-side wait for everything queued on every lane and on the copy stream.
lmx_status sync_lanes(lmx_ctx* c) {
  if (c->copy_stream) LMX_HIP(hipStreamSynchronize(c->copy_stream));
  if (c->pre_stream) LMX_HIP(hipStreamSynchronize(c->pre_stream));
  for (int lane = 0; lane < c->n_lanes; ++lane) LMX_HIP(hipStreamSynchronize(c->lane_stream[lane]));
  return LMX_OK;
}

int upload_threads(const lmx_ctx* c) {
  if (c->env_upload_threads >= 1) return c->env_upload_threads;
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(8u, hw ? hw / 2 : 1u));
}


// One modality's frames written straight into the frame set's host-visible device buffers (see FrameSet::store_buf).
void store_modality(lmx_ctx* c, lmx_ctx::FrameSet& fs, int m, int n_frames, const lmx_image* sources) {
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

}  // namespace lmx

// The same frames written band by band, the progress word behind every band: what the waiting workgroups of the level-0 quantiser poll.
// Bands are numbered through the frames of the batch; `end` says who takes which (lmx_ctx.hpp).
void lmx_ctx::store_modality_streamed(lmx_ctx::FrameSet& fs, int m, int n_frames, const lmx_image* sources, uint32_t seq, int end) {
  lmx_ctx* c = this;
  const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
  const size_t row_bytes = (size_t)c->desc.width * (cg ? 3 : 2);
  const int H = c->desc.height;
  // rows per progress update.  Every update costs two store fences (15 updates per modality at 32 rows: +5.5 us per 640x480 RGB-D frame, measured);
  // what streaming hides is the kernel's launch latency and everything but its last band, and the last band's tiles run in one round of
  // workgroups whatever its height
  const int band = c->stream_band_rows;
  const int per_frame = (H + band - 1) / band, n_bands = per_frame * n_frames;
  uint32_t* flag = fs.store_flag + 32 * m + (end < 0 ? 16 : 0);
  for (int k = 0;; ++k) {
    int b = k;
    if (end != 0) {
      if (c->stream_claim[m].fetch_add(1, std::memory_order_relaxed) >= n_bands) break;
      b = end > 0 ? k : n_bands - 1 - k;   // this thread's k-th claim: the fronts cannot cross, the counter hands out n_bands claims in all
    } else if (k >= n_bands) {
      break;
    }
    const int f = b / per_frame, y = (b - f * per_frame) * band, y1 = std::min(H, y + band);
    const lmx_image& im = sources[(size_t)f * c->M + m];
    uint8_t* dst = fs.store_buf[m] + (size_t)f * c->frame_bytes[m];
    if (im.row_stride_bytes == row_bytes) lmx::stream_copy(dst + (size_t)y * row_bytes, (const uint8_t*)im.data + (size_t)y * row_bytes, row_bytes * (size_t)(y1 - y));
    else
      for (int r = y; r < y1; ++r) lmx::stream_copy(dst + (size_t)r * row_bytes, (const uint8_t*)im.data + (size_t)r * im.row_stride_bytes, row_bytes);
    lmx::stream_store_flag(flag, (seq << 20) | (uint32_t)(f * H + (end < 0 ? y : y1)));
  }
}
void lmx_ctx::stream_reset_hi(lmx_ctx::FrameSet& fs, int m, int n_frames, uint32_t seq) {
  lmx::stream_store_flag(fs.store_flag + 32 * m + 16, (seq << 20) | (uint32_t)(n_frames * desc.height));
  stream_claim[m].store(0, std::memory_order_relaxed);
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
    std::vector<uint8_t> bins(lmx::kNormalBinsDeviceBytes);   // zero-initialised: the trailing entry stays 0
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
    // progress words of streamed stores: one 128-byte line per modality, host-visible like the frames
    c->stream_ok = c->store_ok && std::getenv("LMX_NO_STREAM_STORE") == nullptr;
    for (int set = 0; set < c->n_sets && c->stream_ok; ++set) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, 128 * kMaxModalities, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); c->stream_ok = false; break; }
      c->allocs.push_back(p);
      c->sets[set].store_flag = static_cast<uint32_t*>(p);
      for (int m = 0; m < kMaxModalities; ++m) {
        lmx::stream_store_flag(c->sets[set].store_flag + 32 * m, 0u);        // rows from the top
        lmx::stream_store_flag(c->sets[set].store_flag + 32 * m + 16, 0u);   // first row of the part stored from the bottom (sequence 0: never a call's)
      }
    }
    if (const char* e = std::getenv("LMX_STREAM_TIMEOUT_US")) c->stream_timeout_ticks = (uint32_t)std::max(100L, std::min(std::atol(e), 20000000L)) * 100u;
    c->env_test_drop_stream = std::getenv("LMX_TEST_DROP_STREAM_STORE") != nullptr;
    if (const char* e = std::getenv("LMX_STREAM_BAND_ROWS")) c->stream_band_rows = (int)std::max(8L, std::min(std::atol(e), 4096L));
    c->trace_match = std::getenv("LMX_MATCH_TRACE") != nullptr;
  }
  select_set(c, 0);
  LMX_HIP(hipStreamSynchronize(c->stream));
  return LMX_OK;
}

extern "C" {

// ---- context ----------------------------------------------------------------------------------------------
void lmx_ctx_destroy(lmx_ctx* c) {
  if (!c) return;
  if (c->trace_match && c->tm_n > 0) {
    static const char* names[lmx_ctx::TM_COUNT] = {"upload (deferred)", "launch colour L0", "store colour", "launch depth L0 + colour L1", "store depth", "launch spread, score, refine",
                                                   "wait for the slot", "finalise"};
    std::fprintf(stderr, "lmx_match trace (%ld calls, streamed stores %s): host us per call:", c->tm_n, c->stream_ok ? "on" : "off");
    for (int i = 0; i < lmx_ctx::TM_COUNT; ++i) std::fprintf(stderr, " %s %.1f |", names[i], c->tm_acc[i] / (double)c->tm_n * 1e6);
    std::fprintf(stderr, "\n");
  }
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
  c->env_no_header_poll = std::getenv("LMX_NO_HEADER_POLL") != nullptr;
  c->env_no_launch_thread = std::getenv("LMX_NO_LAUNCH_THREAD") != nullptr;
  c->env_one_store_thread = std::getenv("LMX_ONE_STORE_THREAD") != nullptr;
  c->env_no_delegate_first = std::getenv("LMX_NO_DELEGATE_FIRST_LAUNCH") != nullptr;
  if (const char* e = std::getenv("LMX_UPLOAD_THREADS")) c->env_upload_threads = std::max(0, std::min(std::atoi(e), 64));
  {
    const char* e = std::getenv("LMX_SCORE_KERNEL");
    c->dbank.score_variant = (std::getenv("LMX_SCORE_GENERIC") != nullptr || (e && std::strcmp(e, "generic") == 0)) ? 0 : ((e && std::strcmp(e, "u8") == 0) ? 1 : 2);
    const char* np = std::getenv("LMX_SCORE_NO_PRUNE");
    c->dbank.score_no_prune = (np && *np && std::strcmp(np, "0") != 0) ? 1 : 0;
  }
  lmx_status st = ctx_create_impl(c);
  if (st != LMX_OK) { const std::string keep = lmx_last_error(); lmx_ctx_destroy(c); set_error("%s", keep.c_str()); return st; }
  *out = c;
  return LMX_OK;
  });
}

}  // extern "C"

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

extern "C" {

lmx_status lmx_ctx_upload(lmx_ctx* c, int32_t n_frames, const lmx_image* sources, int32_t n_sources) {
  return lmx::guarded("lmx_ctx_upload", [&]() -> lmx_status {
  if (!c || !sources) { set_error("lmx_ctx_upload: null argument"); return LMX_ERR_INVALID_ARG; }
  lmx_status st = lmx::ctx_check_sources(c, n_frames, sources, n_sources, 0);
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
size_t ctx_bytes_per_frame(const lmx_ctx* c) { return c->F > 0 ? c->h_stage_bytes / (size_t)c->F : 0; }

lmx_status ctx_check_sources(lmx_ctx* c, int n_frames, const lmx_image* sources, int n_sources, int max_frames) {
  if (max_frames < 1) max_frames = c->F;
  if (n_sources != c->M) {
    set_error("sources.size()=%d != modalities.size()=%d (upstream CV_Assert in Detector::match)", n_sources, c->M);
    return LMX_ERR_SHAPE;
  }
  if (n_frames < 1 || n_frames > max_frames) { set_error("n_frames=%d outside [1,%d]", n_frames, max_frames); return LMX_ERR_INVALID_ARG; }
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
// sum_{m' < m} frame_bytes[m'] * stride_frames, frames back to back, rows packed; stride_frames = the context's max_batch, or the group's
// when a device group stages a batch for members that each take a slice of it): one task per image for batches, row bands for a few
// frames (a single 640x480 RGB-D frame is still 1.5 MB: 60 us on one thread).
void ctx_stage_sources(lmx_ctx* c, CopyPool* pool, uint8_t* base, int n_frames, const lmx_image* sources, int stride_frames) {
  if (stride_frames < 1) stride_frames = c->F;
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
    off += c->frame_bytes[m] * (size_t)stride_frames;
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

lmx_status ctx_finish_staged_upload(lmx_ctx* c, int n_frames, const uint8_t* pinned, int first_frame, int stride_frames) {
  LMX_HIP(hipSetDevice(c->device));
  if (stride_frames < 1) stride_frames = c->F;
  if (n_frames < 0 || n_frames > c->F || first_frame < 0 || first_frame + n_frames > stride_frames) { set_error("ctx_finish_staged_upload: frames [%d, %d) outside the staged batch", first_frame, first_frame + n_frames); return LMX_ERR_INVALID_ARG; }
  const int set = (c->cur_set + 1) % c->n_sets;
  lmx_ctx::FrameSet& fs = c->sets[set];
  fs.stored = false;
  size_t off = 0;
  for (int m = 0; m < c->M; ++m) {
    const bool cg = c->bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    uint8_t* dst = cg ? fs.bgr[m] : reinterpret_cast<uint8_t*>(fs.depth[m]);
    // a member of a frame group takes its slice [first_frame, first_frame + n_frames) of the staged batch; none at all when the batch
    // has fewer frames than groups (the set still becomes current: the members' sets advance in lock step)
    if (n_frames > 0)
      LMX_HIP(hipMemcpyAsync(dst, pinned + off + (size_t)first_frame * c->frame_bytes[m], c->frame_bytes[m] * n_frames, hipMemcpyHostToDevice, c->copy_stream));
    off += c->frame_bytes[m] * (size_t)stride_frames;
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

}  // extern "C"
