// Trainer side of the bank: cv::linemod::Detector::addTemplate as the reference's trainers call it
// (/root/reference/src/renderer.cpp:308, src/renderer_only_image.cpp:266).
//
// Device: the per-pixel stages -- quantizedOrientations (+ squared magnitudes), quantizedNormals + medianBlur, the pyrDown /
// nearest-neighbour chains -- with the same kernels match() uses (no size constraint: linearize is not part of training).
// Host: extractTemplate's candidate ranking and the greedy scattered selection, cropTemplates.  Those are short sequential
// loops over a few thousand candidates with a data-dependent restart rule; they stay on the CPU like upstream.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "lmx_internal.hpp"

namespace lmx {
namespace {

#define TR_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorNoDevice ? LMX_ERR_NO_DEVICE : LMX_ERR_HIP;                     \
    }                                                                                      \
  } while (0)

struct Cand { int x, y, label; float score; };
struct Feat { int x, y, label; };

inline int label_of(uint8_t q) {  // upstream getLabel(): one-hot byte -> bin
  for (int k = 0; k < 8; ++k)
    if (q == (1u << k)) return k;
  return -1;
}

// 3x3 minimum with BORDER_REPLICATE (cv::erode with the default 3x3 rectangle)
std::vector<uint8_t> erode3(const std::vector<uint8_t>& m, int H, int W) {
  std::vector<uint8_t> out((size_t)H * W);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      uint8_t v = 255;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = std::min(std::max(y + dy, 0), H - 1);
        for (int dx = -1; dx <= 1; ++dx) v = std::min(v, m[(size_t)yy * W + std::min(std::max(x + dx, 0), W - 1)]);
      }
      out[(size_t)y * W + x] = v;
    }
  return out;
}

// distanceTransform(src, dst, DIST_C, 3): chessboard distance to the nearest zero pixel (two-pass 8-neighbour chamfer with unit
// weights, exact for that metric); a component without any zero pixel in the image keeps a huge value, like upstream's
// initial distance.
std::vector<float> dist_chessboard(const std::vector<uint8_t>& nz, int H, int W) {
  const int INF = 1 << 28;
  std::vector<int> d((size_t)H * W);
  for (size_t i = 0; i < d.size(); ++i) d[i] = nz[i] ? INF : 0;
  auto at = [&](int y, int x) -> int { return (y < 0 || y >= H || x < 0 || x >= W) ? INF : d[(size_t)y * W + x]; };
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int v = d[(size_t)y * W + x];
      v = std::min(v, std::min(std::min(at(y - 1, x - 1), at(y - 1, x)), std::min(at(y - 1, x + 1), at(y, x - 1))) + 1);
      d[(size_t)y * W + x] = v;
    }
  for (int y = H - 1; y >= 0; --y)
    for (int x = W - 1; x >= 0; --x) {
      int v = d[(size_t)y * W + x];
      v = std::min(v, std::min(std::min(at(y + 1, x + 1), at(y + 1, x)), std::min(at(y + 1, x - 1), at(y, x + 1))) + 1);
      d[(size_t)y * W + x] = v;
    }
  std::vector<float> out(d.size());
  for (size_t i = 0; i < d.size(); ++i) out[i] = (float)d[i];
  return out;
}

// QuantizedPyramid::selectScatteredFeatures
void select_scattered(const std::vector<Cand>& cands, std::vector<Feat>& feats, size_t num_features, float distance) {
  feats.clear();
  float distance_sq = distance * distance;
  int i = 0;
  while (feats.size() < num_features) {
    const Cand& c = cands[i];
    bool keep = true;
    for (size_t j = 0; j < feats.size() && keep; ++j) {
      const int dx = c.x - feats[j].x, dy = c.y - feats[j].y;
      keep = (float)(dx * dx + dy * dy) >= distance_sq;
    }
    if (keep) feats.push_back(Feat{c.x, c.y, c.label});
    if (++i == (int)cands.size()) {
      i = 0;
      distance -= 1.0f;
      distance_sq = distance * distance;
    }
  }
}

bool by_score_desc(const Cand& a, const Cand& b) { return a.score > b.score; }  // Candidate::operator<

// ColorGradientPyramid::extractTemplate
bool extract_color(const std::vector<uint8_t>& angle, const std::vector<float>& mag, const std::vector<uint8_t>& mask, int H, int W,
                   size_t num_features, float strong_threshold, std::vector<Feat>& out) {
  std::vector<uint8_t> local;
  const bool no_mask = mask.empty();
  if (!no_mask) {
    std::vector<uint8_t> er = erode3(mask, H, W);
    local.resize(mask.size());
    for (size_t i = 0; i < mask.size(); ++i) local[i] = (uint8_t)(mask[i] > er[i] ? mask[i] - er[i] : 0);  // subtract saturates
  }
  const float threshold_sq = strong_threshold * strong_threshold;
  std::vector<Cand> cands;
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      const size_t i = (size_t)r * W + c;
      if (!no_mask && !local[i]) continue;
      const uint8_t q = angle[i];
      if (q > 0 && mag[i] > threshold_sq) cands.push_back(Cand{c, r, label_of(q), mag[i]});
    }
  if (cands.size() < num_features) return false;
  std::stable_sort(cands.begin(), cands.end(), by_score_desc);
  const float distance = static_cast<float>(cands.size() / num_features + 1);
  select_scattered(cands, out, num_features, distance);
  return true;
}

// DepthNormalPyramid::extractTemplate
bool extract_depth(const std::vector<uint8_t>& normal, const std::vector<uint8_t>& mask, int H, int W, size_t num_features,
                   int extract_threshold, std::vector<Feat>& out) {
  std::vector<uint8_t> local;
  const bool no_mask = mask.empty();
  if (!no_mask) local = erode3(erode3(mask, H, W), H, W);  // erode(..., iterations = 2, BORDER_REPLICATE)
  std::vector<float> dist[8];
  std::vector<uint8_t> temp((size_t)H * W);
  for (int k = 0; k < 8; ++k) {
    for (size_t i = 0; i < temp.size(); ++i) temp[i] = (uint8_t)(((no_mask || local[i]) ? (1u << k) : 0u) & normal[i]);
    dist[k] = dist_chessboard(temp, H, W);
  }
  int label_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<Cand> cands;
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      const size_t i = (size_t)r * W + c;
      if (!no_mask && !local[i]) continue;
      const uint8_t q = normal[i];
      if (q == 0 || q == 255) continue;
      const int label = label_of(q);
      if (label < 0) continue;
      const float score = dist[label][i];
      if (score >= (float)extract_threshold) {
        cands.push_back(Cand{c, r, label, score});
        ++label_counts[label];
      }
    }
  if (cands.size() < num_features) return false;
  for (Cand& c : cands) c.score /= (float)label_counts[c.label];
  std::stable_sort(cands.begin(), cands.end(), by_score_desc);
  size_t area = normal.size();
  if (!no_mask) { area = 0; for (uint8_t v : local) area += v != 0; }
  const float distance = sqrtf((float)area) / sqrtf((float)num_features) + 1.5f;
  select_scattered(cands, out, num_features, distance);
  return true;
}

}  // namespace

lmx_status train_add_template(lmx_bank* bank, int device, const lmx_image* sources, int n_sources, const char* class_id,
                              const lmx_image* object_mask, int32_t* template_id, int32_t* bounding_box) {
  const int L = (int)bank->T.size(), M = (int)bank->mods.size();
  if (n_sources != M) { set_error("sources.size()=%d != modalities.size()=%d", n_sources, M); return LMX_ERR_SHAPE; }
  const int H0 = sources[0].rows, W0 = sources[0].cols;
  if (H0 < 16 || W0 < 16) { set_error("source image too small"); return LMX_ERR_SHAPE; }
  for (int m = 0; m < M; ++m) {
    const bool cg = bank->mods[m].type == LMX_MOD_COLOR_GRADIENT;
    const lmx_image& im = sources[m];
    if (!im.data || im.rows != H0 || im.cols != W0 || im.channels != (cg ? 3 : 1) || im.elem_size != (cg ? 1 : 2) ||
        im.row_stride_bytes < (size_t)W0 * (cg ? 3 : 2)) {
      set_error("source %d: expected %s of size %dx%d", m, cg ? "8UC3" : "16UC1", W0, H0);
      return LMX_ERR_SHAPE;
    }
  }
  std::vector<std::vector<uint8_t>> masks(L);  // per level (INTER_NEAREST /2 per pyrDown); empty = no mask
  if (object_mask && object_mask->data) {
    if (object_mask->rows != H0 || object_mask->cols != W0 || object_mask->channels != 1 || object_mask->elem_size != 1) {
      set_error("object_mask must be 8UC1 of the source size");
      return LMX_ERR_SHAPE;
    }
    masks[0].resize((size_t)H0 * W0);
    for (int y = 0; y < H0; ++y)
      std::memcpy(&masks[0][(size_t)y * W0], (const uint8_t*)object_mask->data + (size_t)y * object_mask->row_stride_bytes, (size_t)W0);
    for (int l = 1, h = H0, w = W0; l < L; ++l) {
      const int hn = h / 2, wn = w / 2;
      masks[l].resize((size_t)hn * wn);
      for (int y = 0; y < hn; ++y)
        for (int x = 0; x < wn; ++x) masks[l][(size_t)y * wn + x] = masks[l - 1][(size_t)(2 * y) * w + 2 * x];
      h = hn; w = wn;
    }
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available; this library has no CPU path"); return LMX_ERR_NO_DEVICE; }
  TR_HIP(hipSetDevice(device));

  // ---- device: labels (+ magnitudes) per modality and level -------------------------------------------------------------
  std::vector<std::vector<std::vector<uint8_t>>> labels(M, std::vector<std::vector<uint8_t>>(L));
  std::vector<std::vector<std::vector<float>>> mags(M, std::vector<std::vector<float>>(L));
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* p : to_free) (void)hipFree(p); to_free.clear(); };
  auto dmalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 256) + 64) != hipSuccess) return nullptr; to_free.push_back(p); return p; };
  lmx_status st = LMX_OK;
  auto run = [&]() -> lmx_status {
    for (int m = 0; m < M; ++m) {
      const lmx_modality_desc& md = bank->mods[m];
      const bool cg = md.type == LMX_MOD_COLOR_GRADIENT;
      const size_t px0 = (size_t)H0 * W0, row_bytes = (size_t)W0 * (cg ? 3 : 2);
      std::vector<uint8_t> packed(px0 * (cg ? 3 : 2));
      for (int y = 0; y < H0; ++y) std::memcpy(&packed[(size_t)y * row_bytes], (const uint8_t*)sources[m].data + (size_t)y * sources[m].row_stride_bytes, row_bytes);
      void* d_src = dmalloc(packed.size());
      if (!d_src) { set_error("hipMalloc failed"); return LMX_ERR_HIP; }
      TR_HIP(hipMemcpy(d_src, packed.data(), packed.size(), hipMemcpyHostToDevice));
      uint8_t* d_prev_q = nullptr;
      uint8_t* d_cur_src = (uint8_t*)d_src;
      int h = H0, w = W0;
      for (int l = 0; l < L; ++l) {
        if (l > 0) { h /= 2; w /= 2; }
        if (h < 1 || w < 1) { set_error("image too small for %d pyramid levels", L); return LMX_ERR_SHAPE; }
        uint8_t* d_q = (uint8_t*)dmalloc((size_t)h * w);
        if (!d_q) { set_error("hipMalloc failed"); return LMX_ERR_HIP; }
        labels[m][l].resize((size_t)h * w);
        if (cg) {
          float* d_mag = (float*)dmalloc((size_t)h * w * 4);
          uint8_t* d_next = l + 1 < L ? (uint8_t*)dmalloc((size_t)(h / 2) * (w / 2) * 3) : nullptr;
          if (!d_mag || (l + 1 < L && !d_next)) { set_error("hipMalloc failed"); return LMX_ERR_HIP; }
          launch_color_quantize(nullptr, d_cur_src, d_q, d_next, h, w, 1, md.weak_threshold, d_mag);
          TR_HIP(hipDeviceSynchronize());
          mags[m][l].resize((size_t)h * w);
          TR_HIP(hipMemcpy(mags[m][l].data(), d_mag, (size_t)h * w * 4, hipMemcpyDeviceToHost));
          d_cur_src = d_next;
        } else {
          if (l == 0) {
            std::vector<uint8_t> bins(lmx::kNormalBinsDeviceBytes);   // zero-initialised: the trailing entry stays 0
            if (!normal_lut_to_bins(bank->normal_lut.data(), bins.data())) { set_error("bank holds an invalid normal LUT"); return LMX_ERR_INVALID_ARG; }
            uint8_t* d_bins = (uint8_t*)dmalloc(bins.size());
            if (!d_bins) { set_error("hipMalloc failed"); return LMX_ERR_HIP; }
            TR_HIP(hipMemcpy(d_bins, bins.data(), bins.size(), hipMemcpyHostToDevice));
            launch_depth_quantize(nullptr, (const uint16_t*)d_src, d_q, nullptr, h, w, 1, md.distance_threshold, md.difference_threshold, d_bins);
          }
          else launch_nn_down2(nullptr, d_prev_q, d_q, h, w, 1);
          TR_HIP(hipDeviceSynchronize());
        }
        TR_HIP(hipMemcpy(labels[m][l].data(), d_q, (size_t)h * w, hipMemcpyDeviceToHost));
        d_prev_q = d_q;
      }
    }
    return LMX_OK;
  };
  st = run();
  cleanup();
  if (st != LMX_OK) return st;

  // ---- host: extractTemplate per (level, modality), cropTemplates ------------------------------------------------------------
  // With an object mask (what the reference's trainers pass: the rendered silhouette) everything extractTemplate looks at lies inside
  // the mask's bounding box: candidates exist only where the eroded mask is set, and the erosions and the chessboard distance
  // transforms at those pixels depend only on pixels within 2 of the box (outside the mask every intermediate image is zero).  The
  // stages therefore run on the box grown by 2 pixels (clamped to the image, where BORDER_REPLICATE keeps its meaning) instead of the
  // whole frame -- the same values, the same candidate order (raster order is preserved), 10-20x fewer pixels for a rendered object:
  // eight full-frame distance transforms were most of the 16 ms a 640x480 RGB-D view took (round 3).
  std::vector<std::vector<Feat>> tp((size_t)L * M);
  *template_id = -1;
  struct Win { int x0, y0, w, h; };
  std::vector<Win> wins(L);
  for (int l = 0, h = H0, w = W0; l < L; ++l) {
    if (l > 0) { h /= 2; w /= 2; }
    Win win{0, 0, w, h};
    if (!masks[l].empty()) {
      int bx0 = w, by0 = h, bx1 = -1, by1 = -1;
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
          if (masks[l][(size_t)y * w + x]) { bx0 = std::min(bx0, x); bx1 = std::max(bx1, x); by0 = std::min(by0, y); by1 = std::max(by1, y); }
      if (bx1 >= 0) {
        win.x0 = std::max(0, bx0 - 2); win.y0 = std::max(0, by0 - 2);
        win.w = std::min(w, bx1 + 3) - win.x0; win.h = std::min(h, by1 + 3) - win.y0;
      } else {
        win = Win{0, 0, std::min(w, 4), std::min(h, 4)};   // an all-zero mask: no candidates anywhere, any window shows that
      }
    }
    wins[l] = win;
  }
  auto crop_u8 = [](const std::vector<uint8_t>& v, int w, const Win& win) {
    std::vector<uint8_t> out((size_t)win.w * win.h);
    for (int y = 0; y < win.h; ++y) std::memcpy(&out[(size_t)y * win.w], &v[(size_t)(win.y0 + y) * w + win.x0], (size_t)win.w);
    return out;
  };
  for (int m = 0; m < M; ++m) {
    const lmx_modality_desc& md = bank->mods[m];
    size_t num_features = (size_t)md.num_features;
    int extract_threshold = md.extract_threshold;
    int h = H0, w = W0;
    for (int l = 0; l < L; ++l) {
      if (l > 0) { h /= 2; w /= 2; num_features /= 2; extract_threshold /= 2; }
      if (num_features > 63) { set_error("num_features %zu > 63", num_features); return LMX_ERR_SHAPE; }
      const Win& win = wins[l];
      const bool whole = win.w == w && win.h == h;
      std::vector<Feat>& out = tp[(size_t)l * M + m];
      bool ok;
      if (whole) {
        if (md.type == LMX_MOD_COLOR_GRADIENT) ok = extract_color(labels[m][l], mags[m][l], masks[l], h, w, num_features, md.strong_threshold, out);
        else ok = extract_depth(labels[m][l], masks[l], h, w, num_features, extract_threshold, out);
      } else {
        const std::vector<uint8_t> lab = crop_u8(labels[m][l], w, win), msk = crop_u8(masks[l], w, win);
        if (md.type == LMX_MOD_COLOR_GRADIENT) {
          std::vector<float> mg((size_t)win.w * win.h);
          for (int y = 0; y < win.h; ++y) std::memcpy(&mg[(size_t)y * win.w], &mags[m][l][(size_t)(win.y0 + y) * w + win.x0], (size_t)win.w * sizeof(float));
          ok = extract_color(lab, mg, msk, win.h, win.w, num_features, md.strong_threshold, out);
        } else {
          ok = extract_depth(lab, msk, win.h, win.w, num_features, extract_threshold, out);
        }
        for (Feat& f : out) { f.x += win.x0; f.y += win.y0; }
      }
      if (!ok) return LMX_OK;  // upstream: addTemplate returns -1, nothing is added
    }
  }
  int min_x = INT32_MAX, min_y = INT32_MAX, max_x = INT32_MIN, max_y = INT32_MIN;
  for (int k = 0; k < L * M; ++k) {
    const int level = k / M;
    for (const Feat& f : tp[k]) {
      const int x = f.x << level, y = f.y << level;
      min_x = std::min(min_x, x); min_y = std::min(min_y, y); max_x = std::max(max_x, x); max_y = std::max(max_y, y);
    }
  }
  if (min_x % 2 == 1) --min_x;
  if (min_y % 2 == 1) --min_y;
  std::vector<int32_t> templates, features;
  int32_t fb = 0;
  for (int k = 0; k < L * M; ++k) {
    const int level = k / M;
    const int ox = min_x >> level, oy = min_y >> level;
    templates.insert(templates.end(), {(max_x - min_x) >> level, (max_y - min_y) >> level, level, fb, (int32_t)tp[k].size()});
    for (const Feat& f : tp[k]) features.insert(features.end(), {f.x - ox, f.y - oy, f.label});
    fb += (int32_t)tp[k].size();
  }
  const int32_t existing = lmx_bank_num_templates(bank, class_id);
  st = lmx_bank_add_class(bank, class_id, 1, templates.data(), features.data(), fb);
  if (st != LMX_OK) return st;
  *template_id = existing;
  if (bounding_box) { bounding_box[0] = min_x; bounding_box[1] = min_y; bounding_box[2] = max_x - min_x; bounding_box[3] = max_y - min_y; }
  return LMX_OK;
}

}  // namespace lmx

extern "C" lmx_status lmx_bank_add_template(lmx_bank* bank, int32_t device, const lmx_image* sources, int32_t n_sources, const char* class_id,
                                            const lmx_image* object_mask, int32_t* template_id, int32_t bounding_box[4]) {
  return lmx::guarded("lmx_bank_add_template", [&]() -> lmx_status {
  if (!bank || !sources || !class_id || !template_id) { lmx::set_error("lmx_bank_add_template: null argument"); return LMX_ERR_INVALID_ARG; }
  return lmx::train_add_template(bank, device, sources, n_sources, class_id, object_mask, template_id, bounding_box);
  });
}
