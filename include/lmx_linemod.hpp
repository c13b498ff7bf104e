// include/lmx_linemod.hpp -- header-only C++ facade over the C ABI (include/lmx.h) that keeps the names and
// semantics of cv::linemod as the reference uses them, so a maintainer can swap the body of
//   rgbdDetector::linemod_detection   (/root/reference/src/rgbdDetector.cpp:31-34)
// and of readLinemod                  (/root/reference/src/rgbdDetector.cpp:1668-1680)
// without touching src/linemod_ensenso_detect_*.cpp.  See INTEGRATION.md for the exact edit.
//
//   lmx::linemod::Feature / Template / Match      <-> cv::linemod::Feature / Template / Match
//   lmx::linemod::Detector::match(sources, threshold, matches, class_ids)
//                                                <-> cv::linemod::Detector::match(sources, threshold, matches, class_ids, noArray())
//   Detector::read / classIds / getTemplates / numTemplates / getT / pyramidLevels: same meaning as upstream.
// Errors: any non-OK lmx_status throws lmx::linemod::Exception (the analogue of the cv::Exception a CV_Assert throws).
// With LMX_HAVE_OPENCV defined, overloads taking cv::Mat are provided (the only place OpenCV types appear).
#pragma once

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmx.h"

#ifdef LMX_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace lmx {
namespace linemod {

struct Exception : std::runtime_error {
  lmx_status status;
  Exception(lmx_status s, const char* what) : std::runtime_error(what), status(s) {}
};

inline void check(lmx_status s) {
  if (s != LMX_OK) throw Exception(s, lmx_last_error());
}

struct Feature {
  int x, y, label;
};

struct Template {
  int width, height, pyramid_level;
  std::vector<Feature> features;
};

struct Match {
  Match() : x(0), y(0), similarity(0), template_id(0) {}
  Match(int x_, int y_, float s, const std::string& c, int t) : x(x_), y(y_), similarity(s), class_id(c), template_id(t) {}
  bool operator<(const Match& rhs) const {
    if (similarity != rhs.similarity) return similarity > rhs.similarity;
    return template_id < rhs.template_id;
  }
  bool operator==(const Match& rhs) const {
    return x == rhs.x && y == rhs.y && similarity == rhs.similarity && class_id == rhs.class_id;
  }
  int x, y;
  float similarity;
  std::string class_id;
  int template_id;
};

// Non-owning image view (what cv::Mat carries for this path).
struct Image {
  const void* data;
  int rows, cols, channels, elem_size;
  size_t step;
  lmx_image c() const { return lmx_image{data, rows, cols, channels, elem_size, step}; }
};

class Detector {
 public:
  Detector() {}
  ~Detector() { reset(); }
  Detector(const Detector&) = delete;
  Detector& operator=(const Detector&) = delete;

  // readLinemod(filename): FileStorage YAML -> Detector::read + readClass per class
  void read(const std::string& filename) {
    reset();
    // through the library's file cache (parsed once per (path, mtime, size), binary side file for the next process); this detector gets a
    // private copy it may modify: ~1 ms for 3000 templates instead of the 0.13-0.24 s parse on every read of the same file
    const lmx_bank* cached = nullptr;
    check(lmx_bank_load_yaml_cached(filename.c_str(), &cached));
    const lmx_status st = lmx_bank_clone(cached, &bank_);
    lmx_bank_release(cached);
    check(st);
  }
  void write(const std::string& filename) const { check(lmx_bank_save_yaml(bank_, filename.c_str())); }

  // Device placement (no upstream analogue).  Called lazily by match() with the frame size of the first call.
  void setDevice(int device, int max_batch = 1, int max_candidates = 0, void* stream = nullptr) {
    device_ = device; max_batch_ = max_batch; max_candidates_ = max_candidates; stream_ = stream;
  }

  std::vector<std::string> classIds() const {
    std::vector<std::string> ids;
    for (int i = 0; i < lmx_bank_num_classes(bank_); ++i) ids.push_back(lmx_bank_class_id(bank_, i));
    return ids;
  }
  int numTemplates() const { return lmx_bank_num_templates(bank_, nullptr); }
  int numTemplates(const std::string& class_id) const { return lmx_bank_num_templates(bank_, class_id.c_str()); }
  int numClasses() const { return lmx_bank_num_classes(bank_); }
  int getT(int pyramid_level) const { return lmx_bank_T(bank_, pyramid_level); }
  int pyramidLevels() const { return lmx_bank_pyramid_levels(bank_); }

  std::vector<Template> getTemplates(const std::string& class_id, int template_id) const {
    const int per = lmx_bank_pyramid_levels(bank_) * lmx_bank_num_modalities(bank_);
    std::vector<Template> out(per);
    for (int k = 0; k < per; ++k) {
      const int32_t* f = nullptr;
      int32_t n = 0, w = 0, h = 0, l = 0;
      check(lmx_bank_get_template(bank_, class_id.c_str(), template_id, k, &w, &h, &l, &f, &n));
      out[k].width = w; out[k].height = h; out[k].pyramid_level = l;
      for (int i = 0; i < n; ++i) out[k].features.push_back(Feature{f[3 * i], f[3 * i + 1], f[3 * i + 2]});
    }
    return out;
  }

  // Detector::match.  `matches` is cleared first, like upstream.
  void match(const std::vector<Image>& sources, float threshold, std::vector<Match>& matches,
             const std::vector<std::string>& class_ids = std::vector<std::string>()) {
    matches.clear();
    if (sources.empty()) throw Exception(LMX_ERR_SHAPE, "match: no sources");
    std::vector<lmx_image> imgs;
    for (const Image& s : sources) imgs.push_back(s.c());
    std::vector<const char*> cids;
    for (const std::string& c : class_ids) cids.push_back(c.c_str());
    if (buf_.size() < 4096) buf_.resize(4096);
    for (int attempt = 0;; ++attempt) {
      ensure_ctx(sources[0].cols, sources[0].rows);
      size_t n = 0;
      lmx_status st = lmx_match(ctx_, imgs.data(), (int)imgs.size(), threshold, cids.empty() ? nullptr : cids.data(), (int)cids.size(),
                                buf_.data(), buf_.size(), &n);
      if (st == LMX_ERR_OVERFLOW && n > buf_.size()) { buf_.resize(n); continue; }  // output buffer too small: retry
      if (st == LMX_ERR_OVERFLOW && attempt < 8) {
        // the device's candidate / match lists overflowed (upstream has no such limit): rebuild the context with lists sized from
        // the counts the failed call reports, and repeat the call
        int64_t n_cand = 0, n_raw = 0;
        check(lmx_ctx_stats(ctx_, &n_cand, &n_raw));
        const int64_t need = (n_cand > n_raw ? n_cand : n_raw) / (max_batch_ > 0 ? max_batch_ : 1) + 1024;
        int grown = max_candidates_ > 16384 ? max_candidates_ : 16384;
        while (grown < need && grown < (1 << 26)) grown *= 2;
        if (grown > max_candidates_ && grown > 16384) {
          max_candidates_ = grown;
          lmx_ctx_destroy(ctx_); ctx_ = nullptr;
          continue;
        }
      }
      check(st);
      for (size_t i = 0; i < n; ++i)
        matches.push_back(Match(buf_[i].x, buf_[i].y, buf_[i].similarity, lmx_bank_class_id(bank_, buf_[i].class_index), buf_[i].template_id));
      return;
    }
  }

#ifdef LMX_HAVE_OPENCV
  static Image view(const cv::Mat& m) { return Image{m.data, m.rows, m.cols, m.channels(), (int)m.elemSize1(), m.step[0]}; }
  void match(const std::vector<cv::Mat>& sources, float threshold, std::vector<Match>& matches,
             const std::vector<std::string>& class_ids = std::vector<std::string>()) {
    std::vector<Image> v;
    for (const cv::Mat& m : sources) v.push_back(view(m));
    match(v, threshold, matches, class_ids);
  }
#endif

  lmx_bank* bank() const { return bank_; }
  lmx_ctx* context() const { return ctx_; }

 private:
  void reset() {
    if (ctx_) lmx_ctx_destroy(ctx_);
    if (bank_) lmx_bank_destroy(bank_);
    ctx_ = nullptr; bank_ = nullptr;
  }
  void ensure_ctx(int w, int h) {
    if (ctx_ && w == w_ && h == h_) return;
    if (ctx_) { lmx_ctx_destroy(ctx_); ctx_ = nullptr; }
    lmx_ctx_desc d;
    std::memset(&d, 0, sizeof(d));
    d.device = device_; d.width = w; d.height = h; d.max_batch = max_batch_; d.max_candidates = max_candidates_; d.stream = stream_;
    check(lmx_ctx_create(bank_, &d, &ctx_));
    w_ = w; h_ = h;
  }
  lmx_bank* bank_ = nullptr;
  lmx_ctx* ctx_ = nullptr;
  int device_ = 0, max_batch_ = 1, max_candidates_ = 0, w_ = 0, h_ = 0;
  void* stream_ = nullptr;
  std::vector<lmx_match_t> buf_;
};

// The reference's readLinemod (src/rgbdDetector.cpp:1668-1680) with the same shape.
inline std::shared_ptr<Detector> readLinemod(const std::string& filename) {
  std::shared_ptr<Detector> d(new Detector);
  d->read(filename);
  return d;
}

}  // namespace linemod
}  // namespace lmx
