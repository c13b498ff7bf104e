// include/lmx_cv_linemod.hpp -- cv::linemod, re-implemented over liblmx (the MI355X engine behind include/lmx.h).
//
// Purpose: the reference's sources keep compiling UNCHANGED.  They use cv::linemod through these names
//   cv::Ptr<cv::linemod::Detector> detector(new cv::linemod::Detector);  detector->read(fs.root());  detector->readClass(*i);
//                                       src/linemod_ensenso_detect_3_mult_detect_service.cpp:708-721, src/rgbdDetector.cpp:1668-1680
//   linemod_detector->match(sources, threshold, matches, std::vector<String>(), noArray());          src/rgbdDetector.cpp:31-34
//   std::vector<cv::linemod::Template> t = detector->getTemplates(it->class_id, it->template_id);     ..._service.cpp:351
//   cv::linemod::Feature f = templates[m].features[i];                                               ..._service.cpp:743
//   detector->classIds().empty()                                                                     ..._detect.cpp:290
//   cv::Ptr<cv::linemod::Detector> d(new cv::linemod::Detector(modalities, T));  d->addTemplate(sources, "obj", mask);
//   detector->write(fs);  detector->writeClass(ids[i], fs);                                          src/renderer.cpp:56-70,179-185,308
// and this header provides every one of them with upstream's signatures and semantics.
//
// How it drops in.  Include it AFTER the OpenCV headers (it needs cv::Mat, cv::Ptr, cv::String, cv::Rect, cv::FileNode,
// cv::FileNodeIterator, cv::FileStorage, cv::OutputArrayOfArrays / cv::noArray()), e.g. as the last include of
// include/linemod_pose_estimation/rgbdDetector.h (after its lines 29-31).  OpenCV's own cv::linemod may be declared already
// (2.4: objdetect, 3.x/4.x: rgbd contrib), so the classes live in namespace cv::lmx_linemod and the header ends with
//     #define linemod lmx_linemod
// from where on every `cv::linemod::X` / `linemod::X` token in the including source names this implementation.  No other edit
// is needed in src/linemod_ensenso_detect_*.cpp, src/linemod_carmine_detect.cpp, src/rgbdDetector.cpp or the trainers; link
// liblmx.so instead of (or next to) OpenCV's linemod.  Define LMX_KEEP_CV_LINEMOD before including to skip the macro and use
// cv::lmx_linemod:: explicitly.  (tests/cpp/ compiles a caller written in the reference's style against a minimal stand-in for
// the OpenCV types, because this image has no OpenCV.)
//
// Differences from upstream, all deliberate:
//   * match() runs on the GPU through a device context that is created on first use for the frame size of that call and shared
//     process-wide between detectors holding the same templates (lmx_ctx_acquire): the service node's per-request rebuild of the
//     detector (..._service.cpp:1784-1786) re-uses the resident bank.
//   * `masks` (the reference passes none, src/rgbdDetector.cpp:33) is honoured: one 8UC1 Mat per modality, empty Mats allowed
//     (lmx_match_masked).
//   * errors that upstream raises with CV_Assert throw cv::lmx_linemod::Error (derived from std::runtime_error; with real OpenCV
//     define LMX_CV_THROW(status, msg) as CV_Error(cv::Error::StsAssert, msg) before including to get cv::Exception instead).
//   * DepthNormal's NORMAL_LUT is data on the bank (include/lmx.h): Detector::setNormalLut / loadNormalLut install OpenCV's
//     normal_lut.i; without it the documented default table is used.
#ifndef LMX_CV_LINEMOD_HPP_
#define LMX_CV_LINEMOD_HPP_

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmx.h"

namespace cv {
namespace lmx_linemod {

struct Error : std::runtime_error {
  lmx_status status;
  Error(lmx_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};
#ifndef LMX_CV_THROW
#define LMX_CV_THROW(status, msg) throw ::cv::lmx_linemod::Error((status), (msg))
#endif
inline void lmx_check(lmx_status s) {
  if (s != LMX_OK) LMX_CV_THROW(s, lmx_last_error());
}

struct Feature {
  int x, y, label;
  Feature() : x(0), y(0), label(0) {}
  Feature(int x_, int y_, int l) : x(x_), y(y_), label(l) {}
  void read(const FileNode& fn) {
    FileNodeIterator it = fn.begin();
    x = (int)(*it); ++it;
    y = (int)(*it); ++it;
    label = (int)(*it);
  }
  void write(FileStorage& fs) const { fs << "[:" << x << y << label << "]"; }
};

struct Template {
  int width, height, pyramid_level;
  std::vector<Feature> features;
  Template() : width(0), height(0), pyramid_level(0) {}
  void read(const FileNode& fn) {
    width = (int)fn["width"];
    height = (int)fn["height"];
    pyramid_level = (int)fn["pyramid_level"];
    FileNode ff = fn["features"];
    features.clear();
    for (FileNodeIterator it = ff.begin(), e = ff.end(); it != e; ++it) {
      Feature f;
      f.read(*it);
      features.push_back(f);
    }
  }
  void write(FileStorage& fs) const {
    fs << "width" << width;
    fs << "height" << height;
    fs << "pyramid_level" << pyramid_level;
    fs << "features" << "[";
    for (size_t i = 0; i < features.size(); ++i) features[i].write(fs);
    fs << "]";
  }
};

// Modalities carry parameters only: the per-pixel work happens inside liblmx.
class Modality {
 public:
  virtual ~Modality() {}
  virtual String name() const = 0;
  virtual void read(const FileNode& fn) = 0;
  virtual void write(FileStorage& fs) const = 0;
  virtual lmx_modality_desc desc() const = 0;
  static Ptr<Modality> create(const String& modality_type);
  static Ptr<Modality> create(const FileNode& fn);
};

class ColorGradient : public Modality {
 public:
  ColorGradient() : weak_threshold(10.0f), num_features(63), strong_threshold(55.0f) {}
  ColorGradient(float weak, size_t nf, float strong) : weak_threshold(weak), num_features(nf), strong_threshold(strong) {}
  String name() const { return "ColorGradient"; }
  void read(const FileNode& fn) {
    weak_threshold = (float)fn["weak_threshold"];
    num_features = (size_t)(int)fn["num_features"];
    strong_threshold = (float)fn["strong_threshold"];
  }
  void write(FileStorage& fs) const {
    fs << "type" << "ColorGradient";
    fs << "weak_threshold" << weak_threshold;
    fs << "num_features" << (int)num_features;
    fs << "strong_threshold" << strong_threshold;
  }
  lmx_modality_desc desc() const {
    lmx_modality_desc d;
    std::memset(&d, 0, sizeof(d));
    d.type = LMX_MOD_COLOR_GRADIENT; d.weak_threshold = weak_threshold; d.strong_threshold = strong_threshold; d.num_features = (int32_t)num_features;
    return d;
  }
  float weak_threshold;
  size_t num_features;
  float strong_threshold;
};

class DepthNormal : public Modality {
 public:
  DepthNormal() : distance_threshold(2000), difference_threshold(50), num_features(63), extract_threshold(2) {}
  DepthNormal(int dist, int diff, size_t nf, int extract) : distance_threshold(dist), difference_threshold(diff), num_features(nf), extract_threshold(extract) {}
  String name() const { return "DepthNormal"; }
  void read(const FileNode& fn) {
    distance_threshold = (int)fn["distance_threshold"];
    difference_threshold = (int)fn["difference_threshold"];
    num_features = (size_t)(int)fn["num_features"];
    extract_threshold = (int)fn["extract_threshold"];
  }
  void write(FileStorage& fs) const {
    fs << "type" << "DepthNormal";
    fs << "distance_threshold" << distance_threshold;
    fs << "difference_threshold" << difference_threshold;
    fs << "num_features" << (int)num_features;
    fs << "extract_threshold" << extract_threshold;
  }
  lmx_modality_desc desc() const {
    lmx_modality_desc d;
    std::memset(&d, 0, sizeof(d));
    d.type = LMX_MOD_DEPTH_NORMAL; d.num_features = (int32_t)num_features; d.distance_threshold = distance_threshold;
    d.difference_threshold = difference_threshold; d.extract_threshold = extract_threshold;
    return d;
  }
  int distance_threshold;
  int difference_threshold;
  size_t num_features;
  int extract_threshold;
};

inline Ptr<Modality> Modality::create(const String& modality_type) {
  if (modality_type == "ColorGradient") return Ptr<Modality>(new ColorGradient());
  if (modality_type == "DepthNormal") return Ptr<Modality>(new DepthNormal());
  LMX_CV_THROW(LMX_ERR_PARSE, std::string("unknown modality type '") + std::string(modality_type) + "'");
}
inline Ptr<Modality> Modality::create(const FileNode& fn) {
  String type = (String)fn["type"];
  Ptr<Modality> m = create(type);
  m->read(fn);
  return m;
}

struct Match {
  Match() : x(0), y(0), similarity(0), template_id(0) {}
  Match(int x_, int y_, float s, const String& c, int t) : x(x_), y(y_), similarity(s), class_id(c), template_id(t) {}
  /// Sort matches with high similarity to the front
  bool operator<(const Match& rhs) const {
    if (similarity != rhs.similarity) return similarity > rhs.similarity;
    return template_id < rhs.template_id;
  }
  bool operator==(const Match& rhs) const { return x == rhs.x && y == rhs.y && similarity == rhs.similarity && class_id == rhs.class_id; }
  int x, y;
  float similarity;
  String class_id;
  int template_id;
};

class Detector {
  struct CtxLock {   // lmx_ctx_lock / lmx_ctx_unlock as a scope
    explicit CtxLock(lmx_ctx* c) : c_(c) { lmx_ctx_lock(c_); }
    ~CtxLock() { lmx_ctx_unlock(c_); }
    lmx_ctx* c_;
   private:
    CtxLock(const CtxLock&);
    CtxLock& operator=(const CtxLock&);
  };

 public:
  typedef std::vector<Template> TemplatePyramid;

  Detector() : foreign_depth_(false), bank_(NULL), shared_(NULL), ctx_(NULL), ctx_w_(0), ctx_h_(0), device_(0), max_candidates_(0) {}
  Detector(const std::vector<Ptr<Modality> >& modalities, const std::vector<int>& T_pyramid)
      : modalities_(modalities), T_at_level_(T_pyramid), foreign_depth_(false), bank_(NULL), shared_(NULL), ctx_(NULL), ctx_w_(0), ctx_h_(0), device_(0), max_candidates_(0) {}
  ~Detector() { drop(); }

  // liblmx extension (no upstream analogue): the whole of the reference's readLinemod(filename) in one call.  The templates come from the
  // library's process-wide cache -- parsed once per (path, mtime, size), a binary side file next to the yml for the next process -- and are
  // SHARED read-only between all detectors loaded from that file; with the device context cached too (lmx_ctx_acquire), a node that builds
  // its detector on every request (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1784-1786) pays ~0.1 ms per request instead of a
  // YAML parse (0.14-0.23 s here for 3000 templates, seconds in OpenCV's FileStorage).  A detector loaded this way that is modified later
  // (addTemplate, readClass, setNormalLut) quietly switches to a private copy.  The one-line edit in the reference:
  //     cv::Ptr<cv::linemod::Detector> rgbdDetector::readLinemod(const std::string& filename) { return cv::linemod::Detector::load(filename); }
  static Ptr<Detector> load(const String& filename) {
    const lmx_bank* b = NULL;
    lmx_check(lmx_bank_load_yaml_cached(std::string(filename).c_str(), &b));
    Ptr<Detector> d(new Detector);
    d->shared_ = b;
    const int L = lmx_bank_pyramid_levels(b);
    for (int l = 0; l < L; ++l) d->T_at_level_.push_back(lmx_bank_T(b, l));
    lmx_modality_desc md;
    for (int i = 0; i < lmx_bank_num_modalities(b); ++i) {
      lmx_check(lmx_bank_modality(b, i, &md));
      if (md.type == LMX_MOD_COLOR_GRADIENT) d->modalities_.push_back(Ptr<Modality>(new ColorGradient(md.weak_threshold, (size_t)md.num_features, md.strong_threshold)));
      else d->modalities_.push_back(Ptr<Modality>(new DepthNormal(md.distance_threshold, md.difference_threshold, (size_t)md.num_features, md.extract_threshold)));
    }
    d->foreign_depth_ = lmx_bank_normal_lut_origin(b) == LMX_LUT_UNKNOWN;
    return d;
  }

  // ---- matching: Detector::match(sources, threshold, matches, class_ids, quantized_images, masks) const -----------------------
  void match(const std::vector<Mat>& sources, float threshold, std::vector<Match>& matches, const std::vector<String>& class_ids = std::vector<String>(),
             OutputArrayOfArrays quantized_images = noArray(), const std::vector<Mat>& masks = std::vector<Mat>()) const {
    matches.clear();
    if (sources.size() != modalities_.size()) LMX_CV_THROW(LMX_ERR_SHAPE, "sources.size() != modalities.size()");  // upstream CV_Assert
    if (!masks.empty() && masks.size() != modalities_.size()) LMX_CV_THROW(LMX_ERR_SHAPE, "masks.size() != modalities.size()");  // upstream CV_Assert
    std::vector<lmx_image> mimgs(masks.size());
    for (size_t i = 0; i < masks.size(); ++i) {
      const Mat& m = masks[i];
      lmx_image im;
      im.data = m.empty() ? NULL : m.data; im.rows = m.rows; im.cols = m.cols; im.channels = m.empty() ? 1 : m.channels(); im.elem_size = m.empty() ? 1 : (int32_t)m.elemSize1();
      im.row_stride_bytes = m.empty() ? 0 : m.step[0];
      mimgs[i] = im;
    }
    std::vector<lmx_image> imgs(sources.size());
    for (size_t i = 0; i < sources.size(); ++i) {
      const Mat& m = sources[i];
      lmx_image im;
      im.data = m.data; im.rows = m.rows; im.cols = m.cols; im.channels = m.channels(); im.elem_size = (int32_t)m.elemSize1(); im.row_stride_bytes = m.step[0];
      imgs[i] = im;
    }
    std::vector<const char*> cids;
    for (size_t i = 0; i < class_ids.size(); ++i) cids.push_back(class_ids[i].c_str());
    // Upstream never runs out of room, so neither may the drop-in: when the device's candidate / match lists overflow (default
    // 16384 per frame; a low threshold on a cluttered scene yields more) the context is re-acquired with lists sized from the
    // counts the failed call reports, and the call repeated.  The larger size sticks to this detector (and is part of the cache key).
    for (int attempt = 0;; ++attempt) {
      lmx_ctx* ctx = context(sources[0].cols, sources[0].rows);
      bool regrow = false;
      {
        CtxLock lock(ctx);   // the context may be shared with another Detector of the same bank (lmx_ctx_acquire): match + read-backs as one unit
        if (buf_.size() < 4096) buf_.resize(4096);   // (under the lock: concurrent match() calls on ONE detector share this buffer)
        size_t n = 0;
        lmx_status st = lmx_match_masked(ctx, imgs.data(), mimgs.empty() ? NULL : &mimgs[0], (int32_t)imgs.size(), threshold, cids.empty() ? NULL : &cids[0],
                                         (int32_t)cids.size(), &buf_[0], buf_.size(), &n);
        if (st == LMX_ERR_OVERFLOW && n > buf_.size()) { buf_.resize(n); continue; }  // output buffer too small: retry
        if (st == LMX_ERR_OVERFLOW && attempt < 8) {
          int64_t n_cand = 0, n_raw = 0;
          lmx_check(lmx_ctx_stats(ctx, &n_cand, &n_raw));
          const int64_t need = (n_cand > n_raw ? n_cand : n_raw) + 1024;
          int grown = max_candidates_ > 16384 ? max_candidates_ : 16384;
          while (grown < need && grown < (1 << 26)) grown *= 2;
          if (grown > max_candidates_ && grown > 16384) { max_candidates_ = grown; regrow = true; }
        }
        if (!regrow) {
          lmx_check(st);
          matches.reserve(n);
          for (size_t i = 0; i < n; ++i)
            matches.push_back(Match(buf_[i].x, buf_[i].y, buf_[i].similarity, String(lmx_bank_class_id(bank(), buf_[i].class_index)), buf_[i].template_id));
          if (quantized_images.needed()) {
            // upstream returns the quantized image of every (level, modality), index l*M + m
            const int L = pyramidLevels(), M = (int)modalities_.size();
            quantized_images.create(1, L * M, CV_8U);
            for (int l = 0; l < L; ++l)
              for (int m = 0; m < M; ++m) {
                Mat& dst = quantized_images.getMatRef(l * M + m);
                dst.create(sources[0].rows >> l, sources[0].cols >> l, CV_8U);
                lmx_check(lmx_ctx_debug_read(ctx, 0, LMX_DBG_QUANTIZED, l, m, dst.data, (size_t)dst.rows * dst.cols));
              }
          }
        }
      }
      if (!regrow) break;
      if (ctx_) { lmx_ctx_unref(ctx_); ctx_ = NULL; }   // outside the lock: the next turn acquires a context with the larger lists
    }
  }

  // ---- training: Detector::addTemplate(sources, class_id, object_mask, bounding_box) -----------------------------------------
  int addTemplate(const std::vector<Mat>& sources, const String& class_id, const Mat& object_mask, Rect* bounding_box = NULL) {
    if (sources.size() != modalities_.size()) LMX_CV_THROW(LMX_ERR_SHAPE, "sources.size() != modalities.size()");
    std::vector<lmx_image> imgs(sources.size());
    for (size_t i = 0; i < sources.size(); ++i) {
      const Mat& m = sources[i];
      lmx_image im;
      im.data = m.data; im.rows = m.rows; im.cols = m.cols; im.channels = m.channels(); im.elem_size = (int32_t)m.elemSize1(); im.row_stride_bytes = m.step[0];
      imgs[i] = im;
    }
    lmx_image mask;
    const bool has_mask = !object_mask.empty();
    if (has_mask) { mask.data = object_mask.data; mask.rows = object_mask.rows; mask.cols = object_mask.cols; mask.channels = 1; mask.elem_size = 1; mask.row_stride_bytes = object_mask.step[0]; }
    int32_t tid = -1, bb[4] = {0, 0, 0, 0};
    lmx_check(lmx_bank_add_template(mutable_bank(), device_, &imgs[0], (int32_t)imgs.size(), class_id.c_str(), has_mask ? &mask : NULL, &tid, bb));
    if (tid >= 0) {
      invalidate();
      if (bounding_box) *bounding_box = Rect(bb[0], bb[1], bb[2], bb[3]);
    }
    return tid;
  }

  // ---- accessors --------------------------------------------------------------------------------------------------------------
  const std::vector<Ptr<Modality> >& getModalities() const { return modalities_; }
  int getT(int pyramid_level) const { return T_at_level_[(size_t)pyramid_level]; }
  int pyramidLevels() const { return (int)T_at_level_.size(); }
  const std::vector<Template>& getTemplates(const String& class_id, int template_id) const {
    const std::pair<std::string, int> key(std::string(class_id), template_id);
    std::map<std::pair<std::string, int>, std::vector<Template> >::const_iterator it = tcache_.find(key);
    if (it != tcache_.end()) return it->second;
    const int per = pyramidLevels() * (int)modalities_.size();
    std::vector<Template> out((size_t)per);
    for (int k = 0; k < per; ++k) {
      const int32_t* f = NULL;
      int32_t n = 0, w = 0, h = 0, l = 0;
      lmx_check(lmx_bank_get_template(bank(), class_id.c_str(), template_id, k, &w, &h, &l, &f, &n));
      out[(size_t)k].width = w; out[(size_t)k].height = h; out[(size_t)k].pyramid_level = l;
      for (int i = 0; i < n; ++i) out[(size_t)k].features.push_back(Feature(f[3 * i], f[3 * i + 1], f[3 * i + 2]));
    }
    return tcache_[key] = out;
  }
  int numTemplates() const { return view() ? lmx_bank_num_templates(view(), NULL) : 0; }
  int numTemplates(const String& class_id) const { return view() ? lmx_bank_num_templates(view(), class_id.c_str()) : 0; }
  int numClasses() const { return view() ? lmx_bank_num_classes(view()) : 0; }
  std::vector<String> classIds() const {
    std::vector<String> ids;
    for (int i = 0; i < numClasses(); ++i) ids.push_back(String(lmx_bank_class_id(view(), i)));
    return ids;
  }

  // ---- persistence: the reference's readLinemod / writeLinemod call exactly these --------------------------------------------
  void read(const FileNode& fn) {
    drop();
    modalities_.clear();
    T_at_level_.clear();
    const int L = (int)fn["pyramid_levels"];
    FileNode nt = fn["T"];
    for (FileNodeIterator it = nt.begin(), e = nt.end(); it != e; ++it) T_at_level_.push_back((int)(*it));
    if ((int)T_at_level_.size() != L) LMX_CV_THROW(LMX_ERR_PARSE, "Detector::read: T does not list pyramid_levels entries");
    FileNode nm = fn["modalities"];
    for (FileNodeIterator it = nm.begin(), e = nm.end(); it != e; ++it) modalities_.push_back(Modality::create(*it));
    // extension key written by liblmx (ignored by OpenCV): which NORMAL_LUT the DepthNormal templates were trained with
    // A yml without it that has a DepthNormal modality was trained against OpenCV's normal_lut.i, which liblmx does not contain:
    // the bank is then marked "table unknown" and match() refuses it until loadNormalLut / setNormalLut (or the environment
    // variable LMX_NORMAL_LUT) says which table to use.
    FileNode lut = fn["lmx_normal_lut"];
    const std::string marker = lut.empty() ? std::string() : std::string((String)lut);
    foreign_depth_ = false;
    for (size_t i = 0; i < modalities_.size(); ++i)
      if (modalities_[i]->name() == "DepthNormal" && marker != "default") foreign_depth_ = true;
  }
  void write(FileStorage& fs) const {
    fs << "pyramid_levels" << pyramidLevels();
    fs << "T" << T_at_level_;
    fs << "modalities" << "[";
    for (size_t i = 0; i < modalities_.size(); ++i) {
      fs << "{";
      modalities_[i]->write(fs);
      fs << "}";
    }
    fs << "]";
    // extension key (OpenCV's Detector::read ignores unknown keys): the NORMAL_LUT the DepthNormal templates were trained with
    for (size_t i = 0; i < modalities_.size(); ++i)
      if (modalities_[i]->name() == "DepthNormal") {
        fs << "lmx_normal_lut" << (lmx_bank_normal_lut_origin(bank()) == LMX_LUT_DEFAULT ? "default" : "external");
        break;
      }
  }
  String readClass(const FileNode& fn, const String& class_id_override = "") {
    // upstream verifies that the class was written with this detector's modalities and pyramid depth
    FileNode mod = fn["modalities"];
    if ((size_t)mod.size() != modalities_.size()) LMX_CV_THROW(LMX_ERR_PARSE, "readClass: modalities do not match the detector");
    size_t mi = 0;
    for (FileNodeIterator it = mod.begin(), e = mod.end(); it != e; ++it, ++mi)
      if ((String)(*it) != modalities_[mi]->name()) LMX_CV_THROW(LMX_ERR_PARSE, "readClass: modalities do not match the detector");
    if ((int)fn["pyramid_levels"] != pyramidLevels()) LMX_CV_THROW(LMX_ERR_PARSE, "readClass: pyramid_levels mismatch");
    String class_id = class_id_override.empty() ? (String)fn["class_id"] : class_id_override;
    const int per = pyramidLevels() * (int)modalities_.size();
    std::vector<int32_t> templates, features;
    int32_t n_pyr = 0;
    FileNode tps = fn["template_pyramids"];
    for (FileNodeIterator it = tps.begin(), e = tps.end(); it != e; ++it, ++n_pyr) {
      if ((int)(*it)["template_id"] != n_pyr) LMX_CV_THROW(LMX_ERR_PARSE, "readClass: template_id out of sequence");  // upstream CV_Assert
      FileNode tl = (*it)["templates"];
      if ((int)tl.size() != per) LMX_CV_THROW(LMX_ERR_PARSE, "readClass: wrong number of templates in a pyramid");
      for (FileNodeIterator jt = tl.begin(), je = tl.end(); jt != je; ++jt) {
        Template t;
        t.read(*jt);
        const int32_t fbegin = (int32_t)(features.size() / 3);
        for (size_t i = 0; i < t.features.size(); ++i) {
          features.push_back(t.features[i].x); features.push_back(t.features[i].y); features.push_back(t.features[i].label);
        }
        templates.push_back(t.width); templates.push_back(t.height); templates.push_back(t.pyramid_level);
        templates.push_back(fbegin); templates.push_back((int32_t)t.features.size());
      }
    }
    lmx_check(lmx_bank_add_class(mutable_bank(), class_id.c_str(), n_pyr, templates.empty() ? NULL : &templates[0], features.empty() ? NULL : &features[0],
                                 (int64_t)(features.size() / 3)));
    invalidate();
    return class_id;
  }
  void writeClass(const String& class_id, FileStorage& fs) const {
    fs << "class_id" << class_id;
    fs << "modalities" << "[:";
    for (size_t i = 0; i < modalities_.size(); ++i) fs << modalities_[i]->name();
    fs << "]";  // modalities
    fs << "pyramid_levels" << pyramidLevels();
    fs << "template_pyramids" << "[";
    const int n = numTemplates(class_id);
    for (int t = 0; t < n; ++t) {
      const std::vector<Template>& tp = getTemplates(class_id, t);
      fs << "{";
      fs << "template_id" << t;
      fs << "templates" << "[";
      for (size_t j = 0; j < tp.size(); ++j) {
        fs << "{";
        tp[j].write(fs);
        fs << "}";  // current template
      }
      fs << "]";  // templates
      fs << "}";  // current pyramid
    }
    fs << "]";  // pyramids
  }

  // ---- liblmx-specific knobs (no upstream analogue) -------------------------------------------------------------------------
  void setDevice(int device, int max_candidates = 0) { device_ = device; max_candidates_ = max_candidates; invalidate(); }
  void setNormalLut(const unsigned char* lut /* [20][20][20], NULL = default generator */) { lmx_check(lmx_bank_set_normal_lut(mutable_bank(), lut)); invalidate(); }
  void loadNormalLut(const String& path) { lmx_check(lmx_bank_load_normal_lut(mutable_bank(), path.c_str())); invalidate(); }
  const lmx_bank* bank() const { return shared_ ? shared_ : mutable_bank(); }
  bool contextWasCached() const { return ctx_cached_; }   // the device context of the last match() was already resident

 private:
  Detector(const Detector&);
  Detector& operator=(const Detector&);

  const lmx_bank* view() const { return shared_ ? shared_ : bank_; }   // what exists now (no lazy creation): the shared cached bank or the private one
  lmx_bank* mutable_bank() const {
    if (shared_) {   // a detector from load() is about to be modified: private copy, the shared bank goes back to the cache
      lmx_check(lmx_bank_clone(shared_, &bank_));
      lmx_bank_release(shared_);
      shared_ = NULL;
    }
    if (!bank_) {
      if (T_at_level_.empty() || modalities_.empty()) LMX_CV_THROW(LMX_ERR_INVALID_ARG, "cv::linemod::Detector used before read() / construction with modalities");
      std::vector<lmx_modality_desc> md;
      for (size_t i = 0; i < modalities_.size(); ++i) md.push_back(modalities_[i]->desc());
      std::vector<int32_t> T(T_at_level_.begin(), T_at_level_.end());
      lmx_bank_desc bd;
      bd.pyramid_levels = (int32_t)T.size(); bd.T = &T[0]; bd.n_modalities = (int32_t)md.size(); bd.modalities = &md[0];
      lmx_check(lmx_bank_create(&bd, &bank_));
      if (foreign_depth_) lmx_check(lmx_bank_require_normal_lut(bank_));   // LMX_NORMAL_LUT from the environment, else "unknown"
    }
    return bank_;
  }
  // the device context: process-wide cache keyed by the bank's content and the frame size (see header comment)
  lmx_ctx* context(int w, int h) const {
    if (ctx_ && w == ctx_w_ && h == ctx_h_) return ctx_;
    if (ctx_) { lmx_ctx_unref(ctx_); ctx_ = NULL; }
    lmx_ctx_desc d;
    std::memset(&d, 0, sizeof(d));
    d.device = device_; d.width = w; d.height = h; d.max_batch = 1; d.max_candidates = max_candidates_;
    int32_t hit = 0;
    lmx_check(lmx_ctx_acquire(bank(), &d, &ctx_, &hit));
    ctx_cached_ = hit != 0;
    ctx_w_ = w; ctx_h_ = h;
    return ctx_;
  }
  void invalidate() const {
    if (ctx_) { lmx_ctx_unref(ctx_); ctx_ = NULL; }
    tcache_.clear();
  }
  void drop() {
    invalidate();
    if (bank_) { lmx_bank_destroy(bank_); bank_ = NULL; }
    if (shared_) { lmx_bank_release(shared_); shared_ = NULL; }
  }

  std::vector<Ptr<Modality> > modalities_;
  std::vector<int> T_at_level_;
  bool foreign_depth_;
  mutable lmx_bank* bank_;           // private, modifiable bank (built by read / readClass / addTemplate), or NULL
  mutable const lmx_bank* shared_;   // load(): the cached bank of a templates file, shared and read-only, or NULL
  mutable lmx_ctx* ctx_;
  mutable int ctx_w_, ctx_h_;
  mutable bool ctx_cached_ = false;
  int device_;
  mutable int max_candidates_;
  mutable std::vector<lmx_match_t> buf_;
  mutable std::map<std::pair<std::string, int>, std::vector<Template> > tcache_;
};

// Factory functions of upstream (modalities with default parameters, T = {5, 8}).
inline Ptr<Detector> getDefaultLINE() {
  std::vector<Ptr<Modality> > m;
  m.push_back(Ptr<Modality>(new ColorGradient()));
  static const int T_DEFAULTS[] = {5, 8};
  return Ptr<Detector>(new Detector(m, std::vector<int>(T_DEFAULTS, T_DEFAULTS + 2)));
}
inline Ptr<Detector> getDefaultLINEMOD() {
  std::vector<Ptr<Modality> > m;
  m.push_back(Ptr<Modality>(new ColorGradient()));
  m.push_back(Ptr<Modality>(new DepthNormal()));
  static const int T_DEFAULTS[] = {5, 8};
  return Ptr<Detector>(new Detector(m, std::vector<int>(T_DEFAULTS, T_DEFAULTS + 2)));
}

}  // namespace lmx_linemod
}  // namespace cv

#ifndef LMX_KEEP_CV_LINEMOD
#define linemod lmx_linemod
#endif

#endif  // LMX_CV_LINEMOD_HPP_
