// tests/cpp/cv_standin/opencv2/opencv.hpp -- TEST INFRASTRUCTURE: a minimal stand-in for the handful of OpenCV core types that
// include/lmx_cv_linemod.hpp and a caller written in the reference's style touch.  This image has no OpenCV, so the
// cv::linemod-shaped facade is compiled against these instead; with a real OpenCV the facade uses the real types and this file is
// not involved.  Only what the call surface needs is here: cv::String, cv::Ptr, cv::Mat (header over caller memory or an owned
// buffer), cv::Rect / Point, cv::OutputArrayOfArrays + noArray(), and FileStorage / FileNode / FileNodeIterator with the
// operators the reference's readLinemod / writeLinemod use (read side backed by liblmx's FileStorage-YAML parser through
// lmx_yaml_*, write side emitting FileStorage's block/flow YAML).  It also declares an (unusable) cv::linemod::Detector the way a
// real OpenCV would, so that the facade's `#define linemod lmx_linemod` is exercised against an existing namespace.
#ifndef LMX_TEST_CV_STANDIN_HPP_
#define LMX_TEST_CV_STANDIN_HPP_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmx.h"

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)

namespace cv {

typedef std::string String;

template <typename T>
class Ptr {
 public:
  Ptr() {}
  explicit Ptr(T* p) : p_(p) {}
  template <typename U>
  Ptr(const Ptr<U>& o) : p_(o.shared()) {}
  T* operator->() const { return p_.get(); }
  T& operator*() const { return *p_; }
  bool empty() const { return !p_; }
  operator T*() const { return p_.get(); }
  const std::shared_ptr<T>& shared() const { return p_; }

 private:
  std::shared_ptr<T> p_;
};

struct Point { int x, y; Point() : x(0), y(0) {} Point(int x_, int y_) : x(x_), y(y_) {} };
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

class Mat {
 public:
  struct Step { size_t p[2]; size_t operator[](int i) const { return p[i]; } operator size_t() const { return p[0]; } };
  Mat() : rows(0), cols(0), data(NULL), type_(0) { step.p[0] = step.p[1] = 0; }
  Mat(int r, int c, int type, void* d, size_t step_bytes = 0) : rows(r), cols(c), data((unsigned char*)d), type_(type) {
    step.p[1] = elemSize();
    step.p[0] = step_bytes ? step_bytes : (size_t)c * elemSize();
  }
  Mat(int r, int c, int type) : rows(0), cols(0), data(NULL), type_(0) { create(r, c, type); }
  void create(int r, int c, int type) {
    type_ = type; rows = r; cols = c;
    step.p[1] = elemSize(); step.p[0] = (size_t)c * elemSize();
    own_.reset(new std::vector<unsigned char>((size_t)r * step.p[0]));
    data = own_->data();
  }
  // ROI view like Mat::operator()(Rect): shares the pixels, keeps the parent's row stride (..._service.cpp:324-326)
  Mat operator()(const Rect& r) const {
    Mat m;
    m.rows = r.height; m.cols = r.width; m.type_ = type_; m.step = step; m.own_ = own_;
    m.data = data + (size_t)r.y * step.p[0] + (size_t)r.x * elemSize();
    return m;
  }
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return (type_ >> 3) + 1; }
  size_t elemSize1() const { static const size_t s[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return s[depth()]; }
  size_t elemSize() const { return elemSize1() * channels(); }
  bool empty() const { return data == NULL || rows == 0 || cols == 0; }
  template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step.p[0] + (size_t)x * sizeof(T)); }
  template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step.p[0] + (size_t)x * sizeof(T)); }
  int rows, cols;
  unsigned char* data;
  Step step;

 private:
  int type_;
  std::shared_ptr<std::vector<unsigned char> > own_;
};

// OutputArrayOfArrays: either "nothing wanted" (noArray()) or a std::vector<Mat> to fill
class _OutputArray {
 public:
  _OutputArray() : v_(NULL) {}
  _OutputArray(std::vector<Mat>& v) : v_(&v) {}
  bool needed() const { return v_ != NULL; }
  void create(int /*rows*/, int n, int /*type*/) const { v_->assign((size_t)n, Mat()); }
  Mat& getMatRef(int i) const { return (*v_)[(size_t)i]; }

 private:
  std::vector<Mat>* v_;
};
typedef const _OutputArray& OutputArrayOfArrays;
typedef const _OutputArray& OutputArray;
inline const _OutputArray& noArray() { static const _OutputArray none; return none; }

// ---- FileStorage -----------------------------------------------------------------------------------------------------------
class FileNode;
class FileNodeIterator {
 public:
  FileNodeIterator() : n_(NULL), i_(0) {}
  FileNodeIterator(const lmx_yaml_node* n, int i) : n_(n), i_(i) {}
  FileNode operator*() const;
  FileNodeIterator& operator++() { ++i_; return *this; }
  bool operator!=(const FileNodeIterator& o) const { return n_ != o.n_ || i_ != o.i_; }
  bool operator==(const FileNodeIterator& o) const { return !(*this != o); }

 private:
  const lmx_yaml_node* n_;
  int i_;
};

class FileNode {
 public:
  FileNode() : n_(NULL) {}
  explicit FileNode(const lmx_yaml_node* n) : n_(n) {}
  FileNode operator[](const char* key) const { return FileNode(lmx_yaml_get(n_, key)); }
  FileNode operator[](const String& key) const { return FileNode(lmx_yaml_get(n_, key.c_str())); }
  FileNode operator[](int i) const { return FileNode(lmx_yaml_item(n_, i)); }
  bool empty() const { return n_ == NULL || lmx_yaml_kind(n_) == LMX_YAML_NULL; }
  size_t size() const { return lmx_yaml_kind(n_) == LMX_YAML_SCALAR ? 1 : (size_t)lmx_yaml_size(n_); }
  FileNodeIterator begin() const { return FileNodeIterator(n_, 0); }
  FileNodeIterator end() const { return FileNodeIterator(n_, lmx_yaml_size(n_)); }
  operator int() const { return empty() ? 0 : (int)std::strtod(lmx_yaml_scalar(n_), NULL); }
  operator float() const { return empty() ? 0.f : (float)std::strtod(lmx_yaml_scalar(n_), NULL); }
  operator double() const { return empty() ? 0.0 : std::strtod(lmx_yaml_scalar(n_), NULL); }
  operator String() const { return String(lmx_yaml_scalar(n_)); }

 private:
  const lmx_yaml_node* n_;
};
inline FileNode FileNodeIterator::operator*() const { return FileNode(lmx_yaml_item(n_, i_)); }
inline void operator>>(const FileNode& n, int& v) { v = (int)n; }
inline void operator>>(const FileNode& n, double& v) { v = (double)n; }
inline void operator>>(const FileNode& n, String& v) { v = (String)n; }

class FileStorage {
 public:
  enum { READ = 0, WRITE = 1 };
  FileStorage(const String& filename, int flags) : doc_(NULL), f_(NULL) {
    if (flags == READ) {
      if (lmx_yaml_open(filename.c_str(), &doc_) != LMX_OK) doc_ = NULL;  // isOpened() == false, like OpenCV
    } else {
      f_ = std::fopen(filename.c_str(), "wb");
      if (f_) std::fputs("%YAML:1.0\n---\n", f_);
      stack_.push_back(Frame{'{', false, false, 0});
    }
  }
  ~FileStorage() { release(); }
  bool isOpened() const { return doc_ != NULL || f_ != NULL; }
  void release() {
    if (doc_) { lmx_yaml_close(doc_); doc_ = NULL; }
    if (f_) { std::fclose(f_); f_ = NULL; }
  }
  FileNode root() const { return FileNode(lmx_yaml_root(doc_)); }
  FileNode operator[](const char* key) const { return root()[key]; }
  FileNode operator[](const String& key) const { return root()[key]; }

  // ---- writing: the streaming protocol of cv::FileStorage (keys and values alternate inside maps; "[" "]" "{" "}" open and
  // close collections, "[:" / "{:" open flow collections) -----------------------------------------------------------------------
  void put_string(const String& s) {
    if (!f_) return;
    Frame& top = stack_.back();
    const bool in_map = top.kind == '{';
    if (s == "]" || s == "}") {
      const bool was_flow = top.flow;
      stack_.pop_back();
      if (was_flow) {
        std::fputs(s == "]" ? " ]" : " }", f_);
        if (!stack_.back().flow) std::fputc('\n', f_);
      }
      return;
    }
    if (in_map && !top.have_key) {        // a key
      key_ = s; top.have_key = true;
      return;
    }
    if (s == "[" || s == "{" || s == "[:" || s == "{:") {
      const bool flow = s.size() == 2 || top.flow;
      begin_value(flow ? NULL : "\n");
      if (flow) std::fputs(s[0] == '[' ? "[ " : "{ ", f_);
      Frame fr;
      fr.kind = s[0]; fr.flow = flow; fr.have_key = false; fr.count = 0;
      stack_.push_back(fr);
      return;
    }
    begin_value(NULL);
    std::fputs(s.c_str(), f_);
    end_value();
  }
  void put_scalar(const char* text) {
    if (!f_) return;
    begin_value(NULL);
    std::fputs(text, f_);
    end_value();
  }

 private:
  struct Frame { char kind; bool flow; bool have_key; int count; };
  int depth() const { return (int)stack_.size() - 1; }
  void indent(int d) { for (int i = 0; i < 3 * d; ++i) std::fputc(' ', f_); }
  // emits what precedes a value: "key: " in a map, "- " (block) or ", " (flow) in a sequence.  `after_key` = text to put right
  // after the key/dash when the value is a block collection (it continues on the following lines)
  void begin_value(const char* after_key) {
    Frame& top = stack_.back();
    if (top.flow) {
      if (top.count++) std::fputs(", ", f_);
      if (top.kind == '{') { std::fprintf(f_, "%s: ", key_.c_str()); top.have_key = false; }
      return;
    }
    indent(depth());
    if (top.kind == '{') { std::fprintf(f_, "%s:%s", key_.c_str(), after_key ? after_key : " "); top.have_key = false; }
    else std::fprintf(f_, "-%s", after_key ? after_key : " ");
    top.count++;
  }
  void end_value() { if (!stack_.back().flow) std::fputc('\n', f_); }

  lmx_yaml_doc* doc_;
  FILE* f_;
  std::vector<Frame> stack_;
  String key_;
  FileStorage(const FileStorage&);
  FileStorage& operator=(const FileStorage&);
};
inline FileStorage& operator<<(FileStorage& fs, const char* s) { fs.put_string(s); return fs; }
inline FileStorage& operator<<(FileStorage& fs, const String& s) { fs.put_string(s); return fs; }
inline FileStorage& operator<<(FileStorage& fs, int v) { char b[32]; std::snprintf(b, sizeof(b), "%d", v); fs.put_scalar(b); return fs; }
inline FileStorage& operator<<(FileStorage& fs, float v) {
  char b[48];
  if (v == (float)(long)v) std::snprintf(b, sizeof(b), "%ld.", (long)v);   // FileStorage writes 10.f as "10."
  else std::snprintf(b, sizeof(b), "%.8e", (double)v);
  fs.put_scalar(b);
  return fs;
}
inline FileStorage& operator<<(FileStorage& fs, double v) { return fs << (float)v; }
inline FileStorage& operator<<(FileStorage& fs, const std::vector<int>& v) {
  fs << "[:";
  for (size_t i = 0; i < v.size(); ++i) fs << v[i];
  return fs << "]";
}

// what a real OpenCV declares under this name (objdetect in 2.4, rgbd contrib later): present so that the facade's macro has
// an existing cv::linemod to step around; deliberately unusable
namespace linemod {
class Detector;
struct Match;
}  // namespace linemod

}  // namespace cv

#endif  // LMX_TEST_CV_STANDIN_HPP_
