// liblmx.so, host bank: the template state of cv::linemod::Detector behind the C ABI of include/lmx.h --
// lmx_bank_create / _add_class / accessors / NORMAL_LUT handling -- and the library's error string.
// Reference call sites: /root/reference/src/renderer.cpp:179-185,308 (ctor, addTemplate), src/rgbdDetector.cpp:1668-1680 (readLinemod).

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

}  // namespace lmx

using namespace lmx;

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

}  // extern "C"
