// Template-bank wire format: the OpenCV FileStorage YAML 1.0 files the reference reads and writes
//   readLinemod  /root/reference/src/rgbdDetector.cpp:1668-1680  (Detector::read(fs.root()) + readClass per class)
//   writeLinemod /root/reference/src/renderer.cpp:56-70           (Detector::write + writeClass)
// Layout: SURVEY.md Appendix B.1.  OpenCV itself is not available here, so this is a small stand-alone
// reader for the YAML subset FileStorage emits (block maps/sequences by indentation, flow `[ .. ]` / `{ .. }`
// collections possibly wrapped over lines, plain or double-quoted scalars, `%YAML:1.0` / `---` headers) and a
// writer that reproduces FileStorage's block style (3-space indentation, "-" on its own line before a map).

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "lmx_internal.hpp"

namespace lmx {
namespace {

struct Node {
  enum Kind { Null, Scalar, Seq, Map } kind = Null;
  std::string scalar;
  std::vector<Node> items;                            // Seq
  std::vector<std::pair<std::string, Node>> entries;  // Map
  const Node* get(const char* key) const {
    for (const auto& e : entries)
      if (e.first == key) return &e.second;
    return nullptr;
  }
};

struct Line { int indent; std::string text; int lineno; };

struct Parser {
  std::vector<Line> lines;
  size_t pos = 0;
  std::string err;

  bool fail(const std::string& m, int lineno) {
    if (err.empty()) err = "line " + std::to_string(lineno) + ": " + m;
    return false;
  }

  static std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
  }
  static std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
  }

  void load(const char* buf, size_t n) {
    size_t i = 0;
    int lineno = 0;
    while (i < n) {
      size_t e = i;
      while (e < n && buf[e] != '\n') ++e;
      ++lineno;
      std::string raw(buf + i, e - i);
      i = e + 1;
      // strip comments outside quotes
      bool inq = false;
      char qc = 0;
      for (size_t k = 0; k < raw.size(); ++k) {
        char ch = raw[k];
        if (inq) { if (ch == qc) inq = false; }
        else if (ch == '"' || ch == '\'') { inq = true; qc = ch; }
        else if (ch == '#' && (k == 0 || raw[k - 1] == ' ')) { raw.resize(k); break; }
      }
      int indent = 0;
      while ((size_t)indent < raw.size() && raw[indent] == ' ') ++indent;
      std::string t = trim(raw);
      if (t.empty() || t[0] == '%' || t == "---" || t == "...") continue;
      lines.push_back({indent, t, lineno});
    }
  }

  // ---- flow collections -------------------------------------------------------------------------------
  static int depth_delta(const std::string& s) {
    int d = 0;
    bool inq = false;
    for (char ch : s) {
      if (ch == '"') inq = !inq;
      if (inq) continue;
      if (ch == '[' || ch == '{') ++d;
      if (ch == ']' || ch == '}') --d;
    }
    return d;
  }
  // gathers the continuation lines of a flow collection starting with `first`
  std::string gather_flow(const std::string& first) {
    std::string s = first;
    int d = depth_delta(s);
    while (d > 0 && pos < lines.size()) {
      s += " " + lines[pos].text;
      d += depth_delta(lines[pos].text);
      ++pos;
    }
    return s;
  }
  bool parse_flow(const std::string& s, size_t& i, Node& out, int lineno) {
    auto skip = [&]() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i; };
    skip();
    if (i >= s.size()) return fail("unexpected end of flow collection", lineno);
    if (s[i] == '[') {
      out.kind = Node::Seq;
      ++i;
      for (;;) {
        skip();
        if (i >= s.size()) return fail("unterminated [", lineno);
        if (s[i] == ']') { ++i; return true; }
        Node item;
        if (!parse_flow(s, i, item, lineno)) return false;
        out.items.push_back(std::move(item));
        skip();
        if (i < s.size() && s[i] == ',') ++i;
        else if (i < s.size() && s[i] != ']') return fail("flow sequence: ',' or ']' expected", lineno);   // e.g. a stray '}': no progress otherwise
      }
    }
    if (s[i] == '{') {
      out.kind = Node::Map;
      ++i;
      for (;;) {
        skip();
        if (i >= s.size()) return fail("unterminated {", lineno);
        if (s[i] == '}') { ++i; return true; }
        size_t k = i;
        while (k < s.size() && s[k] != ':' && s[k] != ',' && s[k] != '}') ++k;
        if (k >= s.size() || s[k] != ':') return fail("flow map entry without ':'", lineno);
        std::string key = unquote(trim(s.substr(i, k - i)));
        i = k + 1;
        Node val;
        if (!parse_flow(s, i, val, lineno)) return false;
        out.entries.emplace_back(key, std::move(val));
        skip();
        if (i < s.size() && s[i] == ',') ++i;
        else if (i < s.size() && s[i] != '}') return fail("flow map: ',' or '}' expected", lineno);
      }
    }
    // scalar up to , ] }
    size_t k = i;
    if (s[i] == '"') {
      k = i + 1;
      while (k < s.size() && s[k] != '"') ++k;
      if (k < s.size()) ++k;
    } else {
      while (k < s.size() && s[k] != ',' && s[k] != ']' && s[k] != '}') ++k;
    }
    out.kind = Node::Scalar;
    out.scalar = unquote(trim(s.substr(i, k - i)));
    i = k;
    return true;
  }

  bool parse_value_text(const std::string& rest, Node& out, int lineno) {
    if (rest[0] == '[' || rest[0] == '{') {
      std::string s = gather_flow(rest);
      size_t i = 0;
      return parse_flow(s, i, out, lineno);
    }
    out.kind = Node::Scalar;
    out.scalar = unquote(rest);
    return true;
  }

  // finds "key:" at the start of a line text; returns false if the text is not a map entry
  static bool split_key(const std::string& t, std::string& key, std::string& rest) {
    if (t.empty() || t[0] == '[' || t[0] == '{' || t[0] == '-') return false;
    size_t i = 0;
    if (t[0] == '"') {
      i = t.find('"', 1);
      if (i == std::string::npos) return false;
      ++i;
    } else {
      while (i < t.size() && t[i] != ':') ++i;
    }
    if (i >= t.size() || t[i] != ':') return false;
    if (i + 1 < t.size() && t[i + 1] != ' ') return false;  // "a:b" is a scalar
    key = unquote(trim(t.substr(0, i)));
    rest = trim(t.substr(i + 1));
    return true;
  }

  bool parse_block(int indent, Node& out) {
    if (pos >= lines.size()) { out.kind = Node::Null; return true; }
    const Line& first = lines[pos];
    if (first.text[0] == '-' && (first.text.size() == 1 || first.text[1] == ' ')) return parse_seq(indent, out);
    std::string k, r;
    if (split_key(first.text, k, r)) return parse_map(indent, out);
    // bare scalar / flow on its own line
    std::string t = first.text;
    int ln = first.lineno;
    ++pos;
    return parse_value_text(t, out, ln);
  }

  bool parse_seq(int indent, Node& out) {
    out.kind = Node::Seq;
    while (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
           (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
      Line ln = lines[pos];
      std::string rest = trim(ln.text.substr(1));
      Node item;
      if (rest.empty()) {
        ++pos;
        if (pos < lines.size() && lines[pos].indent > indent) {
          if (!parse_block(lines[pos].indent, item)) return false;
        }
      } else {
        std::string k, r;
        if (split_key(rest, k, r)) {
          // "- key: value": a map whose first entry sits on the dash line
          int col = indent + (int)(ln.text.size() - rest.size());
          lines[pos].indent = col;
          lines[pos].text = rest;
          if (!parse_map(col, item)) return false;
        } else {
          ++pos;
          if (!parse_value_text(rest, item, ln.lineno)) return false;
        }
      }
      out.items.push_back(std::move(item));
    }
    return true;
  }

  // `key: !!opencv-matrix` (FileStorage writes a cv::Mat as a tagged map: rows, cols, dt, data): the tag carries no information a
  // reader of these files needs, so it is dropped and the value parsed as what follows it
  static void strip_tag(std::string& rest) {
    if (rest.size() >= 2 && rest[0] == '!' && rest[1] == '!') {
      size_t k = 2;
      while (k < rest.size() && rest[k] != ' ') ++k;
      rest = trim(rest.substr(k));
    }
  }

  bool parse_map(int indent, Node& out) {
    out.kind = Node::Map;
    while (pos < lines.size() && lines[pos].indent == indent) {
      std::string key, rest;
      if (!split_key(lines[pos].text, key, rest)) break;
      strip_tag(rest);
      int ln = lines[pos].lineno;
      ++pos;
      Node val;
      if (rest.empty()) {
        if (pos < lines.size() && (lines[pos].indent > indent ||
                                   (lines[pos].indent == indent && lines[pos].text[0] == '-' &&
                                    (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')))) {
          if (!parse_block(lines[pos].indent, val)) return false;
        }
      } else {
        if (!parse_value_text(rest, val, ln)) return false;
      }
      out.entries.emplace_back(key, std::move(val));
    }
    if (pos < lines.size() && lines[pos].indent > indent) return fail("unexpected indentation", lines[pos].lineno);
    return true;
  }
};

bool to_int(const Node* n, int32_t* v) {
  if (!n || n->kind != Node::Scalar || n->scalar.empty()) return false;
  char* end = nullptr;
  errno = 0;
  double d = std::strtod(n->scalar.c_str(), &end);  // OpenCV writes ints plainly, but tolerate "63."
  if (errno || end == n->scalar.c_str()) return false;
  if (!(d >= -2147483648.0 && d <= 2147483647.0)) return false;   // also NaN: a value an int32 cannot hold is a parse error, not a cast
  *v = (int32_t)d;
  return true;
}
bool to_float(const Node* n, float* v) {
  if (!n || n->kind != Node::Scalar || n->scalar.empty()) return false;
  char* end = nullptr;
  errno = 0;
  double d = std::strtod(n->scalar.c_str(), &end);
  if (errno || end == n->scalar.c_str()) return false;
  *v = (float)d;
  return true;
}

const char* mod_name(int type) { return type == LMX_MOD_COLOR_GRADIENT ? "ColorGradient" : "DepthNormal"; }

void write_float(FILE* f, float v) {
  if (v > -1e9f && v < 1e9f && v == (float)(long)v) std::fprintf(f, "%ld.", (long)v);   // range first: (long)v of a huge v is undefined
  else std::fprintf(f, "%.8e", (double)v);
}

}  // namespace

static lmx_status parse_file(const char* path, Node& root) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { set_error("cannot open '%s': %s", path, std::strerror(errno)); return LMX_ERR_IO; }
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
  std::fclose(f);

  Parser p;
  p.load(buf.data(), buf.size());
  if (p.lines.empty()) { set_error("'%s': empty document", path); return LMX_ERR_PARSE; }
  if (!p.parse_block(p.lines[0].indent, root) || root.kind != Node::Map) {
    set_error("'%s': %s", path, p.err.empty() ? "top level is not a map" : p.err.c_str());
    return LMX_ERR_PARSE;
  }
  if (p.pos < p.lines.size()) { set_error("'%s': line %d: trailing content", path, p.lines[p.pos].lineno); return LMX_ERR_PARSE; }
  return LMX_OK;
}

lmx_status yaml_load(const char* path, lmx_bank** out) {
  if (!path || !out) { set_error("lmx_bank_load_yaml: null argument"); return LMX_ERR_INVALID_ARG; }
  Node root;
  lmx_status pst = parse_file(path, root);
  if (pst != LMX_OK) return pst;

  // Detector::read: pyramid_levels, T, modalities[] (type + parameters)
  int32_t L = 0;
  if (!to_int(root.get("pyramid_levels"), &L)) { set_error("'%s': missing pyramid_levels", path); return LMX_ERR_PARSE; }
  const Node* nT = root.get("T");
  if (!nT || nT->kind != Node::Seq || (int)nT->items.size() != L) { set_error("'%s': T must list pyramid_levels=%d entries", path, L); return LMX_ERR_PARSE; }
  std::vector<int32_t> T(L);
  for (int l = 0; l < L; ++l)
    if (!to_int(&nT->items[l], &T[l])) { set_error("'%s': bad T entry", path); return LMX_ERR_PARSE; }
  const Node* nm = root.get("modalities");
  if (!nm || nm->kind != Node::Seq || nm->items.empty()) { set_error("'%s': missing modalities", path); return LMX_ERR_PARSE; }
  std::vector<lmx_modality_desc> mods;
  for (const Node& m : nm->items) {
    const Node* ty = m.get("type");
    if (m.kind != Node::Map || !ty || ty->kind != Node::Scalar) { set_error("'%s': modality without type", path); return LMX_ERR_PARSE; }
    lmx_modality_desc d{};
    if (ty->scalar == "ColorGradient") {
      d.type = LMX_MOD_COLOR_GRADIENT;
      if (!to_float(m.get("weak_threshold"), &d.weak_threshold) || !to_int(m.get("num_features"), &d.num_features) ||
          !to_float(m.get("strong_threshold"), &d.strong_threshold)) { set_error("'%s': incomplete ColorGradient parameters", path); return LMX_ERR_PARSE; }
    } else if (ty->scalar == "DepthNormal") {
      d.type = LMX_MOD_DEPTH_NORMAL;
      if (!to_int(m.get("distance_threshold"), &d.distance_threshold) || !to_int(m.get("difference_threshold"), &d.difference_threshold) ||
          !to_int(m.get("num_features"), &d.num_features) || !to_int(m.get("extract_threshold"), &d.extract_threshold)) {
        set_error("'%s': incomplete DepthNormal parameters", path);
        return LMX_ERR_PARSE;
      }
    } else {
      set_error("'%s': unsupported modality type '%s'", path, ty->scalar.c_str());
      return LMX_ERR_PARSE;
    }
    mods.push_back(d);
  }
  lmx_bank_desc bd{L, T.data(), (int32_t)mods.size(), mods.data()};
  lmx_bank* bank = nullptr;
  lmx_status st = lmx_bank_create(&bd, &bank);
  if (st != LMX_OK) return st;
  std::unique_ptr<lmx_bank> guard(bank);
  const int M = (int)mods.size(), per = L * M;

  // readClass per entry of classes[]
  const Node* nc = root.get("classes");
  if (nc && nc->kind == Node::Seq) {
    for (const Node& c : nc->items) {
      const Node* cid = c.get("class_id");
      if (c.kind != Node::Map || !cid || cid->kind != Node::Scalar) { set_error("'%s': class without class_id", path); return LMX_ERR_PARSE; }
      const Node* cm = c.get("modalities");
      if (!cm || cm->kind != Node::Seq || (int)cm->items.size() != M) { set_error("'%s': class '%s': modalities do not match the detector", path, cid->scalar.c_str()); return LMX_ERR_PARSE; }
      for (int m = 0; m < M; ++m)
        if (cm->items[m].scalar != mod_name(mods[m].type)) { set_error("'%s': class '%s': modality %d is '%s', detector has '%s'", path, cid->scalar.c_str(), m, cm->items[m].scalar.c_str(), mod_name(mods[m].type)); return LMX_ERR_PARSE; }
      int32_t cl = 0;
      if (!to_int(c.get("pyramid_levels"), &cl) || cl != L) { set_error("'%s': class '%s': pyramid_levels mismatch", path, cid->scalar.c_str()); return LMX_ERR_PARSE; }
      const Node* tps = c.get("template_pyramids");
      std::vector<int32_t> templates, features;
      int32_t n_pyr = 0;
      if (tps && tps->kind == Node::Seq) {
        templates.reserve(tps->items.size() * per * 5);
        for (const Node& tp : tps->items) {
          int32_t tid = -1;
          if (!to_int(tp.get("template_id"), &tid) || tid != n_pyr) { set_error("'%s': class '%s': template_id %d where %d was expected", path, cid->scalar.c_str(), tid, n_pyr); return LMX_ERR_PARSE; }
          const Node* tl = tp.get("templates");
          if (!tl || tl->kind != Node::Seq || (int)tl->items.size() != per) { set_error("'%s': class '%s' template %d: expected %d templates", path, cid->scalar.c_str(), tid, per); return LMX_ERR_PARSE; }
          for (const Node& t : tl->items) {
            int32_t w, h, lv;
            if (!to_int(t.get("width"), &w) || !to_int(t.get("height"), &h) || !to_int(t.get("pyramid_level"), &lv)) { set_error("'%s': class '%s' template %d: missing width/height/pyramid_level", path, cid->scalar.c_str(), tid); return LMX_ERR_PARSE; }
            const Node* fs = t.get("features");
            const int32_t fbegin = (int32_t)(features.size() / 3);
            int32_t fcount = 0;
            if (fs && fs->kind == Node::Seq) {
              for (const Node& ft : fs->items) {
                int32_t x, y, lab;
                if (ft.kind != Node::Seq || ft.items.size() != 3 || !to_int(&ft.items[0], &x) || !to_int(&ft.items[1], &y) || !to_int(&ft.items[2], &lab)) {
                  set_error("'%s': class '%s' template %d: malformed feature", path, cid->scalar.c_str(), tid);
                  return LMX_ERR_PARSE;
                }
                features.insert(features.end(), {x, y, lab});
                ++fcount;
              }
            }
            templates.insert(templates.end(), {w, h, lv, fbegin, fcount});
          }
          ++n_pyr;
        }
      }
      st = lmx_bank_add_class(bank, cid->scalar.c_str(), n_pyr, templates.data(), features.data(), (int64_t)(features.size() / 3));
      if (st != LMX_OK) return st;
    }
  }
  // NORMAL_LUT of the DepthNormal modality (include/lmx.h): marker written by yaml_save, side-car next to the yml, the
  // environment's table, or unknown (a bank trained by OpenCV, whose table this library does not contain)
  bool has_dn = false;
  for (const lmx_modality_desc& d : mods) has_dn = has_dn || d.type == LMX_MOD_DEPTH_NORMAL;
  const Node* marker = root.get("lmx_normal_lut");
  const std::string sidecar = std::string(path) + ".normal_lut";
  FILE* sf = std::fopen(sidecar.c_str(), "rb");
  if (sf) std::fclose(sf);
  const char* env = std::getenv("LMX_NORMAL_LUT");
  if (sf || (marker && marker->kind == Node::Scalar && marker->scalar == "sidecar")) {
    std::vector<uint8_t> lut;
    if ((st = normal_lut_from_file(sidecar.c_str(), lut)) != LMX_OK) return st;
    bank->normal_lut = lut;
    bank->normal_lut_origin = LMX_LUT_SIDECAR; bank->lut_epoch += 1;
  } else if (marker && marker->kind == Node::Scalar && marker->scalar == "default") {
    bank->normal_lut_origin = LMX_LUT_DEFAULT; bank->lut_epoch += 1;
  } else if (has_dn && env && *env) {
    std::vector<uint8_t> lut;
    if ((st = normal_lut_from_file(env, lut)) != LMX_OK) return st;
    bank->normal_lut = lut;
    bank->normal_lut_origin = LMX_LUT_SIDECAR; bank->lut_epoch += 1;
  } else if (has_dn) {
    bank->normal_lut_origin = LMX_LUT_UNKNOWN; bank->lut_epoch += 1;
  }
  *out = guard.release();
  return LMX_OK;
}

lmx_status yaml_save(const lmx_bank* bank, const char* path) {
  if (!bank || !path) { set_error("lmx_bank_save_yaml: null argument"); return LMX_ERR_INVALID_ARG; }
  FILE* f = std::fopen(path, "wb");
  if (!f) { set_error("cannot open '%s' for writing: %s", path, std::strerror(errno)); return LMX_ERR_IO; }
  std::vector<char> iobuf(1 << 20);
  std::setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
  const int L = (int)bank->T.size(), M = (int)bank->mods.size(), per = L * M;
  std::fprintf(f, "%%YAML:1.0\n---\n");
  std::fprintf(f, "pyramid_levels: %d\n", L);
  std::fprintf(f, "T: [ ");
  for (int l = 0; l < L; ++l) std::fprintf(f, "%s%d", l ? ", " : "", bank->T[l]);
  std::fprintf(f, " ]\nmodalities:\n");
  for (const lmx_modality_desc& d : bank->mods) {
    std::fprintf(f, "   -\n      type: %s\n", mod_name(d.type));
    if (d.type == LMX_MOD_COLOR_GRADIENT) {
      std::fprintf(f, "      weak_threshold: "); write_float(f, d.weak_threshold);
      std::fprintf(f, "\n      num_features: %d\n      strong_threshold: ", d.num_features); write_float(f, d.strong_threshold);
      std::fprintf(f, "\n");
    } else {
      std::fprintf(f, "      distance_threshold: %d\n      difference_threshold: %d\n      num_features: %d\n      extract_threshold: %d\n",
                   d.distance_threshold, d.difference_threshold, d.num_features, d.extract_threshold);
    }
  }
  // extra key (ignored by OpenCV's Detector::read): which NORMAL_LUT the DepthNormal templates were trained with
  bool has_dn = false;
  for (const lmx_modality_desc& d : bank->mods) has_dn = has_dn || d.type == LMX_MOD_DEPTH_NORMAL;
  std::vector<uint8_t> def(LMX_NORMAL_LUT_SIZE);
  default_normal_lut(def.data());
  const bool lut_is_default = bank->normal_lut == def;
  if (has_dn) std::fprintf(f, "lmx_normal_lut: %s\n", lut_is_default ? "default" : "sidecar");
  std::fprintf(f, "classes:\n");
  for (const auto& kv : bank->classes) {
    const ClassData& cd = kv.second;
    std::fprintf(f, "   -\n      class_id: %s\n      modalities: [ ", kv.first.c_str());
    for (int m = 0; m < M; ++m) std::fprintf(f, "%s%s", m ? ", " : "", mod_name(bank->mods[m].type));
    std::fprintf(f, " ]\n      pyramid_levels: %d\n      template_pyramids:\n", L);
    for (int t = 0; t < cd.n_pyramids; ++t) {
      std::fprintf(f, "         -\n            template_id: %d\n            templates:\n", t);
      for (int k = 0; k < per; ++k) {
        const int32_t* tm = &cd.templates[((size_t)t * per + k) * 5];
        std::fprintf(f, "               -\n                  width: %d\n                  height: %d\n                  pyramid_level: %d\n                  features:\n",
                     tm[0], tm[1], tm[2]);
        for (int i = 0; i < tm[4]; ++i) {
          const int32_t* ft = &cd.features[((size_t)tm[3] + i) * 3];
          std::fprintf(f, "                     - [ %d, %d, %d ]\n", ft[0], ft[1], ft[2]);
        }
      }
    }
  }
  const bool ok = std::fflush(f) == 0 && !std::ferror(f);
  std::fclose(f);
  if (!ok) { set_error("write error on '%s'", path); return LMX_ERR_IO; }
  const std::string sidecar = std::string(path) + ".normal_lut";
  if (has_dn && !lut_is_default) {
    FILE* sf = std::fopen(sidecar.c_str(), "wb");
    if (!sf || std::fwrite(bank->normal_lut.data(), 1, LMX_NORMAL_LUT_SIZE, sf) != LMX_NORMAL_LUT_SIZE) {
      if (sf) std::fclose(sf);
      set_error("cannot write '%s'", sidecar.c_str());
      return LMX_ERR_IO;
    }
    std::fclose(sf);
  } else {
    std::remove(sidecar.c_str());  // a stale side-car of an earlier save would override the marker on load
  }
  return LMX_OK;
}

}  // namespace lmx

// ---- document tree over the C ABI (include/lmx.h: lmx_yaml_*) -----------------------------------------------------------------
// ---- the renderer-params side-car of a bank (what rcd_voting / the NMS read next to the matches) ----------------
// Written by the reference's trainers (writeLinemodTemplateParams, src/renderer.cpp:72-130 / src/renderer_only_image.cpp:57-101), read
// by readLinemodTemplateParams (src/rgbdDetector.cpp:1681-1749): "Template <i>": {ID, R (3x3 d), T (3x1 d), K (3x3 f), D, Ori_dist,
// Rect [x, y, w, h]} for i = 0, 1, ... until a key is missing, then the renderer_* scalars.
extern "C" lmx_status lmx_renderer_params_load(const char* path, lmx_renderer_params** out) {
  return lmx::guarded("lmx_renderer_params_load", [&]() -> lmx_status {
  if (!path || !out) { lmx::set_error("lmx_renderer_params_load: null argument"); return LMX_ERR_INVALID_ARG; }
  lmx::Node root;
  lmx_status st = lmx::parse_file(path, root);
  if (st != LMX_OK) return st;
  auto num = [&](const lmx::Node* n, double* v) {
    if (!n || n->kind != lmx::Node::Scalar || n->scalar.empty()) return false;
    char* end = nullptr;
    *v = std::strtod(n->scalar.c_str(), &end);
    return end != n->scalar.c_str();
  };
  auto mat = [&](const lmx::Node* n, size_t count, double* dst) {   // !!opencv-matrix -> `data`, row major
    const lmx::Node* d = n && n->kind == lmx::Node::Map ? n->get("data") : nullptr;
    if (!d || d->kind != lmx::Node::Seq || d->items.size() != count) return false;
    for (size_t i = 0; i < count; ++i)
      if (!num(&d->items[i], &dst[i])) return false;
    return true;
  };
  std::vector<double> dists, D, R, T, K;
  std::vector<int32_t> rects;
  size_t n = 0;
  for (;; ++n) {
    const std::string key = "Template " + std::to_string(n);
    const lmx::Node* t = root.get(key.c_str());
    if (!t) break;   // the reference stops at the first missing "Template <i>" too
    double ori = 0, dd = 0, r9[9], t3[3], k9[9];
    const lmx::Node* rc = t->kind == lmx::Node::Map ? t->get("Rect") : nullptr;
    if (t->kind != lmx::Node::Map || !num(t->get("Ori_dist"), &ori) || !num(t->get("D"), &dd) || !mat(t->get("R"), 9, r9) || !mat(t->get("T"), 3, t3) ||
        !mat(t->get("K"), 9, k9) || !rc || rc->kind != lmx::Node::Seq || rc->items.size() != 4) {
      lmx::set_error("'%s': %s is incomplete (R, T, K, D, Ori_dist, Rect expected)", path, key.c_str());
      return LMX_ERR_PARSE;
    }
    // the reference reads Ori_dist and D through a float (`float obj_dist_tmp; ... >> obj_dist_tmp; push_back(obj_dist_tmp)`)
    dists.push_back((double)(float)ori);
    D.push_back((double)(float)dd);
    R.insert(R.end(), r9, r9 + 9); T.insert(T.end(), t3, t3 + 3); K.insert(K.end(), k9, k9 + 9);
    for (int k = 0; k < 4; ++k) {
      double v = 0;
      if (!num(&rc->items[(size_t)k], &v) || !(v >= -2147483648.0 && v <= 2147483647.0)) { lmx::set_error("'%s': %s: bad Rect", path, key.c_str()); return LMX_ERR_PARSE; }
      rects.push_back((int32_t)v);
    }
  }
  // owned until every array has been allocated: a bad_alloc half-way (the hostile-file case guarded() exists for) must not leak the rest
  std::unique_ptr<lmx_renderer_params, void (*)(lmx_renderer_params*)> owner(new lmx_renderer_params(), lmx_renderer_params_free);
  lmx_renderer_params* p = owner.get();
  std::memset(p, 0, sizeof(*p));
  double v = 0;
  p->renderer_n_points = (num(root.get("renderer_n_points"), &v) && v >= -2147483648.0 && v <= 2147483647.0) ? (int32_t)v : 0;
  p->renderer_angle_step = (num(root.get("renderer_angle_step"), &v) && v >= -2147483648.0 && v <= 2147483647.0) ? (int32_t)v : 0;
  p->renderer_width = (num(root.get("renderer_width"), &v) && v >= -2147483648.0 && v <= 2147483647.0) ? (int32_t)v : 0;
  p->renderer_height = (num(root.get("renderer_height"), &v) && v >= -2147483648.0 && v <= 2147483647.0) ? (int32_t)v : 0;
  (void)num(root.get("renderer_radius_min"), &p->renderer_radius_min);
  (void)num(root.get("renderer_radius_max"), &p->renderer_radius_max);
  (void)num(root.get("renderer_radius_step"), &p->renderer_radius_step);
  (void)num(root.get("renderer_focal_length_x"), &p->renderer_focal_length_x);
  (void)num(root.get("renderer_focal_length_y"), &p->renderer_focal_length_y);
  (void)num(root.get("renderer_near"), &p->renderer_near);
  (void)num(root.get("renderer_far"), &p->renderer_far);
  p->n_templates = n;
  auto keep_d = [](const std::vector<double>& src) { double* q = new double[std::max<size_t>(src.size(), 1)]; std::copy(src.begin(), src.end(), q); return q; };
  p->obj_origin_dists = keep_d(dists); p->distances = keep_d(D); p->R = keep_d(R); p->T = keep_d(T); p->K = keep_d(K);
  p->rects = new int32_t[std::max<size_t>(rects.size(), 1)];
  std::copy(rects.begin(), rects.end(), p->rects);
  *out = owner.release();
  return LMX_OK;
  });
}

extern "C" void lmx_renderer_params_free(lmx_renderer_params* p) {
  if (!p) return;
  delete[] p->obj_origin_dists; delete[] p->distances; delete[] p->R; delete[] p->T; delete[] p->K; delete[] p->rects;
  delete p;
}

extern "C" lmx_status lmx_renderer_params_save(const lmx_renderer_params* p, const char* path) {
  return lmx::guarded("lmx_renderer_params_save", [&]() -> lmx_status {
  if (!p || !path) { lmx::set_error("lmx_renderer_params_save: null argument"); return LMX_ERR_INVALID_ARG; }
  FILE* f = std::fopen(path, "wb");
  if (!f) { lmx::set_error("cannot open '%s' for writing: %s", path, std::strerror(errno)); return LMX_ERR_IO; }
  std::fprintf(f, "%%YAML:1.0\n");
  auto matrix = [&](const char* name, int rows, int cols, char dt, const double* d) {
    std::fprintf(f, "   %s: !!opencv-matrix\n      rows: %d\n      cols: %d\n      dt: %c\n      data: [ ", name, rows, cols, dt);
    for (int i = 0; i < rows * cols; ++i) {
      if (dt == 'f') std::fprintf(f, "%.8e%s", d[i], i + 1 < rows * cols ? ", " : " ]\n");
      else std::fprintf(f, "%.16e%s", d[i], i + 1 < rows * cols ? ", " : " ]\n");
    }
  };
  for (size_t i = 0; i < p->n_templates; ++i) {
    std::fprintf(f, "Template %zu:\n   ID: %zu\n", i, i);
    matrix("R", 3, 3, 'd', p->R + 9 * i);
    matrix("T", 3, 1, 'd', p->T + 3 * i);
    matrix("K", 3, 3, 'f', p->K + 9 * i);
    std::fprintf(f, "   D: %.16e\n   Ori_dist: %.16e\n   Rect: [ %d, %d, %d, %d ]\n", p->distances[i], p->obj_origin_dists[i], p->rects[4 * i], p->rects[4 * i + 1],
                 p->rects[4 * i + 2], p->rects[4 * i + 3]);
  }
  std::fprintf(f, "renderer_n_points: %d\nrenderer_angle_step: %d\nrenderer_radius_min: %.16e\nrenderer_radius_max: %.16e\nrenderer_radius_step: %.16e\n", p->renderer_n_points,
               p->renderer_angle_step, p->renderer_radius_min, p->renderer_radius_max, p->renderer_radius_step);
  std::fprintf(f, "renderer_width: %d\nrenderer_height: %d\nrenderer_focal_length_x: %.16e\nrenderer_focal_length_y: %.16e\nrenderer_near: %.16e\nrenderer_far: %.16e\n",
               p->renderer_width, p->renderer_height, p->renderer_focal_length_x, p->renderer_focal_length_y, p->renderer_near, p->renderer_far);
  const bool ok = std::fflush(f) == 0;
  if (std::fclose(f) != 0 || !ok) { lmx::set_error("write error on '%s'", path); return LMX_ERR_IO; }
  return LMX_OK;
  });
}

struct lmx_yaml_doc { lmx::Node root; };
struct lmx_yaml_node;  // opaque alias of lmx::Node

namespace {
inline const lmx::Node* N(const lmx_yaml_node* n) { return reinterpret_cast<const lmx::Node*>(n); }
inline const lmx_yaml_node* H(const lmx::Node* n) { return reinterpret_cast<const lmx_yaml_node*>(n); }
}  // namespace

extern "C" {
lmx_status lmx_yaml_open(const char* path, lmx_yaml_doc** out) {
  if (!path || !out) { lmx::set_error("lmx_yaml_open: null argument"); return LMX_ERR_INVALID_ARG; }
  std::unique_ptr<lmx_yaml_doc> d(new lmx_yaml_doc());
  lmx_status st = lmx::parse_file(path, d->root);
  if (st != LMX_OK) return st;
  *out = d.release();
  return LMX_OK;
}
void lmx_yaml_close(lmx_yaml_doc* doc) { delete doc; }
const lmx_yaml_node* lmx_yaml_root(const lmx_yaml_doc* doc) { return doc ? H(&doc->root) : nullptr; }
int32_t lmx_yaml_kind(const lmx_yaml_node* n) { return n ? (int32_t)N(n)->kind : LMX_YAML_NULL; }
const char* lmx_yaml_scalar(const lmx_yaml_node* n) { return (n && N(n)->kind == lmx::Node::Scalar) ? N(n)->scalar.c_str() : ""; }
int32_t lmx_yaml_size(const lmx_yaml_node* n) {
  if (!n) return 0;
  return N(n)->kind == lmx::Node::Seq ? (int32_t)N(n)->items.size() : (N(n)->kind == lmx::Node::Map ? (int32_t)N(n)->entries.size() : 0);
}
const lmx_yaml_node* lmx_yaml_item(const lmx_yaml_node* n, int32_t i) {
  if (!n || i < 0) return nullptr;
  if (N(n)->kind == lmx::Node::Seq) return (size_t)i < N(n)->items.size() ? H(&N(n)->items[i]) : nullptr;
  if (N(n)->kind == lmx::Node::Map) return (size_t)i < N(n)->entries.size() ? H(&N(n)->entries[i].second) : nullptr;
  return nullptr;
}
const char* lmx_yaml_key(const lmx_yaml_node* n, int32_t i) {
  if (!n || i < 0 || N(n)->kind != lmx::Node::Map || (size_t)i >= N(n)->entries.size()) return nullptr;
  return N(n)->entries[i].first.c_str();
}
const lmx_yaml_node* lmx_yaml_get(const lmx_yaml_node* n, const char* key) {
  if (!n || !key || N(n)->kind != lmx::Node::Map) return nullptr;
  return H(N(n)->get(key));
}
}  // extern "C"
