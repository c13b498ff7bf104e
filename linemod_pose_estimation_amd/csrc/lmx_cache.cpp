// liblmx.so, what makes the reference's per-request detector rebuild cheap (..._service.cpp:1784-1786; include/lmx.h): bank
// fingerprint, the compact binary bank file, the process-wide bank cache behind lmx_bank_load_yaml_cached and the device-context
// cache behind lmx_ctx_acquire.

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

// ---- caches for the per-request detector rebuild of the reference's service node (include/lmx.h) -----------------------------
namespace {

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

// Eight bytes per step for the bank fingerprint (an in-process cache key, not a file format: the binary bank's checksum stays the
// byte-wise FNV-1a above).  A freshly built bank of 3000 RGB-D templates (7 MB of templates and features) hashes in ~0.7 instead of 5.5 ms,
// which is what a detector rebuilt from FileNodes on every request pays before lmx_ctx_acquire can find its context.
uint64_t hash_words(uint64_t h, const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    std::memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x9e3779b97f4a7c15ull;
    h ^= h >> 29;
  }
  return fnv1a(h, p + i, n - i);
}

struct BankCacheEntry { std::string path; long long mtime_ns; long long size; uint64_t lut_key; lmx_bank* bank; int refs; };
struct CtxCacheEntry { uint64_t fingerprint; lmx_ctx_desc desc; lmx_bank* bank; lmx_ctx* ctx; int refs; uint64_t last_use; };

std::mutex g_cache_mutex;
std::vector<BankCacheEntry> g_bank_cache;
std::vector<CtxCacheEntry> g_ctx_cache;
uint64_t g_cache_clock = 0;
constexpr size_t kMaxIdleContexts = 8;

bool same_desc(const lmx_ctx_desc& a, const lmx_ctx_desc& b) {
  return a.device == b.device && a.width == b.width && a.height == b.height && a.max_batch == b.max_batch && a.max_candidates == b.max_candidates &&
         a.shard_rank == b.shard_rank && a.shard_world == b.shard_world && a.stream == b.stream && a.flags == b.flags;
}

}  // namespace

extern "C" {

lmx_status lmx_bank_clone(const lmx_bank* bank, lmx_bank** out) {
  return lmx::guarded("lmx_bank_clone", [&]() -> lmx_status {
  if (!bank || !out) { set_error("lmx_bank_clone: null argument"); return LMX_ERR_INVALID_ARG; }
  *out = new lmx_bank(*bank);
  return LMX_OK;
  });
}

uint64_t lmx_bank_fingerprint(const lmx_bank* bank) {
  if (!bank) return 0;
  // what the remembered value covered: counts of everything hashed below + the table's epoch (never 0, so that an empty cache misses)
  uint64_t sig = 0x9e3779b97f4a7c15ull ^ bank->T.size() ^ (bank->mods.size() << 8) ^ ((uint64_t)bank->lut_epoch << 16) ^ ((uint64_t)bank->normal_lut.size() << 40) ^
                 ((uint64_t)(uint32_t)bank->normal_lut_origin << 56);
  for (const auto& kv : bank->classes)
    sig = (sig * 0x100000001b3ull) ^ (kv.second.templates.size() * 0x9e3779b1ull) ^ (kv.second.features.size() << 20) ^ kv.first.size();
  sig |= 1ull;
  if (bank->fp_cache.signature.load(std::memory_order_acquire) == sig) return bank->fp_cache.value.load(std::memory_order_relaxed);
  uint64_t h = 0xcbf29ce484222325ull;
  h = fnv1a(h, bank->T.data(), bank->T.size() * sizeof(int32_t));
  for (const lmx_modality_desc& m : bank->mods) {
    // field by field: the struct has no padding today, but the hash must not depend on that
    h = fnv1a(h, &m.type, sizeof(m.type)); h = fnv1a(h, &m.weak_threshold, sizeof(float)); h = fnv1a(h, &m.strong_threshold, sizeof(float));
    h = fnv1a(h, &m.num_features, 4); h = fnv1a(h, &m.distance_threshold, 4); h = fnv1a(h, &m.difference_threshold, 4); h = fnv1a(h, &m.extract_threshold, 4);
  }
  h = hash_words(h, bank->normal_lut.data(), bank->normal_lut.size());
  for (const auto& kv : bank->classes) {
    h = fnv1a(h, kv.first.data(), kv.first.size() + 1);
    h = fnv1a(h, &kv.second.n_pyramids, 4);
    h = hash_words(h, kv.second.templates.data(), kv.second.templates.size() * sizeof(int32_t));
    h = hash_words(h, kv.second.features.data(), kv.second.features.size() * sizeof(int32_t));
  }
  bank->fp_cache.value.store(h, std::memory_order_relaxed);
  bank->fp_cache.signature.store(sig, std::memory_order_release);
  return h;
}

// ---- compact binary form of a bank ("+ a compact binary cache") ---------------------------------------------
// Layout (little endian): "LMXBANK1", then u64 fields and raw int32 arrays as written below, then the FNV-1a hash of everything
// before it.  A 3000-template RGB-D bank is 22.7 MB of FileStorage YAML (0.5 s to parse here, seconds in OpenCV) and 7 MB /
// 14 ms in this form.
namespace {
struct Writer {
  std::vector<uint8_t> buf;
  void raw(const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); buf.insert(buf.end(), b, b + n); }
  void u64(uint64_t v) { raw(&v, 8); }
};
struct Reader {
  const uint8_t* p; size_t n, pos = 0; bool ok = true;
  bool raw(void* dst, size_t k) { if (!ok || pos + k > n) { ok = false; return false; } std::memcpy(dst, p + pos, k); pos += k; return true; }
  uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
};

void serialize_bank(const lmx_bank* b, Writer& w) {
  w.raw("LMXBANK1", 8);
  w.u64(b->T.size()); w.raw(b->T.data(), b->T.size() * 4);
  w.u64(b->mods.size());
  for (const lmx_modality_desc& m : b->mods) {
    const int32_t ints[5] = {m.type, m.num_features, m.distance_threshold, m.difference_threshold, m.extract_threshold};
    w.raw(ints, sizeof(ints)); w.raw(&m.weak_threshold, 4); w.raw(&m.strong_threshold, 4);
  }
  w.u64((uint64_t)b->normal_lut_origin); w.raw(b->normal_lut.data(), LMX_NORMAL_LUT_SIZE);
  w.u64(b->classes.size());
  for (const auto& kv : b->classes) {
    w.u64(kv.first.size()); w.raw(kv.first.data(), kv.first.size());
    w.u64((uint64_t)kv.second.n_pyramids);
    w.u64(kv.second.templates.size()); w.raw(kv.second.templates.data(), kv.second.templates.size() * 4);
    w.u64(kv.second.features.size()); w.raw(kv.second.features.data(), kv.second.features.size() * 4);
  }
  w.u64(fnv1a(0xcbf29ce484222325ull, w.buf.data(), w.buf.size()));
}

lmx_status deserialize_bank(const uint8_t* data, size_t n, lmx_bank** out, const char* what) {
  if (n < 16 || std::memcmp(data, "LMXBANK1", 8) != 0) { set_error("'%s' is not a liblmx binary bank", what); return LMX_ERR_PARSE; }
  uint64_t stored = 0;
  std::memcpy(&stored, data + n - 8, 8);
  if (stored != fnv1a(0xcbf29ce484222325ull, data, n - 8)) { set_error("'%s': checksum mismatch (truncated or corrupted)", what); return LMX_ERR_PARSE; }
  Reader r{data, n - 8};
  r.pos = 8;
  std::unique_ptr<lmx_bank> b(new lmx_bank());
  const uint64_t L = r.u64();
  if (!r.ok || L < 1 || L > (uint64_t)kMaxLevels) { set_error("'%s': bad header", what); return LMX_ERR_PARSE; }
  b->T.resize(L); r.raw(b->T.data(), L * 4);
  for (uint64_t l = 0; r.ok && l < L; ++l)
    if (b->T[l] < 1 || b->T[l] > 16) { set_error("'%s': T=%d at level %d outside 1..16", what, b->T[l], (int)l); return LMX_ERR_PARSE; }
  const uint64_t M = r.u64();
  if (!r.ok || M < 1 || M > (uint64_t)kMaxModalities) { set_error("'%s': bad header", what); return LMX_ERR_PARSE; }
  for (uint64_t m = 0; m < M; ++m) {
    int32_t ints[5];
    lmx_modality_desc d{};
    r.raw(ints, sizeof(ints)); r.raw(&d.weak_threshold, 4); r.raw(&d.strong_threshold, 4);
    d.type = ints[0]; d.num_features = ints[1]; d.distance_threshold = ints[2]; d.difference_threshold = ints[3]; d.extract_threshold = ints[4];
    if (r.ok && d.type != LMX_MOD_COLOR_GRADIENT && d.type != LMX_MOD_DEPTH_NORMAL) { set_error("'%s': unknown modality type %d", what, d.type); return LMX_ERR_PARSE; }
    b->mods.push_back(d);
  }
  b->normal_lut_origin = (int32_t)r.u64();
  b->normal_lut.resize(LMX_NORMAL_LUT_SIZE); r.raw(b->normal_lut.data(), LMX_NORMAL_LUT_SIZE);
  if (r.ok) {
    std::vector<uint8_t> bins(LMX_NORMAL_LUT_SIZE);
    if (b->normal_lut_origin < LMX_LUT_DEFAULT || b->normal_lut_origin > LMX_LUT_UNKNOWN) { set_error("'%s': bad normal-LUT origin %d", what, b->normal_lut_origin); return LMX_ERR_PARSE; }
    if (!normal_lut_to_bins(b->normal_lut.data(), bins.data())) { set_error("'%s': normal LUT entries must be 0 or a single bit", what); return LMX_ERR_PARSE; }
  }
  const uint64_t nc = r.u64();
  for (uint64_t c = 0; r.ok && c < nc; ++c) {
    const uint64_t len = r.u64();
    if (!r.ok || len > 4096) { r.ok = false; break; }
    std::string name(len, '\0');
    r.raw(&name[0], len);
    const int64_t n_pyr = (int64_t)r.u64();
    const uint64_t nt = r.u64();
    if (!r.ok || nt > (n / 4) || n_pyr < 0 || n_pyr > (int64_t)(n / 20)) { r.ok = false; break; }
    std::vector<int32_t> templates(nt);
    r.raw(templates.data(), nt * 4);
    const uint64_t nf = r.u64();
    if (!r.ok || nf > (n / 4) || nf % 3 != 0) { r.ok = false; break; }
    std::vector<int32_t> features(nf);
    r.raw(features.data(), nf * 4);
    if (!r.ok || templates.size() != (size_t)n_pyr * L * M * 5 || b->classes.count(name)) { r.ok = false; break; }
    // the same validation every other way into a bank goes through (feature counts <= 63, ranges inside `features`, pyramid
    // levels, coordinates, labels 0..7): a stale-format, damaged-but-rehashed or crafted file must not reach build_device_bank
    static const int32_t none[5] = {0, 0, 0, 0, 0};
    const lmx_status vs = lmx_bank_add_class(b.get(), name.c_str(), (int32_t)n_pyr, templates.empty() ? none : templates.data(), features.empty() ? none : features.data(),
                                             (int64_t)(nf / 3));
    if (vs != LMX_OK) { const std::string why = lmx_last_error(); set_error("'%s': class '%s' is invalid: %s", what, name.c_str(), why.c_str()); return LMX_ERR_PARSE; }
  }
  if (!r.ok || r.pos != n - 8) { set_error("'%s': malformed binary bank", what); return LMX_ERR_PARSE; }
  *out = b.release();
  return LMX_OK;
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  uint8_t tmp[1 << 16];
  size_t k;
  while ((k = std::fread(tmp, 1, sizeof(tmp), f)) > 0) out.insert(out.end(), tmp, tmp + k);
  std::fclose(f);
  return true;
}
}  // namespace

lmx_status lmx_bank_save_binary(const lmx_bank* bank, const char* path) {
  return lmx::guarded("lmx_bank_save_binary", [&]() -> lmx_status {
  if (!bank || !path) { set_error("lmx_bank_save_binary: null argument"); return LMX_ERR_INVALID_ARG; }
  Writer w;
  serialize_bank(bank, w);
  FILE* f = std::fopen(path, "wb");
  if (!f) { set_error("cannot open '%s' for writing", path); return LMX_ERR_IO; }
  const bool ok = std::fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
  if (std::fclose(f) != 0 || !ok) { set_error("write error on '%s'", path); return LMX_ERR_IO; }
  return LMX_OK;
  });
}

lmx_status lmx_bank_load_binary(const char* path, lmx_bank** out) {
  return lmx::guarded("lmx_bank_load_binary", [&]() -> lmx_status {
  if (!path || !out) { set_error("lmx_bank_load_binary: null argument"); return LMX_ERR_INVALID_ARG; }
  std::vector<uint8_t> data;
  if (!read_file(path, data)) { set_error("cannot open '%s'", path); return LMX_ERR_IO; }
  return deserialize_bank(data.data(), data.size(), out, path);
  });
}

// Everything outside the yml that yaml_load folds into the bank: the side-car table `<yml>.normal_lut` and the file the environment
// variable LMX_NORMAL_LUT names (existence, mtime, size, and the variable's value).  Part of both cache keys: a table that appears
// or changes later must not be masked by a bank cached without it (advisor finding, round 2).
static uint64_t lut_inputs_key(const char* yml_path) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix_file = [&](const char* p) {
    struct stat sb;
    long long v[3] = {0, 0, 0};
    if (stat(p, &sb) == 0) { v[0] = 1; v[1] = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec; v[2] = (long long)sb.st_size; }
    h = fnv1a(h, v, sizeof(v));
  };
  mix_file((std::string(yml_path) + ".normal_lut").c_str());
  const char* env = std::getenv("LMX_NORMAL_LUT");
  if (env && *env) { h = fnv1a(h, env, std::strlen(env) + 1); mix_file(env); }
  return h;
}

lmx_status lmx_bank_load_yaml_cached(const char* path, const lmx_bank** out) {
  return lmx::guarded("lmx_bank_load_yaml_cached", [&]() -> lmx_status {
  if (!path || !out) { set_error("lmx_bank_load_yaml_cached: null argument"); return LMX_ERR_INVALID_ARG; }
  struct stat sb;
  if (stat(path, &sb) != 0) { set_error("cannot open '%s'", path); return LMX_ERR_IO; }
  const long long mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec, size = (long long)sb.st_size;
  const uint64_t lut_key = lut_inputs_key(path);
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (size_t i = 0; i < g_bank_cache.size(); ++i) {
    BankCacheEntry& e = g_bank_cache[i];
    if (e.path != path) continue;
    if (e.mtime_ns == mtime_ns && e.size == size && e.lut_key == lut_key) { e.refs += 1; *out = e.bank; return LMX_OK; }
    if (e.refs == 0) { delete e.bank; g_bank_cache.erase(g_bank_cache.begin() + (long)i); --i; }  // stale and unused
    // a stale entry that is still referenced stays until released; the new version gets its own entry
  }
  // second level: "<path>.lmxcache" next to the yml = {"LMXCACH2", mtime and size of the yml it was made from, key of the table inputs,
  // binary bank}; written on a miss when the directory allows it, ignored when stale, of another format or unreadable -- the yml is then
  // parsed again (LMX_NO_DISK_CACHE=1 turns it off)
  lmx_bank* b = nullptr;
  static const bool disk = std::getenv("LMX_NO_DISK_CACHE") == nullptr;
  const std::string cache_path = std::string(path) + ".lmxcache";
  if (disk) {
    std::vector<uint8_t> data;
    if (read_file(cache_path.c_str(), data) && data.size() > 32 && std::memcmp(data.data(), "LMXCACH2", 8) == 0) {
      long long c_mtime = 0, c_size = 0;
      uint64_t c_lut = 0;
      std::memcpy(&c_mtime, data.data() + 8, 8); std::memcpy(&c_size, data.data() + 16, 8); std::memcpy(&c_lut, data.data() + 24, 8);
      if (c_mtime == mtime_ns && c_size == size && c_lut == lut_key && deserialize_bank(data.data() + 32, data.size() - 32, &b, cache_path.c_str()) != LMX_OK) b = nullptr;
    }
  }
  if (!b) {
    lmx_status st = yaml_load(path, &b);
    if (st != LMX_OK) return st;
    if (disk) {
      Writer w;
      w.raw("LMXCACH2", 8); w.raw(&mtime_ns, 8); w.raw(&size, 8); w.raw(&lut_key, 8);
      Writer body;
      serialize_bank(b, body);
      w.raw(body.buf.data(), body.buf.size());
      const std::string tmp = cache_path + ".tmp";
      FILE* f = std::fopen(tmp.c_str(), "wb");
      if (f) {
        const bool ok = std::fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
        if (std::fclose(f) == 0 && ok) (void)std::rename(tmp.c_str(), cache_path.c_str());
        else (void)std::remove(tmp.c_str());
      }
    }
  }
  g_bank_cache.push_back(BankCacheEntry{path, mtime_ns, size, lut_key, b, 1});
  *out = b;
  return LMX_OK;
  });
}

void lmx_bank_release(const lmx_bank* bank) {
  if (!bank) return;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (BankCacheEntry& e : g_bank_cache)
    if (e.bank == bank && e.refs > 0) { e.refs -= 1; return; }   // stays cached for the next request
}

lmx_status lmx_ctx_acquire(const lmx_bank* bank, const lmx_ctx_desc* desc, lmx_ctx** out, int32_t* cache_hit) {
  return lmx::guarded("lmx_ctx_acquire", [&]() -> lmx_status {
  if (!bank || !desc || !out) { set_error("lmx_ctx_acquire: null argument"); return LMX_ERR_INVALID_ARG; }
  const uint64_t fp = lmx_bank_fingerprint(bank);
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (CtxCacheEntry& e : g_ctx_cache)
    if (e.fingerprint == fp && same_desc(e.desc, *desc) && e.bank->normal_lut_origin == bank->normal_lut_origin) {
      e.refs += 1; e.last_use = ++g_cache_clock;
      *out = e.ctx;
      if (cache_hit) *cache_hit = 1;
      return LMX_OK;
    }
  lmx_bank* own = new lmx_bank(*bank);
  lmx_ctx* ctx = nullptr;
  lmx_status st = lmx_ctx_create(own, desc, &ctx);
  if (st != LMX_OK) { delete own; return st; }
  // evict idle contexts beyond the limit, least recently used first
  for (;;) {
    size_t idle = 0, victim = g_ctx_cache.size();
    for (size_t i = 0; i < g_ctx_cache.size(); ++i)
      if (g_ctx_cache[i].refs == 0) { ++idle; if (victim == g_ctx_cache.size() || g_ctx_cache[i].last_use < g_ctx_cache[victim].last_use) victim = i; }
    if (idle < kMaxIdleContexts) break;
    lmx_ctx_destroy(g_ctx_cache[victim].ctx);
    delete g_ctx_cache[victim].bank;
    g_ctx_cache.erase(g_ctx_cache.begin() + (long)victim);
  }
  g_ctx_cache.push_back(CtxCacheEntry{fp, *desc, own, ctx, 1, ++g_cache_clock});
  *out = ctx;
  if (cache_hit) *cache_hit = 0;
  return LMX_OK;
  });
}

void lmx_ctx_unref(lmx_ctx* ctx) {
  if (!ctx) return;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (CtxCacheEntry& e : g_ctx_cache)
    if (e.ctx == ctx && e.refs > 0) { e.refs -= 1; e.last_use = ++g_cache_clock; return; }
}

void lmx_cache_trim(void) {
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  for (size_t i = 0; i < g_ctx_cache.size();) {
    if (g_ctx_cache[i].refs == 0) {
      lmx_ctx_destroy(g_ctx_cache[i].ctx);
      delete g_ctx_cache[i].bank;
      g_ctx_cache.erase(g_ctx_cache.begin() + (long)i);
    } else {
      ++i;
    }
  }
  for (size_t i = 0; i < g_bank_cache.size();) {
    if (g_bank_cache[i].refs == 0) { delete g_bank_cache[i].bank; g_bank_cache.erase(g_bank_cache.begin() + (long)i); }
    else ++i;
  }
}

}  // extern "C"
