// Multi-GPU matching from the C++ side: lmx_group_* of include/lmx.h.
// The caller of the hot path is C++ (rgbdDetector::linemod_detection, /root/reference/src/rgbdDetector.cpp:31-34), so sharding
// must not need Python: a group owns one device context per GPU (rank r holds templates [r*N/R, (r+1)*N/R) of every class,
// every rank pre-processes the same frames) and exchanges ONE RCCL all-gather of fixed-capacity per-rank blocks
// {64-byte header, lmx_raw_match_t[K]} per batch over xGMI; the host merges rank 0's copy with the same std::sort /
// std::unique a single GPU runs (lmx_merge_gathered), so the result equals the 1-GPU result for any R.
//   single process, all GPUs of the node : ncclCommInitAll over the chosen devices (the C++ node process)
//   one process per GPU                   : ncclCommInitRank with an id from lmx_group_unique_id (torchrun-style launchers)
// RCCL is loaded with dlopen when the first group is created: liblmx.so itself does not depend on it.
// Overflow of the gather block is not an error: the header carries every rank's record count, and when one exceeds the
// block's capacity the blocks are re-allocated to fit and the exchange is repeated from the records still held in the
// contexts' output slots (two-phase "counts first" form of SURVEY 8e, paid only when it is needed).

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "lmx_internal.hpp"

namespace {

// the slice of rccl.h this file uses (same ABI; the header is not required at build time of a caller)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint8 = 1 };

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) return nullptr;
  bool ok = true;
  auto sym = [&](const char* name) { void* p = dlsym(r.lib, name); if (!p) ok = false; return p; };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(r.lib); r.lib = nullptr; return nullptr; }
  return &r;
}

#define G_HIP(expr)                                                                                                     \
  do {                                                                                                                  \
    hipError_t e_ = (expr);                                                                                             \
    if (e_ != hipSuccess) { lmx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return LMX_ERR_HIP; } \
  } while (0)
#define G_NCCL(expr)                                                                                                    \
  do {                                                                                                                  \
    ncclResult_t r_ = (expr);                                                                                           \
    if (r_ != 0) { lmx::set_error("%s failed: %s", #expr, rccl()->GetErrorString(r_)); return LMX_ERR_HIP; }            \
  } while (0)

struct Member {          // one GPU of this process
  int device = 0;
  int rank = 0;
  lmx_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  uint8_t* d_send = nullptr;
  uint8_t* d_recv = nullptr;
};

}  // namespace

struct lmx_group {
  const lmx_bank* bank = nullptr;
  lmx_group_desc desc{};
  int world = 1;
  std::vector<Member> members;   // all ranks (single process) or this process's one rank
  size_t capacity = 0;           // records per rank block
  uint8_t* h_blocks = nullptr;   // pinned [world][block_bytes]: the merged view of rank `members[0]`
  size_t h_capacity = 0;
  size_t block_bytes() const { return LMX_GATHER_HEADER_BYTES + capacity * sizeof(lmx_raw_match_t); }
};

namespace {

lmx_status alloc_blocks(lmx_group* g, size_t capacity) {
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    if (m.d_send) (void)hipFree(m.d_send);
    if (m.d_recv) (void)hipFree(m.d_recv);
    m.d_send = m.d_recv = nullptr;
  }
  if (g->h_blocks) { (void)hipHostFree(g->h_blocks); g->h_blocks = nullptr; }
  g->capacity = capacity;
  const size_t bb = g->block_bytes();
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    G_HIP(hipMalloc((void**)&m.d_send, bb));
    G_HIP(hipMalloc((void**)&m.d_recv, bb * g->world));
    G_HIP(hipMemset(m.d_send, 0, bb));
  }
  G_HIP(hipSetDevice(g->members[0].device));
  G_HIP(hipHostMalloc((void**)&g->h_blocks, bb * g->world, hipHostMallocDefault));
  return LMX_OK;
}

// export every member's records of its most recent enqueue -> all-gather -> rank members[0]'s gathered blocks in pinned memory
lmx_status exchange(lmx_group* g) {
  Rccl* R = rccl();
  const size_t bb = g->block_bytes();
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    lmx_status st = lmx_ctx_export_raw_on(m.ctx, m.d_send, g->capacity, m.comm_stream);
    if (st != LMX_OK) return st;
  }
  G_NCCL(R->GroupStart());
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    G_NCCL(R->AllGather(m.d_send, m.d_recv, bb, ncclUint8, m.comm, m.comm_stream));
  }
  G_NCCL(R->GroupEnd());
  Member& m0 = g->members[0];
  G_HIP(hipSetDevice(m0.device));
  lmx_status st = lmx_stream_copy_blocks(g->h_blocks, m0.d_recv, g->world, bb, g->capacity, m0.comm_stream);
  if (st != LMX_OK) return st;
  for (Member& m : g->members) {
    G_HIP(hipSetDevice(m.device));
    G_HIP(hipStreamSynchronize(m.comm_stream));
  }
  return LMX_OK;
}

}  // namespace

extern "C" {

lmx_status lmx_group_unique_id(void* out128) {
  if (!out128) { lmx::set_error("lmx_group_unique_id: null argument"); return LMX_ERR_INVALID_ARG; }
  Rccl* R = rccl();
  if (!R) { lmx::set_error("librccl.so could not be loaded: %s", dlerror() ? dlerror() : "symbols missing"); return LMX_ERR_NOT_FOUND; }
  ncclUniqueId id;
  G_NCCL(R->GetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
  return LMX_OK;
}

void lmx_group_destroy(lmx_group* g) {
  if (!g) return;
  Rccl* R = rccl();
  for (Member& m : g->members) {
    (void)hipSetDevice(m.device);
    if (m.comm_stream) (void)hipStreamSynchronize(m.comm_stream);
    if (m.comm && R) (void)R->CommDestroy(m.comm);
    if (m.ctx) lmx_ctx_destroy(m.ctx);
    if (m.d_send) (void)hipFree(m.d_send);
    if (m.d_recv) (void)hipFree(m.d_recv);
    if (m.comm_stream) (void)hipStreamDestroy(m.comm_stream);
  }
  if (g->h_blocks) (void)hipHostFree(g->h_blocks);
  delete g;
}

lmx_status lmx_group_create(const lmx_bank* bank, const lmx_group_desc* desc, lmx_group** out) {
  if (!bank || !desc || !out) { lmx::set_error("lmx_group_create: null argument"); return LMX_ERR_INVALID_ARG; }
  const bool multi_process = desc->unique_id != nullptr;
  if (multi_process && (desc->world < 1 || desc->rank < 0 || desc->rank >= desc->world)) { lmx::set_error("lmx_group_create: rank %d outside world %d", desc->rank, desc->world); return LMX_ERR_INVALID_ARG; }
  if (!multi_process && desc->n_devices < 1) { lmx::set_error("lmx_group_create: n_devices must be >= 1"); return LMX_ERR_INVALID_ARG; }
  if (desc->max_batch < 1) { lmx::set_error("lmx_group_create: max_batch must be >= 1"); return LMX_ERR_INVALID_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { lmx::set_error("no HIP device available; this library has no CPU path"); return LMX_ERR_NO_DEVICE; }
  Rccl* R = rccl();
  if (!R) { lmx::set_error("librccl.so could not be loaded (needed for the all-gather of a device group)"); return LMX_ERR_NOT_FOUND; }
  lmx_group* g = new lmx_group();
  g->bank = bank; g->desc = *desc;
  g->world = multi_process ? desc->world : desc->n_devices;
  const int n_local = multi_process ? 1 : desc->n_devices;
  g->members.resize((size_t)n_local);
  std::vector<int> devlist((size_t)n_local);
  for (int i = 0; i < n_local; ++i) {
    Member& m = g->members[(size_t)i];
    m.device = multi_process ? desc->device : (desc->devices ? desc->devices[i] : i);
    m.rank = multi_process ? desc->rank : i;
    devlist[(size_t)i] = m.device;
    if (m.device < 0 || m.device >= ndev) { lmx::set_error("device %d out of range (%d devices)", m.device, ndev); lmx_group_destroy(g); return LMX_ERR_NO_DEVICE; }
  }
  auto fail = [&](lmx_status st) { std::string keep = lmx_last_error(); lmx_group_destroy(g); lmx::set_error("%s", keep.c_str()); return st; };
  for (Member& m : g->members) {
    if (hipSetDevice(m.device) != hipSuccess) return fail(LMX_ERR_HIP);
    lmx_ctx_desc cd;
    std::memset(&cd, 0, sizeof(cd));
    cd.device = m.device; cd.width = desc->width; cd.height = desc->height; cd.max_batch = desc->max_batch; cd.max_candidates = desc->max_candidates;
    cd.shard_rank = m.rank; cd.shard_world = g->world; cd.flags = desc->flags;
    lmx_status st = lmx_ctx_create(bank, &cd, &m.ctx);
    if (st != LMX_OK) return fail(st);
    if (hipStreamCreateWithFlags(&m.comm_stream, hipStreamNonBlocking) != hipSuccess) { lmx::set_error("hipStreamCreate failed"); return fail(LMX_ERR_HIP); }
  }
  if (multi_process) {
    ncclUniqueId id;
    std::memcpy(&id, desc->unique_id, sizeof(id));
    (void)hipSetDevice(g->members[0].device);
    ncclResult_t r = R->CommInitRank(&g->members[0].comm, g->world, id, desc->rank);
    if (r != 0) { lmx::set_error("ncclCommInitRank failed: %s", R->GetErrorString(r)); return fail(LMX_ERR_HIP); }
  } else {
    std::vector<ncclComm_t> comms((size_t)n_local);
    ncclResult_t r = R->CommInitAll(comms.data(), n_local, devlist.data());
    if (r != 0) { lmx::set_error("ncclCommInitAll failed: %s", R->GetErrorString(r)); return fail(LMX_ERR_HIP); }
    for (int i = 0; i < n_local; ++i) g->members[(size_t)i].comm = comms[(size_t)i];
  }
  lmx_status st = alloc_blocks(g, desc->gather_capacity > 0 ? (size_t)desc->gather_capacity : 8192);
  if (st != LMX_OK) return fail(st);
  *out = g;
  return LMX_OK;
}

int32_t lmx_group_size(const lmx_group* g) { return g ? g->world : 0; }
int32_t lmx_group_gather_capacity(const lmx_group* g) { return g ? (int32_t)g->capacity : 0; }

lmx_status lmx_group_match_batch(lmx_group* g, int32_t n_frames, const lmx_image* sources, int32_t n_sources, float threshold,
                                 const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out) {
  if (!g || !sources || !n_out || (cap > 0 && !out)) { lmx::set_error("lmx_group_match_batch: null argument"); return LMX_ERR_INVALID_ARG; }
  // every rank sees the same frames (pre-processing is replicated: cheaper than moving linear memories over xGMI)
  for (Member& m : g->members) {
    lmx_status st = lmx_ctx_upload(m.ctx, n_frames, sources, n_sources);
    if (st == LMX_OK) st = lmx_ctx_enqueue(m.ctx, n_frames, threshold, class_ids, n_class_ids);
    if (st != LMX_OK) return st;
  }
  lmx_status st = exchange(g);
  // two-phase fallback: the headers say how many records every rank really has
  if (st == LMX_OK) {
    size_t need = 0;
    for (int r = 0; r < g->world; ++r) need = std::max<size_t>(need, reinterpret_cast<const uint32_t*>(g->h_blocks + (size_t)r * g->block_bytes())[1]);
    if (need > g->capacity) {
      size_t grown = g->capacity;
      while (grown < need) grown *= 2;
      st = alloc_blocks(g, grown);
      if (st == LMX_OK) st = exchange(g);   // the records are still in the contexts' output slots
    }
  }
  std::vector<size_t> offsets((size_t)n_frames + 1, 0);
  std::vector<lmx_match_t> flat;
  if (st == LMX_OK) {
    flat.resize(std::max<size_t>(1, cap * (size_t)n_frames));
    st = lmx_merge_gathered(g->h_blocks, g->world, g->block_bytes(), g->capacity, n_frames, flat.data(), flat.size(), offsets.data());
    if (st == LMX_ERR_OVERFLOW && offsets[(size_t)n_frames] > flat.size()) {   // more matches than cap * n_frames in total: size exactly, report per frame below
      flat.resize(offsets[(size_t)n_frames]);
      st = lmx_merge_gathered(g->h_blocks, g->world, g->block_bytes(), g->capacity, n_frames, flat.data(), flat.size(), offsets.data());
    }
  }
  for (Member& m : g->members) {   // frees the output slot whether or not the exchange worked
    lmx_status rs = lmx_ctx_release(m.ctx);
    if (st == LMX_OK && rs != LMX_OK) st = rs;
  }
  if (st != LMX_OK) { for (int f = 0; f < n_frames; ++f) n_out[f] = 0; return st; }
  for (int f = 0; f < n_frames; ++f) {
    const size_t n = offsets[(size_t)f + 1] - offsets[(size_t)f];
    n_out[f] = n;
    std::memcpy(out + (size_t)f * cap, flat.data() + offsets[(size_t)f], std::min(n, cap) * sizeof(lmx_match_t));
    if (n > cap) { lmx::set_error("frame %d: %zu matches > output capacity %zu", f, n, cap); st = LMX_ERR_OVERFLOW; }
  }
  return st;
}

}  // extern "C"
