/* include/lmx.h -- C ABI of the MI355X-native LINEMOD template-matching engine (liblmx.so).
 *
 * Drop-in boundary.  Every entry point names the reference interface it replaces; paths are
 * relative to the reference repository (birlrobotics/linemod_pose_estimation):
 *
 *   lmx_match / lmx_match_batch   <-  cv::linemod::Detector::match(sources, threshold, matches, class_ids,
 *                                     noArray()) as called by rgbdDetector::linemod_detection,
 *                                     src/rgbdDetector.cpp:31-34 (decl include/linemod_pose_estimation/rgbdDetector.h:150);
 *                                     callers src/linemod_ensenso_detect_3_mult_detect_service.cpp:344,1190,
 *                                     src/linemod_ensenso_detect_3_mult_detect.cpp:321,1167, src/linemod_carmine_detect.cpp:348.
 *   lmx_bank_create / _add_class  <-  cv::linemod::Detector(modalities, T) + addTemplate results,
 *                                     src/renderer.cpp:179-185,308.
 *   lmx_bank_load_yaml            <-  readLinemod: Detector::read(fs.root()) + readClass per classes[] entry,
 *                                     src/rgbdDetector.cpp:1668-1680 (copies: src/renderer.cpp:42-54, ..._service.cpp:708-721).
 *   lmx_bank_save_yaml            <-  writeLinemod: Detector::write + writeClass, src/renderer.cpp:56-70.
 *   lmx_bank_num_classes/_class_id<-  Detector::classIds(), src/linemod_ensenso_detect_3_mult_detect.cpp:290.
 *   lmx_bank_get_template         <-  Detector::getTemplates(class_id, template_id),
 *                                     src/linemod_ensenso_detect_3_mult_detect_service.cpp:351.
 *   lmx_bank_num_templates, lmx_bank_T, lmx_bank_pyramid_levels <- Detector::numTemplates / getT / pyramidLevels.
 *
 * Plain C types only: pointers, sizes, fixed-width integers.  No C++/torch types cross this boundary.
 * All functions return lmx_status; lmx_last_error() gives the message of the calling thread's last failure.
 * Misuse that upstream signals with CV_Assert (source count != modality count, image size not a multiple
 * of T at some level, more than 63 features in a template) returns LMX_ERR_SHAPE; the C++ facade
 * (include/lmx_linemod.hpp) rethrows it as an exception like cv::Exception.
 *
 * There is NO CPU fallback: every compute entry point fails with LMX_ERR_NO_DEVICE when no gfx950 device
 * is usable.
 */
#ifndef LMX_H_
#define LMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lmx_status {
  LMX_OK = 0,
  LMX_ERR_INVALID_ARG = 1,
  LMX_ERR_SHAPE = 2,      /* upstream CV_Assert equivalents */
  LMX_ERR_NO_DEVICE = 3,
  LMX_ERR_HIP = 4,
  LMX_ERR_OVERFLOW = 5,   /* candidate / match capacity exceeded; raise lmx_ctx_desc.max_candidates */
  LMX_ERR_IO = 6,
  LMX_ERR_PARSE = 7,
  LMX_ERR_NOT_FOUND = 8
} lmx_status;

enum { LMX_MOD_COLOR_GRADIENT = 0, LMX_MOD_DEPTH_NORMAL = 1 };

typedef struct lmx_bank lmx_bank; /* host-side template bank == the template state of cv::linemod::Detector */
typedef struct lmx_ctx lmx_ctx;   /* device context: a bank (or a shard of it) resident in HBM + per-frame workspaces */

/* Modality parameters (upstream defaults: ColorGradient(10, 63, 55), DepthNormal(2000, 50, 63, 2)). */
typedef struct lmx_modality_desc {
  int32_t type; /* LMX_MOD_* */
  float weak_threshold;
  float strong_threshold;
  int32_t num_features;
  int32_t distance_threshold;
  int32_t difference_threshold;
  int32_t extract_threshold;
} lmx_modality_desc;

typedef struct lmx_bank_desc {
  int32_t pyramid_levels;
  const int32_t* T; /* [pyramid_levels], e.g. {5, 8} (reference src/renderer.cpp:182-185) */
  int32_t n_modalities;
  const lmx_modality_desc* modalities;
} lmx_bank_desc;

/* A source image as the reference hands it to match(): possibly a strided ROI view
 * (src/linemod_ensenso_detect_3_mult_detect_service.cpp:324-326).  ColorGradient wants 8UC3 (channels 3,
 * elem_size 1), DepthNormal 16UC1 depth in mm (channels 1, elem_size 2). */
typedef struct lmx_image {
  const void* data;
  int32_t rows, cols;
  int32_t channels;
  int32_t elem_size;
  size_t row_stride_bytes;
} lmx_image;

/* cv::linemod::Match with class_id replaced by its index in lmx_bank_class_id() order (std::map order). */
typedef struct lmx_match_t {
  int32_t x, y;
  float similarity;
  int32_t template_id;
  int32_t class_index;
} lmx_match_t;

/* Unsorted shard-local match record (what ranks exchange in the multi-GPU all-gather):
 * `order_key` restores upstream insertion order (class slot, template_id, coarse raster index). */
typedef struct lmx_raw_match_t {
  int32_t x, y;
  float similarity;
  int32_t template_id;
  int32_t class_index;
  int32_t frame;
  uint64_t order_key;
} lmx_raw_match_t;

typedef struct lmx_ctx_desc {
  int32_t device;         /* HIP device ordinal */
  int32_t width, height;  /* frame size at pyramid level 0; must satisfy the T divisibility of every level */
  int32_t max_batch;      /* frames resident at once (>= 1) */
  int32_t max_candidates; /* per-frame capacity of coarse candidates and of matches; 0 = default (65536) */
  int32_t shard_rank;     /* this context holds templates [rank*N/world, (rank+1)*N/world) of every class */
  int32_t shard_world;    /* 0 or 1 = whole bank */
  void* stream;           /* hipStream_t to run on, or NULL to create a private one */
  int32_t flags;          /* LMX_CTX_* */
} lmx_ctx_desc;
/* Capture the per-batch kernel chain of enqueue() into a hipGraph (one per output slot, frame set, batch size and threshold) and
 * replay it: one launch instead of ~9.  The captured chain consists of kernel nodes only.  Ignored while per-kernel profiling is on. */
#define LMX_CTX_HIPGRAPH 1
/* Three device "lanes": lane 0 = the context's stream, the others = private streams with their own intermediate buffers.
 * Output slots alternate between the lanes and lmx_ctx_max_outstanding() = 6 lmx_ctx_enqueue calls may be outstanding instead
 * of two, so each stream always has its next batch queued and the kernels of one lane fill the tails of the others' (+17 %
 * throughput at 64 frames per batch on MI355X).  Results are unchanged.  Ordering: the private lanes start behind the most
 * recent upload; collect() returns results oldest first and waits on its own slot only; lmx_ctx_sync, uploads and debug reads
 * wait for every lane; lmx_ctx_export_raw is ordered on the context's stream behind the enqueue that produced the records.
 * May be combined with LMX_CTX_HIPGRAPH (one graph per output slot, frame set, batch size and threshold; the graphs of different
 * lanes replay concurrently: BASELINE config 5's per-GPU shape, tests/test_gpu_parity.py::test_config5_...). */
#define LMX_CTX_OVERLAP 2
/* Zero-copy input: sources that lie in pinned host memory (lmx_host_alloc, hipHostMalloc, hipHostRegister) are NOT staged by the
 * host; one kernel per modality pulls them over PCIe straight from the caller's buffers and lmx_ctx_upload / lmx_match* return
 * while that transfer is in flight: the caller keeps the pixels unchanged until lmx_ctx_upload_wait() or until a collect of an
 * enqueue that read them has returned.  Costs no host CPU time, moves 44 GB/s (the staged default: 54 GB/s with two or more
 * copy threads).  Without this flag every source, pinned or not, is copied before the call returns ("callee copies, never retains
 * pointers": the boundary's contract). */
#define LMX_CTX_ASYNC_INPUT 4

/* ---- bank ---------------------------------------------------------------------------------------------- */
lmx_status lmx_bank_create(const lmx_bank_desc* desc, lmx_bank** out);
/* templates: int32 [n_pyramids * L * M][5] = {width, height, pyramid_level, feat_begin, feat_count}, entry
 * l*M+m of pyramid p at row p*L*M + l*M + m; features: int32 [n_features_total][3] = {x, y, label}. */
lmx_status lmx_bank_add_class(lmx_bank* bank, const char* class_id, int32_t n_pyramids, const int32_t* templates,
                              const int32_t* features, int64_t n_features_total);
/* Trainer side, SURVEY.md 8f row 3: cv::linemod::Detector::addTemplate(sources, class_id, object_mask, &bounding_box)
 * (reference call sites src/renderer.cpp:308, src/renderer_only_image.cpp:266).  The per-pixel work (quantised orientations
 * with their magnitudes, quantised normals, the pyrDown chain) runs on the device with the same kernels match() uses; the
 * greedy scattered feature selection (extractTemplate / selectScatteredFeatures / cropTemplates) is sequential host code.
 * Sources as for match() (8UC3 / 16UC1 mm, any size: no T divisibility is needed to train); object_mask 8UC1 or NULL.
 * *template_id is the new id within the class, or -1 when some level has fewer candidates than num_features (upstream
 * returns -1 and adds nothing); bounding_box = {x, y, w, h} of cropTemplates. */
lmx_status lmx_bank_add_template(lmx_bank* bank, int32_t device, const lmx_image* sources, int32_t n_sources, const char* class_id,
                                 const lmx_image* object_mask, int32_t* template_id, int32_t bounding_box[4]);
/* DepthNormal's NORMAL_LUT: upstream quantises a unit normal with `NORMAL_LUT[v3][v2][v1]`, a constant
 * `uchar [20][20][20]` of one-hot labels that OpenCV ships as data (modules/.../normal_lut.i; used by the reference's trainers
 * and its carmine node through cv::linemod::DepthNormal, src/renderer.cpp:180-185, src/linemod_carmine_detect.cpp:802-840).
 * That file is not part of the reference repository, so the table is DATA on the bank here: the device kernels, the trainer
 * (lmx_bank_add_template) and the context all read the bank's table, indexed exactly like upstream (byte v3*400 + v2*20 + v1;
 * v1, v2, v3 may reach 20 and then run into the next row/plane as C's flat layout does; flat indices >= 8000, an out-of-bounds
 * read upstream, give "no label").  Entries must be 0 or one of 1, 2, 4, ..., 128.
 *   lmx_default_normal_lut     the documented default generator (azimuth octant of (nx, ny); NOT upstream's values)
 *   lmx_bank_set_normal_lut    install a table (NULL: explicitly choose the default generator)
 *   lmx_bank_load_normal_lut   the same from a file: 8000 raw bytes, or text holding 8000 integers such as OpenCV's normal_lut.i
 *   lmx_bank_normal_lut_origin LMX_LUT_*: where the bank's table came from
 * A bank read from a *_templates.yml that has a DepthNormal modality but neither the `lmx_normal_lut` marker that
 * lmx_bank_save_yaml writes nor a side-car table (`<yml>.normal_lut`, or the file named by the environment variable
 * LMX_NORMAL_LUT) was trained against a table this library cannot see: its origin is LMX_LUT_UNKNOWN and lmx_ctx_create
 * refuses it (LMX_ERR_INVALID_ARG) until one of the calls above states which table to match with. */
#define LMX_NORMAL_LUT_SIZE 8000
enum { LMX_LUT_DEFAULT = 0, LMX_LUT_USER = 1, LMX_LUT_SIDECAR = 2, LMX_LUT_UNKNOWN = 3 };
lmx_status lmx_default_normal_lut(uint8_t* out /* [LMX_NORMAL_LUT_SIZE] */);
lmx_status lmx_bank_set_normal_lut(lmx_bank* bank, const uint8_t* lut /* [LMX_NORMAL_LUT_SIZE] or NULL */);
lmx_status lmx_bank_get_normal_lut(const lmx_bank* bank, uint8_t* out /* [LMX_NORMAL_LUT_SIZE] */);
lmx_status lmx_bank_load_normal_lut(lmx_bank* bank, const char* path);
int32_t lmx_bank_normal_lut_origin(const lmx_bank* bank);
/* For readers that build a bank from a foreign yml themselves (the cv::FileNode facade): unless a table was installed with
 * set/load, take the one named by the environment variable LMX_NORMAL_LUT, else mark the bank LMX_LUT_UNKNOWN. */
lmx_status lmx_bank_require_normal_lut(lmx_bank* bank);
/* lmx_bank_save_yaml writes upstream's layout plus one extra top-level key OpenCV's reader ignores, `lmx_normal_lut: default`
 * or `lmx_normal_lut: sidecar`; in the second case the table goes to `<path>.normal_lut` (8000 raw bytes). */
lmx_status lmx_bank_load_yaml(const char* path, lmx_bank** out);
lmx_status lmx_bank_save_yaml(const lmx_bank* bank, const char* path);
void lmx_bank_destroy(lmx_bank* bank);
/* Deep copy (the copy is independent of `bank`; same templates, modalities, T and NORMAL_LUT). */
lmx_status lmx_bank_clone(const lmx_bank* bank, lmx_bank** out);
/* 64-bit content hash of everything that determines match() results: T, modality parameters, NORMAL_LUT, every class id,
 * template and feature.  Equal banks -> equal fingerprints; used as the key of the device-context cache below. */
uint64_t lmx_bank_fingerprint(const lmx_bank* bank);

/* The reference's service node rebuilds its detector on EVERY request: readLinemod(template yml) in the constructor
 * (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1784-1786 -> :204-264 -> :224, :708-721) and never frees it.  Two caches
 * make that cheap behind an unchanged call pattern:
 *   lmx_bank_load_yaml_cached: process-wide cache keyed by (path, mtime, size), backed by a binary side file: the yml is parsed
 *     once, later calls (and later processes) get the same immutable bank (reference-counted; give it back with lmx_bank_release, never lmx_bank_destroy, never modify it).
 *   lmx_ctx_acquire / lmx_ctx_unref: process-wide cache of device contexts keyed by (bank fingerprint, every lmx_ctx_desc
 *     field): a detector built again from the same templates for the same frame size gets the context that is already
 *     resident in HBM instead of uploading the bank again.  The cached context owns a private copy of the bank, so the
 *     caller's bank may be modified or destroyed at any time.  lmx_ctx_unref only drops the reference; up to 8 idle contexts
 *     stay cached (least recently used goes first).  A context's calls must not overlap: see lmx_ctx_lock below (lmx_match / lmx_match_batch lock by themselves). */
lmx_status lmx_bank_load_yaml_cached(const char* path, const lmx_bank** out);
void lmx_bank_release(const lmx_bank* bank);
/* Compact binary form of a bank (templates, modalities, T, NORMAL_LUT; checksummed): a 3000-template RGB-D bank is 22.7 MB of
 * FileStorage YAML and 7 MB here, and loads in 14 ms instead of half a second (seconds with OpenCV's parser).
 * lmx_bank_load_yaml_cached keeps one next to the yml ("<yml>.lmxcache", tagged with the yml's mtime and size; written when the
 * directory allows it, LMX_NO_DISK_CACHE=1 disables), so a restarted node does not parse the YAML again either. */
lmx_status lmx_bank_save_binary(const lmx_bank* bank, const char* path);
lmx_status lmx_bank_load_binary(const char* path, lmx_bank** out);

/* The FileStorage-YAML document tree behind lmx_bank_load_yaml, for callers that walk a *_templates.yml themselves (the
 * cv::FileNode-shaped facade include/lmx_cv_linemod.hpp reads banks through Detector::read(FileNode) / readClass(FileNode)
 * like the reference's readLinemod, src/rgbdDetector.cpp:1668-1680).  Nodes are owned by the document. */
typedef struct lmx_yaml_doc lmx_yaml_doc;
typedef struct lmx_yaml_node lmx_yaml_node;
enum { LMX_YAML_NULL = 0, LMX_YAML_SCALAR = 1, LMX_YAML_SEQ = 2, LMX_YAML_MAP = 3 };
lmx_status lmx_yaml_open(const char* path, lmx_yaml_doc** out);
void lmx_yaml_close(lmx_yaml_doc* doc);
const lmx_yaml_node* lmx_yaml_root(const lmx_yaml_doc* doc);
int32_t lmx_yaml_kind(const lmx_yaml_node* node);
const char* lmx_yaml_scalar(const lmx_yaml_node* node);                       /* "" unless the node is a scalar */
int32_t lmx_yaml_size(const lmx_yaml_node* node);                             /* items of a sequence / entries of a map, else 0 */
const lmx_yaml_node* lmx_yaml_item(const lmx_yaml_node* node, int32_t i);     /* sequence item i / value of map entry i, or NULL */
const char* lmx_yaml_key(const lmx_yaml_node* node, int32_t i);               /* key of map entry i, or NULL */
const lmx_yaml_node* lmx_yaml_get(const lmx_yaml_node* node, const char* key); /* map lookup, or NULL */

int32_t lmx_bank_pyramid_levels(const lmx_bank* bank);
int32_t lmx_bank_T(const lmx_bank* bank, int32_t level);
int32_t lmx_bank_num_modalities(const lmx_bank* bank);
lmx_status lmx_bank_modality(const lmx_bank* bank, int32_t index, lmx_modality_desc* out);
int32_t lmx_bank_num_classes(const lmx_bank* bank);
const char* lmx_bank_class_id(const lmx_bank* bank, int32_t class_index); /* sorted (std::map) order */
int32_t lmx_bank_num_templates(const lmx_bank* bank, const char* class_id /* NULL = all classes */);
/* Template k = l*M+m of pyramid `template_id`; *features points at n_features x {x,y,label} owned by the bank. */
lmx_status lmx_bank_get_template(const lmx_bank* bank, const char* class_id, int32_t template_id, int32_t k,
                                 int32_t* width, int32_t* height, int32_t* pyramid_level, const int32_t** features,
                                 int32_t* n_features);

/* ---- device context ------------------------------------------------------------------------------------ */
lmx_status lmx_ctx_create(const lmx_bank* bank, const lmx_ctx_desc* desc, lmx_ctx** out);
void lmx_ctx_destroy(lmx_ctx* ctx);
/* Cached form (see lmx_bank_load_yaml_cached above).  *cache_hit (may be NULL) = 1 when an existing context was returned. */
lmx_status lmx_ctx_acquire(const lmx_bank* bank, const lmx_ctx_desc* desc, lmx_ctx** out, int32_t* cache_hit);
void lmx_ctx_unref(lmx_ctx* ctx);
/* Drops what the two process-wide caches hold without a user: device contexts nobody references (their device and pinned memory,
 * streams) and cached banks nobody references.  A long-running node with many template files may call it to give memory back; it also
 * matters for speed: an idle context alive in the process can cost ANOTHER context's pipelined matching 4-8 % (stream / hardware-queue
 * placement; scripts/idle_context_effect.py).  Referenced entries stay. */
void lmx_cache_trim(void);
/* A context's calls must not overlap.  The synchronous composites lmx_match / lmx_match_batch take the context's (recursive) lock
 * themselves, so callers that only use those -- e.g. two cv::linemod::Detector objects of the facade that were given the SAME cached
 * context by lmx_ctx_acquire, matching from two threads -- are serialised by the library.  Users of the split-phase calls
 * (upload / enqueue / collect / debug reads) that share a context across threads bracket their sequence with these. */
void lmx_ctx_lock(lmx_ctx* ctx);
void lmx_ctx_unlock(lmx_ctx* ctx);

/* "Next" row 4 of SURVEY.md 8f: the node-side steps immediately before match(), fused on the device so that the raw
 * camera frame is uploaded once and never touched by the host:
 *   MONO8 -> BGR by channel replication (mixChannels)          src/linemod_ensenso_detect_3_mult_detect_service.cpp:293-297
 *   cv::GaussianBlur(img, img, Size(3,3), 0, 0) on the FULL frame, then the crop Rect(bias_x, 0, 640, 480)   ...:324-326
 *   depth in float metres -> 16U millimetres, convertTo(CV_16UC1, 1000.0)   ...:837-858, src/linemod_carmine_detect.cpp:829-839
 * A source whose size equals (src_height, src_width) is cropped at (crop_x, crop_y) to the context's frame size; a source
 * that already has the context's size is taken as is.  Colour: 8UC3, or 8UC1 when `mono`; depth: 16UC1 mm, or 32FC1
 * metres when `depth_float_m`. */
typedef struct lmx_pre_desc {
  int32_t src_width, src_height;
  int32_t crop_x, crop_y;
  int32_t blur3;
  int32_t mono;
  int32_t depth_float_m;
} lmx_pre_desc;
lmx_status lmx_ctx_upload_raw(lmx_ctx* ctx, int32_t n_frames, const lmx_image* sources, int32_t n_sources, const lmx_pre_desc* pre);

/* The drop-in call.  Clears nothing on the caller's side: writes up to `cap` matches in upstream output order
 * (std::sort + std::unique applied) and the total in *n_out (LMX_ERR_OVERFLOW if more than cap).
 * class_ids == NULL / n_class_ids == 0 matches every class (what the reference always passes). */
lmx_status lmx_match(lmx_ctx* ctx, const lmx_image* sources, int32_t n_sources, float threshold,
                     const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out);
/* n_frames independent frames; sources[f*n_sources + m]; out[f*cap ...], n_out[f]. */
lmx_status lmx_match_batch(lmx_ctx* ctx, int32_t n_frames, const lmx_image* sources, int32_t n_sources, float threshold,
                           const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap,
                           size_t* n_out);

/* Split-phase form of the same call, for inputs kept resident in HBM (bench, streaming, multi-GPU):
 *   upload  : host frames -> the NEXT of the context's frame sets (device frames + pinned staging each; one more set than
 *             device lanes), asynchronous on a private copy stream: the call copies pageable sources into the set's pinned
 *             staging with a few host threads (LMX_UPLOAD_THREADS, default min(8, cores/2)) and queues the transfer; it waits
 *             for no kernel.  The transfer of batch i+1 therefore overlaps the kernels of batch i, which is how the host-frame
 *             boundary of match() is pipelined: upload(i+1); enqueue(i+1); collect(i) ...
 *   enqueue : the whole kernel chain for frames [0, n_frames) of the most recently uploaded set, behind its transfer (device-
 *             side wait), no host synchronisation
 *   collect : wait for the OLDEST outstanding enqueue, take its match records (their read-back was queued behind its
 *             kernels), restore insertion order, std::sort + std::unique
 * Up to lmx_ctx_max_outstanding() enqueues may be outstanding (two output slots per device lane), so the host-side
 * finalisation of batch i overlaps the kernels of the following batches:  enqueue(0); loop { enqueue(i+1); collect(i); }
 * One enqueue more than that without a collect is an error. */
lmx_status lmx_ctx_upload(lmx_ctx* ctx, int32_t n_frames, const lmx_image* sources, int32_t n_sources);
/* Detector::match's last argument, `masks` (one 8UC1 Mat per modality, or an empty vector; the reference passes none,
 * src/rgbdDetector.cpp:33): the quantised labels of a modality survive where its mask is non-zero, at every pyramid level (upstream
 * QuantizedPyramid::quantize copies through the mask and halves the mask per level with INTER_NEAREST).  masks[f * n_masks + m] for the
 * first n_frames frames of the MOST RECENT upload (n_frames may be smaller than what was uploaded: the remaining frames are matched
 * unmasked); an entry with data == NULL means "no mask for this source".  The masks stay attached to those frames until the next upload
 * or the next lmx_ctx_upload_masks.  Batches with masks run the plain kernel chain (no graph replay, no fused small-batch launches). */
lmx_status lmx_ctx_upload_masks(lmx_ctx* ctx, int32_t n_frames, const lmx_image* masks, int32_t n_masks);
/* lmx_match with masks (masks == NULL: plain lmx_match): upload + upload_masks + enqueue + collect under the context's lock. */
lmx_status lmx_match_masked(lmx_ctx* ctx, const lmx_image* sources, const lmx_image* masks, int32_t n_sources, float threshold,
                            const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out);
/* Host-side wait for the most recent upload's transfer (see LMX_CTX_ASYNC_INPUT). */
lmx_status lmx_ctx_upload_wait(lmx_ctx* ctx);
/* Pinned host memory for frames (camera drivers / benchmarks that want the zero-copy path): hipHostMalloc / hipHostFree. */
lmx_status lmx_host_alloc(size_t bytes, void** out);
void lmx_host_free(void* p);
/* Matches the first n_frames frames of the MOST RECENT upload (n_frames may be smaller than what was uploaded, not larger:
 * LMX_ERR_INVALID_ARG). */
lmx_status lmx_ctx_enqueue(lmx_ctx* ctx, int32_t n_frames, float threshold, const char* const* class_ids,
                           int32_t n_class_ids);
lmx_status lmx_ctx_collect(lmx_ctx* ctx, int32_t n_frames, lmx_match_t* out, size_t cap, size_t* n_out);
/* Same, frames packed back to back: frame f's matches are out[offsets[f] .. offsets[f+1]); offsets has n_frames+1 entries. */
lmx_status lmx_ctx_collect_flat(lmx_ctx* ctx, int32_t n_frames, lmx_match_t* out, size_t cap_total, size_t* offsets);

/* Multi-GPU plumbing: device pointers of the shard-local raw match records written by enqueue
 * (records: lmx_raw_match_t [capacity], all frames of the batch in one list, each tagged with its frame;
 * counts: uint32 [16] header, [0] = coarse candidates, [1] = records written) so a caller can all-gather
 * them over RCCL, and the host merge that turns gathered records of ONE frame into the final match list. */
lmx_status lmx_ctx_raw_matches(lmx_ctx* ctx, void** d_records, void** d_counts, size_t* capacity);
lmx_status lmx_merge_raw(const lmx_raw_match_t* records, size_t n_records, lmx_match_t* out, size_t cap, size_t* n_out);
/* Enqueue ONE device-to-device copy of this rank's gather block into a caller-owned device buffer (e.g. the send
 * buffer of an RCCL all-gather) on the context's stream.  Block layout (LMX_GATHER_HEADER_BYTES + capacity_records *
 * sizeof(lmx_raw_match_t) bytes): uint32 header[16] with header[0] = coarse candidates, header[1] = records written,
 * header[2] = capacity of the rank's candidate list (0 = not stated; header[0] > header[2] means candidates were dropped and
 * lmx_merge_gathered reports LMX_ERR_OVERFLOW), then the records.  Only the header and the first min(header[1], capacity_records) records are written; the rest of
 * the block keeps whatever it held. */
#define LMX_GATHER_HEADER_BYTES 64
lmx_status lmx_ctx_export_raw(lmx_ctx* ctx, void* d_block, size_t capacity_records);
/* The same copy on a caller-chosen stream (e.g. a communication stream that carries the all-gather): `stream` first waits,
 * on the device, for the most recent enqueue; nothing is queued on the context's own streams, so further enqueues are not
 * held up behind the exchange. */
lmx_status lmx_ctx_export_raw_on(lmx_ctx* ctx, void* d_block, size_t capacity_records, void* stream);
/* The same for the OLDEST outstanding enqueue (the one the next collect / release refers to) instead of the most recent one: a
 * pipelined caller that finds, when it finishes batch i, that its gather block was too small re-exports batch i's records --
 * they stay in the output slot until the slot is released -- into a larger block while later batches are already enqueued. */
lmx_status lmx_ctx_export_oldest_on(lmx_ctx* ctx, void* d_block, size_t capacity_records, void* stream);
/* Copy `bytes` (multiple of 16, both pointers 16-byte aligned) on `stream` with a kernel instead of a DMA engine; `dst` may
 * be pinned host memory.  For small latency-sensitive read-backs in a pipelined caller: hipMemcpyAsync(DeviceToHost) was
 * measured to block the submitting thread for milliseconds now and then when copies of several streams are in flight. */
lmx_status lmx_stream_copy(void* dst, const void* src, size_t bytes, void* stream);
/* The same for `n_blocks` gather blocks (the result of an all-gather, one block per rank, `block_stride_bytes` apart in
 * both buffers): of every block only the header and the records it counts are copied, which is what lmx_merge_gathered
 * reads; a few KB over PCIe instead of n_blocks * capacity * 32 bytes. */
lmx_status lmx_stream_copy_blocks(void* dst, const void* src, int32_t n_blocks, size_t block_stride_bytes, size_t capacity_records, void* stream);
/* How many lmx_ctx_enqueue calls may be outstanding before one has to be collected (2; 6 with LMX_CTX_OVERLAP). */
int32_t lmx_ctx_max_outstanding(const lmx_ctx* ctx);
/* Drop the OLDEST outstanding enqueue without reading it back and free its output slot: waits (host) until it has
 * finished, like collect, but moves no data.  For callers that consume the records on the device (lmx_ctx_export_raw*). */
lmx_status lmx_ctx_release(lmx_ctx* ctx);
/* Host merge of `n_ranks` gathered blocks (each `block_stride_bytes` apart, layout above) into the final per-frame match
 * lists: frame f's matches are out[offsets[f] .. offsets[f+1]).  LMX_ERR_OVERFLOW if a rank wrote more records than
 * its block holds (raise the gather capacity) or cap_total is too small. */
lmx_status lmx_merge_gathered(const void* blocks, int32_t n_ranks, size_t block_stride_bytes, size_t capacity_records,
                              int32_t n_frames, lmx_match_t* out, size_t cap_total, size_t* offsets);
/* The same merge for a frame_groups x template_shards grid of ranks (see lmx_group below): n_ranks = G * R, rank k belongs to frame group
 * k / R and its records carry frame indices local to that group, whose frames are [g*n_frames/G, (g+1)*n_frames/G) of the batch.
 * frame_groups = 1 is lmx_merge_gathered. */
lmx_status lmx_merge_gathered_groups(const void* blocks, int32_t n_ranks, size_t block_stride_bytes, size_t capacity_records, int32_t n_frames,
                                     int32_t frame_groups, lmx_match_t* out, size_t cap_total, size_t* offsets);
/* ---- multi-GPU from the C++ side -------------------------------------------------------------------------
 * The caller of the hot path is C++ (rgbdDetector::linemod_detection, src/rgbdDetector.cpp:31-34): a group spreads a batch of frames
 * and the bank over several GPUs behind the same call shape.  The `world` members form a frame_groups x template_shards grid
 * (world = G * R, member k = frame group k / R, template shard k % R):
 *   member (g, r) pre-processes and matches frames [g*n/G, (g+1)*n/G) of a batch of n frames against templates [r*N/R, (r+1)*N/R)
 *   of every class.
 * G = 1 is pure template sharding (every member pre-processes the same frames: right for ONE frame's latency, BASELINE configs[3]);
 * R = 1 is pure frame sharding (every GPU holds the whole bank -- tens of MB at 50 000 templates -- and takes n/G of the frames: no
 * replicated pre-processing at all, right for streams of frames, BASELINE configs[4]).  Either way the only exchange is ONE RCCL
 * all-gather per batch of fixed-capacity per-member raw-record blocks (layout above), merged on the host exactly like the single-GPU
 * path (a member's records carry frame indices local to its group; lmx_merge_gathered_groups maps them back), so results are identical
 * for any G x R.  RCCL is dlopen'ed on first use.
 *   single process : n_devices GPUs (devices[] or 0..n-1), ncclCommInitAll; unique_id = NULL
 *   one process per GPU: unique_id = the 128 bytes lmx_group_unique_id produced on rank 0 (broadcast by the launcher), rank,
 *                    world, device
 * A rank that produces more records than gather_capacity does not fail: the exchanged headers carry the true counts, the blocks
 * are re-allocated to fit and the exchange is repeated (SURVEY 8e's two-phase fallback). */
typedef struct lmx_group lmx_group;
/* How the per-rank blocks are exchanged.  RCCL is the default and the only choice across processes.  PEER_COPY is a
 * single-process all-gather made of device-to-device block copies between the members' buffers (every member pulls the other
 * members' blocks on its own communication stream, ordered by events): no communicator, works between devices with peer access
 * AND between members that share a device, which RCCL refuses -- devices[] may then repeat a device id, so a group of any size
 * runs (and is tested) on a single GPU.  The environment variable LMX_GROUP_COLLECTIVE=rccl|peer overrides the field. */
#define LMX_GROUP_COLLECTIVE_RCCL 0
#define LMX_GROUP_COLLECTIVE_PEER_COPY 1
typedef struct lmx_group_desc {
  int32_t n_devices;         /* single-process mode */
  const int32_t* devices;    /* [n_devices] or NULL = 0..n_devices-1 */
  int32_t width, height, max_batch, max_candidates;
  int32_t gather_capacity;   /* records per rank in the all-gather block; 0 = 8192 */
  int32_t flags;             /* LMX_CTX_* for the member contexts */
  const void* unique_id;     /* multi-process mode: 128 bytes from lmx_group_unique_id, else NULL */
  int32_t rank, world, device;
  int32_t collective;        /* LMX_GROUP_COLLECTIVE_* */
  int32_t frame_groups;      /* G: 0 or 1 = template sharding only; must divide the number of members.  max_batch stays the size of the
                                whole batch the group accepts; a member's context holds ceil(max_batch / G) frames */
} lmx_group_desc;
lmx_status lmx_group_unique_id(void* out128);
lmx_status lmx_group_create(const lmx_bank* bank, const lmx_group_desc* desc, lmx_group** out);
void lmx_group_destroy(lmx_group* group);
int32_t lmx_group_size(const lmx_group* group);            /* members = frame_groups * template shards */
int32_t lmx_group_frame_groups(const lmx_group* group);   /* G */
int32_t lmx_group_gather_capacity(const lmx_group* group);   /* grows when a batch needed the two-phase fallback */
const char* lmx_group_collective_name(const lmx_group* group);   /* "rccl" or "peer_copy" */
/* lmx_match_batch over the group: out[f*cap ...], n_out[f], upstream output order.  = upload + submit + finish. */
lmx_status lmx_group_match_batch(lmx_group* group, int32_t n_frames, const lmx_image* sources, int32_t n_sources, float threshold,
                                 const char* const* class_ids, int32_t n_class_ids, lmx_match_t* out, size_t cap, size_t* n_out);
/* Split-phase form, mirroring lmx_ctx_upload / _enqueue / _collect (and linemod_pose_estimation_amd/dist.py ShardedMatcher):
 *   upload : the host frames are staged ONCE into pinned memory (one non-temporal copy by the group's host threads) and every
 *            member DMAs the same staging area into its own next frame set over its own PCIe link; waits for no kernel.
 *   submit : every member enqueues its kernel chain for the most recent upload and exports its records into the batch's send
 *            block on its communication stream (device-side wait on the enqueue); then ONE all-gather and the copy of rank 0's
 *            view (headers + counted records) into pinned host memory are queued.  Returns without waiting for the device; the
 *            members are driven by the group's host threads (one per member, at most 8).
 *   finish : waits for the OLDEST submitted batch, re-runs its exchange with larger blocks if a rank had more records than the
 *            block held (the records are still in the members' output slots), merges on the host, frees the members' slots.
 * Up to lmx_group_depth() batches may be in flight, so the exchange and the host merge of batch i overlap the kernels of the
 * following batches:  upload(0); submit(0); loop { upload(i+1); submit(i+1); finish(i); } */
lmx_status lmx_group_upload(lmx_group* group, int32_t n_frames, const lmx_image* sources, int32_t n_sources);
lmx_status lmx_group_submit(lmx_group* group, int32_t n_frames, float threshold, const char* const* class_ids, int32_t n_class_ids);
lmx_status lmx_group_finish(lmx_group* group, int32_t n_frames, lmx_match_t* out, size_t cap, size_t* n_out);
int32_t lmx_group_depth(const lmx_group* group);

/* Synchronise the context's stream and fold pending profiling events (what collect does, without a read-back). */
lmx_status lmx_ctx_sync(lmx_ctx* ctx);

/* ---- "next" row 2 of SURVEY.md 8f: the immediate consumer of match() ------------------------------------------------
 * rcd_voting -> cluster_filter -> cluster_scoring (mean similarity) -> nonMaximaSuppressionUsingIOU, as chained by the nodes
 * (src/linemod_ensenso_detect_3_mult_detect_service.cpp:376-447; functions src/rgbdDetector.cpp:36-70, 72-85, 118-144, 462-574).
 * Host code on purpose: cluster membership depends on which duplicates std::unique removed, i.e. on libstdc++'s order of
 * ties in the final sort, so it must run on the finalised list.  Differences from the reference, both documented in
 * cluster_filter(map, thresh) erases while iterating (undefined behaviour there); here every cluster with
 * size <= thresh is removed.  `neighborSize` of the NMS is unused upstream (IoU threshold 0.4 is hard-coded). */
typedef struct lmx_cluster_params {
  int32_t vote_row_col_step;   /* clustering_step_ */
  double renderer_radius_min;
  double renderer_radius_step;
  int32_t cluster_size_thresh; /* clusters with size <= thresh are dropped (the nodes pass 2, carmine 0) */
} lmx_cluster_params;
typedef struct lmx_cluster_t {
  int32_t index[3];  /* {y / step, x / step, depth ring} */
  int32_t rect[4];   /* x, y, width, height: means over the cluster's matches (integer division) */
  double score;      /* mean similarity */
  int32_t member_begin, member_count; /* range in `members` (indices into `matches`, in cluster order) */
} lmx_cluster_t;
/* obj_origin_dists[template_id], rects[template_id][4] = {x, y, w, h} are the renderer-params side-car
 * (src/rgbdDetector.cpp:1681-1749).  Clusters come out in upstream's final order (sorted by score, NMS survivors only). */
lmx_status lmx_cluster_matches(const lmx_match_t* matches, size_t n_matches, const double* obj_origin_dists, const int32_t* rects,
                               size_t n_templates, const lmx_cluster_params* params, lmx_cluster_t* clusters, size_t cap_clusters,
                               size_t* n_clusters, int32_t* members, size_t cap_members);

/* The renderer-params side-car that the consumer chain reads next to the matches: `<object>_renderer_params.yml`, written by the
 * reference's trainers (writeLinemodTemplateParams, src/renderer.cpp:72-130) and read by readLinemodTemplateParams
 * (src/rgbdDetector.cpp:1681-1749): per template "Template <i>": {R 3x3, T 3x1, K 3x3, D, Ori_dist, Rect}, then the renderer_* scalars.
 * obj_origin_dists / rects / renderer_radius_min / _step are exactly what lmx_cluster_matches and lmx_ctx_set_cluster_sidecar take
 * (Ori_dist and D pass through a float, as in the reference).  Arrays are owned by the struct: lmx_renderer_params_free. */
typedef struct lmx_renderer_params {
  size_t n_templates;
  double* obj_origin_dists;   /* [n] Ori_dist */
  int32_t* rects;             /* [n][4] Rect x, y, width, height */
  double* distances;          /* [n] D */
  double* R;                  /* [n][9] row major */
  double* T;                  /* [n][3] */
  double* K;                  /* [n][9] */
  int32_t renderer_n_points, renderer_angle_step, renderer_width, renderer_height;
  double renderer_radius_min, renderer_radius_max, renderer_radius_step;
  double renderer_focal_length_x, renderer_focal_length_y, renderer_near, renderer_far;
} lmx_renderer_params;
lmx_status lmx_renderer_params_load(const char* path, lmx_renderer_params** out);
lmx_status lmx_renderer_params_save(const lmx_renderer_params* params, const char* path);
void lmx_renderer_params_free(lmx_renderer_params* params);

/* The same consumer chain ON THE DEVICE, fed from the raw-match slot of an enqueue instead of a host list: one kernel per batch
 * (csrc/lmx_f2.hip, one workgroup per frame) restores upstream insertion order, applies Detector::match's std::sort + std::unique
 * and the reference's rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU, reproducing libstdc++'s
 * order of ties in both sorts (csrc/lmx_sort_emul.hpp), so the output equals lmx_ctx_collect + lmx_cluster_matches for every
 * input.  Only the final matches and the surviving clusters cross PCIe.
 *   lmx_ctx_set_cluster_sidecar : obj_origin_dists[n_templates], rects[n_templates][4], parameters (copied to the device)
 *   lmx_ctx_collect_clusters    : replaces lmx_ctx_collect for the OLDEST outstanding enqueue.  Frame f's matches are
 *     matches[match_offsets[f] .. match_offsets[f+1]), its clusters clusters[cluster_offsets[f] .. cluster_offsets[f+1]);
 *     a cluster's members are members[member_begin .. member_begin + member_count), indices into ITS FRAME's matches.
 *     matches may be NULL (cap_matches 0) when only clusters are wanted.  Frames with more than 2048 raw records, or whose
 *     bins fall outside +-65536, are finished by the host path transparently. */
lmx_status lmx_ctx_set_cluster_sidecar(lmx_ctx* ctx, const double* obj_origin_dists, const int32_t* rects, size_t n_templates,
                                       const lmx_cluster_params* params);
lmx_status lmx_ctx_collect_clusters(lmx_ctx* ctx, int32_t n_frames, lmx_match_t* matches, size_t cap_matches, size_t* match_offsets,
                                    lmx_cluster_t* clusters, size_t cap_clusters, size_t* cluster_offsets, int32_t* members, size_t cap_members);

/* ---- introspection (stage-level parity tests, profiling) ------------------------------------------------- */
enum {
  LMX_DBG_QUANTIZED = 0,     /* u8 [H_l][W_l] one-hot labels after quantize(), A.2/A.4 */
  LMX_DBG_LINEAR_MEMORY = 1, /* u8 [8][T*T][W'H'] in upstream linearize() layout, A.7 */
  LMX_DBG_PYRAMID_BGR = 2,   /* u8 [H_l][W_l][3] colour source at level l (pyrDown chain), A.3 */
  LMX_DBG_DEPTH = 3          /* u16 [H][W] level-0 depth in mm as the DepthNormal modality sees it */
};
lmx_status lmx_ctx_debug_read(lmx_ctx* ctx, int32_t frame, int32_t what, int32_t level, int32_t modality, void* out,
                              size_t out_bytes);
/* Test hook for the float stage: the 16-bin orientation label (0..16, before upstream's '& 7') the device code assigns to
 * n gradients (dx[i], dy[i]) -- fastAtan2 in degrees, times 16/360, round half to even. */
lmx_status lmx_debug_orientation_labels(int32_t device, const int16_t* dx, const int16_t* dy, size_t n, uint8_t* out);
/* Test hooks: the permutation produced by the library's restatement of libstdc++'s std::sort (csrc/lmx_sort_emul.hpp, the code the
 * device runs to reproduce upstream's order of ties) for Match::operator< and for the clusters' score-descending comparator. */
lmx_status lmx_debug_introsort_perm(const float* similarity, const int32_t* template_id, int32_t n, int32_t* perm);
lmx_status lmx_debug_introsort_perm_score(const double* score, int32_t n, int32_t* perm);
/* The permutation the DEVICE's workgroup-parallel form of the same algorithm (csrc/lmx_sort_block.hpp, what k_f2_finalize_cluster runs
 * for Detector::match's std::sort) produces for n <= 2048 (similarity, template_id) pairs: must equal lmx_debug_introsort_perm's. */
lmx_status lmx_debug_device_sort_perm(int32_t device, const float* similarity, const int32_t* template_id, int32_t n, int32_t* perm);
/* Counters of the last collect(): coarse candidates and refined matches summed over frames. */
lmx_status lmx_ctx_stats(lmx_ctx* ctx, int64_t* n_candidates, int64_t* n_raw_matches);

/* Per-kernel HIP-event timing on the context's stream (off by default).  `enabled` is a bitmask over kernel ids
 * (bit k = time kernel k; -1 = all): every timed launch costs two event records, so time only what is reported. */
int32_t lmx_num_kernels(void);
const char* lmx_kernel_name(int32_t kernel_id);
/* Name of the device kernel this context launches for a kernel id (what a profiler shows), e.g. "k_score_coarse_u8" for
 * LMX kernel id "k_score_coarse" when every template has <= 63 coarsest-level features, "k_depth_quantize<int>", ... */
const char* lmx_ctx_device_kernel_name(lmx_ctx* ctx, int32_t kernel_id);
lmx_status lmx_ctx_set_profiling(lmx_ctx* ctx, int32_t enabled);
lmx_status lmx_ctx_kernel_time(lmx_ctx* ctx, int32_t kernel_id, double* total_ms, int64_t* launches);
lmx_status lmx_ctx_reset_profiling(lmx_ctx* ctx);
/* Algorithmic bytes one enqueue() of n_frames moves through kernel `kernel_id` (SURVEY.md 8d formula). */
lmx_status lmx_ctx_algorithmic_bytes(lmx_ctx* ctx, int32_t kernel_id, int32_t n_frames, double* bytes);

const char* lmx_last_error(void);
const char* lmx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LMX_H_ */
