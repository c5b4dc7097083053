/*
 * vj.h — C ABI of the MI355X-native Viola–Jones detect path (libvjhip.so).
 *
 * This is the drop-in boundary for the reference's clif/clod detect path
 * (GabrieleCocco/CLFaceDetection).  Every entry point names the reference
 * interface it replaces (file:line, relative to CLFaceDetection/).  The
 * reference's own boundary is C++ with OpenCV/OpenCL types; neither exists on
 * the target, so this header uses plain pointers, sizes and POD structs only
 * (no torch types, no C++ types).  Errors are integer return codes — the
 * reference calls exit() through clCheckOrExit (clod.cpp:114…); this library
 * never exits the process.
 *
 * There is NO CPU fallback behind these calls: every function that computes on
 * images runs hand-written HIP kernels on a gfx950 device and fails with
 * VJ_ERR_NO_DEVICE when none is usable.  Host-only helpers (cascade loading,
 * scale planning, feature-table construction) are the reference's host logic
 * (clod.cpp:371-415, 529-578) and run on the CPU in the reference too.
 */
#ifndef VJ_H_
#define VJ_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(VJ_BUILDING) && defined(__GNUC__)
#pragma GCC visibility push(default)   /* only the names declared here are exported */
#endif

/* ------------------------------------------------------------------ errors */
enum {
    VJ_OK = 0,
    VJ_ERR_ARG = 1,         /* null / out-of-range argument                      */
    VJ_ERR_IO = 2,          /* file cannot be opened / read / written            */
    VJ_ERR_PARSE = 3,       /* malformed cascade file                            */
    VJ_ERR_UNSUPPORTED = 4, /* e.g. tilted features (clod ignores them; we refuse) */
    VJ_ERR_NO_DEVICE = 5,   /* no usable HIP device                              */
    VJ_ERR_HIP = 6,         /* a HIP runtime call failed (see vj_last_error)     */
    VJ_ERR_NOMEM = 7,
    VJ_ERR_LIMIT = 8        /* image / batch exceeds an addressing limit         */
};
const char* vj_strerror(int code);
/* Thread-local detail string for the last failing call on this thread. */
const char* vj_last_error(void);

/* ----------------------------------------------------------------- cascade */
/* Replaces cvLoad(xml) → CvHaarClassifierCascade* (main.cpp:36; loader spec
 * tempcv.cpp:1749-2089, struct layout tempcv.hpp:70-112).                      */
typedef struct vj_cascade vj_cascade;

typedef struct vj_cascade_info {
    int32_t win_w, win_h;      /* orig_window_size                              */
    int32_t n_stages, n_trees, n_nodes, n_alpha;
    int32_t max_trees_per_stage, max_nodes_per_tree;
    int32_t n_tilted;          /* nodes with <tilted>1                          */
    int32_t n_three_rect;      /* nodes with 3 rectangles                       */
    int32_t is_stump_based;    /* every tree has exactly one node               */
    int32_t is_stage_tree;     /* some stage has next != -1                     */
} vj_cascade_info;

typedef struct vj_stage_desc {  /* CvHaarStageClassifier (tempcv.hpp:95-105)    */
    int32_t first_tree, n_trees;
    float   threshold;
    int32_t parent, next, child;
} vj_stage_desc;

typedef struct vj_rect_desc { int32_t x, y, w, h; float weight; } vj_rect_desc;

typedef struct vj_node_desc {   /* one node of a CvHaarClassifier (tempcv.hpp:81-93) */
    int32_t n_rects;            /* 2 or 3 (rects with weight != 0)              */
    int32_t tilted;
    float   threshold;
    int32_t left, right;        /* >0: node index inside the tree; <=0: alpha[-v] */
    vj_rect_desc rect[3];
} vj_node_desc;

typedef struct vj_tree_desc { int32_t first_node, n_nodes, first_alpha; } vj_tree_desc;

/* OpenCV old-format XML (<opencv_storage><NAME type_id="opencv-haar-classifier">). */
int  vj_cascade_load_xml(const char* path, vj_cascade** out);
/* Compact binary form shipped under clfacedetection_amd/data (*.vjc). */
int  vj_cascade_load(const char* path, vj_cascade** out);
int  vj_cascade_save(const vj_cascade* c, const char* path);
/* A cascade the caller already holds in memory — the reference's callers get a CvHaarClassifierCascade*
 * from cvLoad (main.cpp:36; struct layout tempcv.hpp:70-112) and hand it to clodDetectObjects unchanged
 * (clod.h:72-81).  The arrays are copied and validated; stage `child` links are derived as
 * icvReadHaarClassifier does (tempcv.cpp:2080-2083) when every child is -1.  INTEGRATION.md shows the
 * CvHaarClassifierCascade -> arrays walk that keeps clodDetectObjects' signature.                       */
int  vj_cascade_from_arrays(int win_w, int win_h, const vj_stage_desc* stages, int n_stages,
                            const vj_tree_desc* trees, int n_trees, const vj_node_desc* nodes, int n_nodes,
                            const float* alpha, int n_alpha, vj_cascade** out);
void vj_cascade_free(vj_cascade* c);
int  vj_cascade_get_info(const vj_cascade* c, vj_cascade_info* out);
/* Read-only views into the flat arrays (valid until vj_cascade_free). */
const vj_stage_desc* vj_cascade_stages(const vj_cascade* c);
const vj_tree_desc*  vj_cascade_trees(const vj_cascade* c);
const vj_node_desc*  vj_cascade_nodes(const vj_cascade* c);
const float*         vj_cascade_alpha(const vj_cascade* c);
const char*          vj_cascade_notice(const vj_cascade* c);  /* license text of the source XML */

/* ------------------------------------------------------------------ params */
/* Arguments of clodDetectObjects (clod.h:72-81) that are not the image/cascade. */
enum {
    VJ_FLAG_COUNTERS     = 1u << 0, /* fill vj_result.counters + stage_entered   */
    VJ_FLAG_SIGNED_MEAN  = 1u << 1, /* reproduce clod.cpp:426 literally: window
                                       pixel sum read through int* (differs from
                                       the default unsigned read only when the
                                       sum >= 2^31)            */
    /* The reference's CPU variants thin the window grid with a data-dependent skip ("P2");
     * its OpenCL kernel — the default contract here, "P1" — evaluates every grid window.  Linear cascades. */
    VJ_FLAG_SKIP_LIST    = 1u << 2, /* CLOD_PER_STAGE_ITERATIONS CPU variant (clod.cpp:1434-1482, runSubwindow
                                       :681-734): after a stage-0 reject the next entry of the FLATTENED
                                       row-major window list is not evaluated (:729-732; crosses row ends) */
    VJ_FLAG_SKIP_ROW     = 1u << 3, /* plain CPU variant (clod.cpp:1409-1432): window positions are
                                       round(index * step) — half away from zero (:1416) instead of
                                       precomputeWindows' lrint (:514) — and the next window of the ROW is
                                       skipped after a stage-0 reject (x_incr, :1430)                      */
    VJ_FLAG_GRID_F64     = 1u << 4, /* with one of the two flags above: the same loop inside the block variant
                                       (CLOD_BLOCK_IMPLEMENTATION, clod.cpp:821-1173), which keeps `step` as a
                                       double (:862): grid ends lrint((W - w) / step) in f64 (:890-891) and
                                       positions from the f64 product — lrint in the row loop (:941-942),
                                       round() in the per-stage lists (:1034): a third and a fourth grid       */
    VJ_FLAG_TILTED_AS_UPRIGHT = 1u << 5, /* a cascade with <tilted>1 features in the clod profile: precomputeFeatures
                                       never reads the flag (clod.cpp:448-492 takes haar_feature[0]'s rectangles as they are),
                                       so the reference evaluates such rectangles as UPRIGHT ones — 12 of the 19 cascades
                                       it ships have them.  Without this flag the clod-profile entry points refuse the
                                       cascade (VJ_ERR_UNSUPPORTED: the result is not a meaningful detection); with it they
                                       reproduce the reference's arithmetic.  vj_detect_opencv evaluates tilted features on
                                       the tilted integral as OpenCV does and ignores the flag.                            */
};

typedef struct vj_params {
    int32_t  min_w, min_h;     /* min_window_size (0 = none)                    */
    int32_t  max_w, max_h;     /* max_window_size (0 = unlimited, clod.cpp:394) */
    float    scale_factor;     /* reference hard-codes 1.1f (clod.cpp:1184)     */
    uint32_t min_neighbors;    /* 0 = raw candidates (the parity contract); else grouped */
    uint32_t flags;
    uint64_t scale_mask[2];    /* bit k set = evaluate scale index k (k < 127); both
                                  words 0 = every scale; bit 127 (VJ_SCALE_MASK_NONE
                                  in word 1) = no scale at all: an empty share, the
                                  call returns no rectangles.  Shards one frame's
                                  scales across GPUs; no counterpart in the reference. */
} vj_params;
#define VJ_SCALE_MASK_NONE (1ull << 63)   /* in scale_mask[1]: what vj_shard_scales gives a rank that gets no scale */
void vj_params_default(vj_params* p);   /* {0,0,0,0,1.1f,0,0,{0,0}} */

/* ------------------------------------------------------ host scale planning */
/* setupScale + scale enumeration (clod.cpp:371-415, 1198-1204). */
typedef struct vj_scale_info {
    int32_t  scale_idx;        /* k in s_k = fl32(s_{k-1} * scale_factor)       */
    float    scale;            /* s_k                                           */
    float    step;             /* (float)MAX(2.0, s)                            */
    int32_t  win_w, win_h;     /* scaled_window_size                            */
    int32_t  equ_x, equ_y, equ_w, equ_h;  /* equ_rect                           */
    uint32_t area;             /* scaled_window_area                            */
    int32_t  nx, ny;           /* end_point: window grid is [0,nx) x [0,ny)     */
    int32_t  accepted;         /* 0 when setupScale returned -1                 */
} vj_scale_info;
/* Writes up to cap entries (all enumerated scales, accepted or not); *n = count. */
int vj_plan_scales(const vj_cascade* c, int width, int height, const vj_params* p,
                   vj_scale_info* out, int cap, int* n);

/* precomputeKernelCascade (clod.cpp:529-578) for one scale: per node, 3 rects of
 * {left_top, right_top, left_bottom, right_bottom element offsets, weight}.
 * `offsets` receives n_nodes*12 uint32, `weights` n_nodes*3 floats.            */
int vj_plan_feature_table(const vj_cascade* c, int width, const vj_scale_info* s,
                          uint32_t* offsets, float* weights);

/* ------------------------------------------------------------- environment */
/* clodInitEnvironment/clodReleaseEnvironment (clod.h:61-65, clod.cpp:72-100,
 * 173-180) — one env per device; not thread-safe (neither is the reference).   */
typedef struct vj_env vj_env;
int  vj_env_create(int device_index, vj_env** out);
void vj_env_destroy(vj_env* e);
/* clodInitBuffers + clifInitBuffers (clod.cpp:102-163, clif.cpp:105-224):
 * pre-size device buffers; optional — vj_detect grows them on demand.          */
int  vj_env_reserve(vj_env* e, int max_w, int max_h, int max_batch);
int  vj_env_device_name(const vj_env* e, char* buf, size_t cap);
/* Tunables — speed only: results never depend on them (tests sweep every group).  One line per group here; every key
 * with its values, default and the measurement behind the default is in DESIGN.md §7.
 *   launch structure   pass_split, pass_cut_nodes, blocks_per_cu, concurrent, concurrent_blocks_per_cu, max_subbatch, det_cap
 *   LDS tiles          tile_classes_kb, tile_lds_reserve_kb, tile_min_windows, tile_accept_windows, tile_end, tile_min_lanes,
 *                      tile_max_dwords_per_window, tile_class_order, tile_lds_nest, tile_repack, tile_deinterleave, tile_stage_x4,
 *                      global_blocks
 *   tile finish        tile_finish, tile_sp_begin, tile_sp_max, tile_ws_min, tile_ws_max
 *   chain balance      tile_split ("small,mid,large" or one value: static), auto_balance (1 / 0 / "reset": feedback on the
 *                      first calls of a batch workload, keyed by cascade content, frame size, parameters and batch-size class),
 *                      balance_export / balance_import (value: a file path; the found balances as text, for another environment
 *                      or process — import AFTER setting auto_balance, which clears the table), balance_exact (tests)
 *   global-gather      grid_block_w, gather_waves, gather_pairs, sp_tail_max, wide_tail, min_chunk, thin_pass_spread, q_slices,
 *                      xcd_affinity, q_band_px, q_group_units, q_band_min_frames (band-major first-pass units and queue pass)
 *   stage trees        general_prefix, tile_segments, seg_cut2, tree_split_queues
 *   regions / chain    rois_on_device, roi_tiles, group_max
 *   OpenCV profile     cv_tiles, cv_row_blocks, cv_tile_min_windows, cv_tile_min_windows0, cv_tile_ws_max, cv_row_blocks_tree,
 *                      cv_tile_min_windows_tree, cv_tree_chains, cv_tree_chunk, cv_tree_chain_blocks, cv_tail_max, cv_pairs,
 *                      cv_row_band_px, cv_tree2, cv_tiles_tilted, tilted_bands,
 *                      cv_tree_queue_cap (tests)
 *   single frames      one_pass_max_frames (the gather chain in one pass for calls of few large frames; 0 = off)
 *   integral           integral_rows (0 one wave per band of rows, 1 a band's chunks side by side, 2 by call size)
 *   housekeeping       plan_cache_max
 * Unknown keys return VJ_ERR_ARG.                                               */
int  vj_env_configure(vj_env* e, const char* key, const char* value);

/* --------------------------------------------------------------- integral */
/* clifIntegral (clif.h:63-66, clif.cpp:273-285 → cvIntegral layout):
 * sum u32 and sqsum u64, both (h+1) x (w+1) row-major, row 0 / col 0 zero.
 * `gray`, `sum`, `sqsum` are HOST pointers (the reference returns host CvMat).  */
int vj_integral(vj_env* e, const uint8_t* gray, int w, int h, int stride,
                uint32_t* sum, uint64_t* sqsum);

/* Page-locked host memory for frames (what clodInitBuffers / clifInitBuffers pre-allocate in the reference,
 * clod.cpp:102-163): frames that live in it are uploaded by DMA without a staging copy.                  */
int  vj_host_alloc(vj_env* e, size_t bytes, void** out);
void vj_host_free(vj_env* e, void* p);

/* ----------------------------------------------------------------- detect */
struct vj_image;
/* clifGrayscaleIntegral (clif.h:67-70, clif.cpp:326-335): gray conversion + both integrals of one image
 * (host or device pointer, 1 / 3 / 4 channels); outputs as vj_integral.                       */
int  vj_integral_image(vj_env* e, const struct vj_image* image, uint32_t* sum, uint64_t* sqsum);
/* clifGrayscale (clif.h:55-58, clif.cpp:226-271 -> cvCvtColor BGR2GRAY): the 8-bit gray image the integral
 * kernels see, written to the HOST buffer `gray` (gray_stride bytes per row).  1-channel input is copied. */
int  vj_grayscale(vj_env* e, const struct vj_image* image, uint8_t* gray, int gray_stride);
/* The tilted integral cvIntegral(img, sum, sqsum, tilted) returns for cascades with tilted features
 * (tempcv.cpp:1335, :743-750): (h+1) x (w+1) u32, tilted(X, Y) = sum of gray(x, y) over y < Y,
 * |x - X + 1| <= Y - y - 1.  HOST output.                                                              */
int  vj_integral_tilted(vj_env* e, const struct vj_image* image, uint32_t* tilted);
typedef struct vj_image {
    const uint8_t* data;       /* 8-bit, interleaved channels                   */
    int32_t width, height;
    int32_t stride;            /* bytes per row                                 */
    int32_t on_device;         /* 0: host pointer; 1: device pointer on env's GPU */
    int32_t channels;          /* 0 or 1: gray (the configs' contract); 3: BGR, 4: BGRA — converted on the
                                  fly with OpenCV's 8-bit BGR2GRAY, as setupImage does (clif.cpp:326-335) */
} vj_image;

typedef struct vj_rect {       /* CLODWeightedRect (clod.h:39-42) + provenance  */
    int32_t x, y, w, h;
    float   weight;            /* reference leaves it unset/0 for raw results   */
    int32_t frame;
    int32_t scale_idx;
} vj_rect;

#define VJ_MAX_STAGES 64
typedef struct vj_counters {
    uint64_t windows;          /* candidate windows enumerated                  */
    uint64_t stump_evals;      /* tree-node evaluations as SURVEY.md §8d counts them: the nodes a window's walk visits
                                  (every node of an entered stage for stumps; root + visited children for trees) */
    uint64_t gather_bytes;     /* 48*windows + 16*sum(nrects)   */
    uint64_t stage_entered[VJ_MAX_STAGES]; /* windows entering each stage       */
} vj_counters;

#define VJ_MAX_PASSES 8
#define VJ_MAX_LAUNCHES 16
enum { VJ_LAUNCH_GRID = 0,   /* first pass, windows enumerated from the grid, L2 gathers */
       VJ_LAUNCH_QUEUE = 1,  /* later pass over the survivor queue, L2 gathers           */
       VJ_LAUNCH_TILE = 2,   /* whole cascade on image tiles staged in LDS               */
       VJ_LAUNCH_BLOCK = 3 };/* whole cascade on 2-D window blocks, L2 gathers (large scales) */
typedef struct vj_launch {
    int32_t  kind;             /* VJ_LAUNCH_*                                   */
    int32_t  lds_class;        /* tile launches: LDS size class                 */
    int32_t  stage_begin, stage_end;  /* stages it may run (tile launches: up to stage_end) */
    float    ms;               /* HIP-event time, summed over sub-batches       */
    uint32_t lds_bytes;
    uint64_t scale_mask[2];    /* scale indices it covers, wholly or in part (queue passes: all) */
    uint64_t stage_entered[VJ_MAX_STAGES];  /* VJ_FLAG_COUNTERS: windows this launch took into each stage */
} vj_launch;
typedef struct vj_timing {     /* HIP-event times of the last vj_detect, ms     */
    float integral_ms;         /* the three integral launches                   */
    float cascade_ms;          /* all cascade passes                            */
    float total_ms;            /* first kernel start → last kernel end          */
    int32_t n_cascade_launches;
    float pass_ms[VJ_MAX_PASSES];            /* each cascade pass (its launches) */
    int32_t pass_stage_begin[VJ_MAX_PASSES]; /* stages [begin, end) it ran      */
    int32_t pass_stage_end[VJ_MAX_PASSES];
    int32_t n_launches;                      /* kernel launches of the cascade  */
    vj_launch launch[VJ_MAX_LAUNCHES];       /* each with its own HIP events    */
    float tile_split;          /* vj_detect: scales' worth of tile work the plan of this call gave to the global-gather
                                  chain ("tile_split"; found per workload by feedback unless configured) */
    int32_t balance_state;     /* 0: static balance (no feedback for this call); 1: the workload's feedback search is still
                                  running; 2: finished — the workload runs its best split from now on */
    int32_t balance_calls;     /* calls the search has measured so far for this workload (at most 40) */
} vj_timing;

typedef struct vj_result {
    vj_rect*   rects;          /* sorted by (frame, scale_idx, y, x)            */
    uint32_t   count;
    vj_counters counters;      /* valid when VJ_FLAG_COUNTERS                   */
    vj_timing  timing;
} vj_result;

/* clodDetectObjects(image, cascade, data, min, max, min_neighbors, flags, CL_TRUE)
 * (clod.h:72-81, clod.cpp:1176-1336), batched over n_frames frames of equal size. */
int  vj_detect(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames,
               const vj_params* p, vj_result* out);
void vj_result_free(vj_result* r);

/* ------------------------------------------------ OpenCV arithmetic profile */
/* cvHaarDetectObjects(image, cascade, storage, scale_factor, min_neighbors, flags = 0, min_size)
 * (call site main.cpp:145; scale-cascade path as tempcv.cpp:1188-1456 keeps it, scalar branches —
 * CV_HAAR_USE_SSE is commented out at :28-36): f64 variance and stage sums; node sums as
 * cvRunHaarClassifierCascadeSum writes them — an f64 product per rectangle in stump stages flagged
 * two_rects (:872-888), otherwise int * float, i.e. a BINARY32 product widened to double before it is
 * accumulated (:907-911, icvEvalHidHaarClassifier :783-788); stage threshold - 0.0001f; cvRound-ed
 * rectangles and grid; ystep = max(2, factor); the skip after a reject (stage 0 for linear cascades, any
 * stage for stage trees, which return 0 on every reject: :834-861, :1163); the window-touches-border
 * rule; stage trees; tilted features on the tilted integral (:731, :743-750).  Raw candidates
 * (min_neighbors = 0) or cv::groupRectangles.  counters.windows = positions the sequential walk
 * visits.  Parity: against the oracle's restatement of the same lines — OpenCV itself cannot be run
 * here (unpinned).                                                                                */
typedef struct vj_cv_params {
    int32_t  min_w, min_h;     /* minSize (0 = none)                              */
    double   scale_factor;     /* 1.1                                             */
    uint32_t min_neighbors;
    uint32_t flags;            /* VJ_FLAG_COUNTERS                                */
} vj_cv_params;
void vj_cv_params_default(vj_cv_params* p);
int  vj_detect_opencv(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames,
                      const vj_cv_params* p, vj_result* out);

/* A second cascade on regions of interest (BASELINE config 5: haarcascade_eye inside every face;
 * the reference's caller would hand clodDetectObjects a sub-image header: pointer + widthStep).
 * ROIs are views into `frames`.  In the result, rect.frame is the ROI's index and x / y are relative
 * to the ROI's origin.  Frames of one size: the frames' integral images are computed once and ALL
 * regions, of whatever sizes, run in one pass on them (a rectangle sum does not depend on where the
 * integral image starts; vj_detect_chain's region pass with an uploaded list; stage trees and scale
 * masks included).  Otherwise (frames of different sizes, skip modes): one vj_detect call per region
 * size on the sub-images.  Same result either way.                                               */
typedef struct vj_roi { int32_t frame, x, y, w, h; } vj_roi;
int  vj_detect_rois(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames,
                    const vj_roi* rois, int n_rois, const vj_params* p, vj_result* out);

/* Two cascades back to back with the hand-off on the device (BASELINE config 5): `first`
 * runs on the frames as vj_detect does; what it finds becomes a DEVICE-resident list of regions of interest
 * (built by kernels from the detection buffer), and `second` runs on those regions reading the frames'
 * integral images in place — rectangle sums over a region do not depend on where the integral image starts,
 * so the result equals running `second` on the sub-image (vj_detect_rois) — before anything returns to the
 * host.  p_first->min_neighbors == 0: every raw candidate is a region.  != 0: the candidates are grouped ON
 * THE DEVICE (cv::groupRectangles per frame, as vj_detect groups them on the host: same classes, same
 * averages, same order) and the grouped faces are the regions.  out_first: as vj_detect with the same
 * parameters.  out_second: rect.frame = index of the region in out_first->rects, x / y relative to the
 * region's origin.  `second` may be any cascade of upright features (stumps, trees, stage trees);
 * p_second->scale_mask selects scales by index as in vj_detect.  With a skip mode (VJ_FLAG_SKIP_ROW / _LIST) on
 * either cascade the hand-off goes through the host: vj_detect, then vj_detect_rois on the sub-images.     */
int  vj_detect_chain(vj_env* e, const vj_cascade* first, const vj_cascade* second, const vj_image* frames,
                     int n_frames, const vj_params* p_first, const vj_params* p_second, vj_result* out_first,
                     vj_result* out_second);

/* ------------------------------------------------------------ frame streams */
/* Video-style use (the demo's per-frame loop, main.cpp:104-125): batches of host frames are uploaded into
 * one of two device buffers by DMA on a copy stream while the kernels of the previous batch run, and the
 * results come back one submit later.  submit() returns once the batch is queued (it blocks only while
 * both buffers are busy); collect() returns the oldest submitted batch (VJ_ERR_ARG when none is pending).
 * Frames must stay valid until their batch is collected; put them in vj_host_alloc memory for a copy-free
 * upload (pageable memory works, through a staging copy).  Results equal vj_detect's.                   */
typedef struct vj_stream vj_stream;
int  vj_stream_create(vj_env* e, const vj_cascade* c, int width, int height, int channels, int max_batch,
                      const vj_params* p, vj_stream** out);
int  vj_stream_submit(vj_stream* s, const vj_image* frames, int n_frames);
int  vj_stream_collect(vj_stream* s, vj_result* out);
void vj_stream_destroy(vj_stream* s);

/* filterResult (clod.cpp:182-357) as cv::groupRectangles defines it (tempcv.cpp:130-243): groups
 * `rects` (sorted by frame; grouped per frame, in place), keeps classes with more than
 * group_threshold members, weight = members, scale_idx = -1.  vj_detect applies it with
 * MAX(min_neighbors, 1) and eps 0.2 (clod.cpp:11, 1326) when min_neighbors != 0.            */
int vj_group_rectangles(vj_rect* rects, uint32_t* count, int group_threshold, double eps);

/* ---------------------------------------------------------------- multi-GPU */
/* One environment per device (clodInitEnvironment(device_index), clod.cpp:72-100), one rank per environment — threads of
 * one C++ host or processes.  The path shards without a data-path exchange: (frame, scale) pairs are
 * independent given a frame's integral images.  These two host helpers give every rank its share; the only collective is
 * the final all-gather of rectangles (include/vj_rccl.h: header-only, RCCL's ncclAllGather, so that this library itself
 * does not link librccl).
 * vj_shard_frames: batches with at least as many frames as ranks — contiguous blocks whose sizes differ by at most one.
 * vj_shard_scales: fewer frames than ranks (one large frame) — every rank integrates the frame and takes a subset of the
 * scales, longest-processing-time greedy on an estimated cost (windows x a per-window weight; the LDS-tile scales and the
 * global-gather scales are dealt separately so that each rank keeps both of its chains busy; integers only, ties to the
 * lower rank); the result goes into vj_params.scale_mask.  A rank that gets no scale (more ranks than scales) receives VJ_SCALE_MASK_NONE — an all-zero
 * mask would mean "every scale" and duplicate the other ranks' rectangles.                                                                                            */
int vj_shard_frames(int n_frames, int n_ranks, int rank, int* first, int* count);
int vj_shard_scales(const vj_cascade* c, int width, int height, const vj_params* p, int n_ranks, int rank,
                    uint64_t scale_mask[2]);

/* Candidate windows per frame for (cascade, size, params): sum of nx*ny over
 * accepted scales — the denominator of the windows/s metric.                   */
int vj_count_windows(const vj_cascade* c, int width, int height, const vj_params* p,
                     uint64_t* out);

#if defined(VJ_BUILDING) && defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VJ_H_ */
