// Internals of the host driver shared by its translation units (vj_env.cpp: the clod arithmetic profile;
// vj_cv.cpp: the OpenCV arithmetic profile).  Not part of the C ABI.
#pragma once
#include "vj_internal.hpp"
#include "vj_device.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

namespace vj {

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return VJ_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return VJ_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        // (a buffer never shrinks: alternating sizes reallocate only when a request exceeds every earlier one)
        size_t want = std::max(bytes, (size_t)256);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return e == hipErrorOutOfMemory ? VJ_ERR_NOMEM : VJ_ERR_HIP;
        }
        cap = want;
        return VJ_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Everything that depends on (cascade, W, H, params) but not on pixel data.
struct Plan {
    std::vector<vj_scale_info> scales_all;   // every enumerated scale
    std::vector<ScaleDev> scales;            // accepted scales with nwin > 0
    std::vector<vj_scale_info> scales_info;  // same order as `scales`
    std::vector<StageDev> stages;
    std::vector<UnitDev> units;              // first-pass units of one frame (global-gather scales), band-major (q_band_px)
    std::vector<UnitDev> unit_groups;        // runs of consecutive units of one (band, scale): what a wave of the band-major queue pass draws
    DevBuf d_unit_groups;
    std::vector<UnitDev> tile_units;         // first-pass tiles of one frame, grouped by LDS class
    uint32_t class_first[TILE_CLASSES + 1] = {};  // tile_units range of each class
    uint32_t class_lds[TILE_CLASSES] = {};        // dynamic LDS bytes of each class launch
    uint32_t block_first = 0, n_block_units = 0;  // tile_units range of the unstaged 2-D blocks (global-gather scales)
    uint32_t block_lds = 0;
    uint32_t tile_end = 0;                        // stage at which the tile launches stop
    bool tree2 = false;                           // every tree: two nodes, node 1 the only node child of node 0
    uint32_t sp_pad = 0;                          // LDS pitch of the stump-parallel stage table (0 = off)
    std::vector<uint32_t> pass_bounds;       // stage indices: pass p runs [b[p], b[p+1])
    uint64_t windows_per_frame = 0;
    uint32_t frame_elems = 0;
    uint32_t max_reach_elems = 0;  // furthest element a window origin + feature corner touches
    bool trees = false;    // some tree has more than one node
    bool general = false;  // stage tree (not a linear chain of stages)
    uint32_t general_prefix = 0;  // stage tree: positions [0, prefix) of the sweep order form a linear chain (pass -> next,
                                  // fail -> reject) that runs on the linear kernels (tiles included)
    // Stage tree cut into linear segments (chains of the sweep order whose rejects all go to one place): pass ps reads
    // queue ps; seg_last[ps]: its survivors are detections; seg_fail[ps]: queue that takes its rejects (0 = none).
    // Empty when the tree does not have that shape: then one general pass (run_stages_general) finishes it.
    std::vector<uint8_t> seg_last, seg_fail;
    // the same chains for the tile kernel, which runs them itself when chain k's rejects feed chain k+1
    uint32_t tile_n_seg = 0, tile_seg_end[4] = {}, tile_seg_chain = 0;
    uint32_t n_order = 0;  // stages reachable from stage 0, in StageDev::order
    uint32_t max_stage_nodes = 0;
    StageProgram prog;
    // device copies
    DevBuf d_table, d_scales, d_stages, d_units, d_tile_units, d_sp_blocks, d_skip_units, d_skip_segs, d_pos_tab;
    // P2 skip modes (VJ_FLAG_SKIP_LIST / VJ_FLAG_SKIP_ROW): bitmap geometry and work lists of the two bitmap kernels
    uint32_t skip_mode = 0, pos_mode = 0, skip_frame_words = 0, n_skip_units = 0, n_skip_segs = 0;
    uint32_t n_sp_blocks = 0;
    int frames_q = 0;  // frames the ScaleDev.q_base/q_cap currently describe
    uint64_t last_used = 0;   // vj_env::plan_tick of the last call that used this plan (LRU eviction)
    float tile_split = 0.0f;  // the chain balance this plan was built for (vj_env::split_for)
    void release_device() {
        for (DevBuf* b : {&d_table, &d_scales, &d_stages, &d_units, &d_tile_units, &d_sp_blocks, &d_skip_units, &d_skip_segs, &d_pos_tab, &d_unit_groups}) b->release();
    }
};

// What vj_detect_opencv derives from (cascade, frame size, parameters): kept per environment like the clod plans, so that a
// caller that hands over one frame per call (main.cpp:145) does not rebuild and upload 42 feature tables every time.
struct CvPlan {
    std::vector<CvScaleDev> scales;   // host copy (window sizes and scale indices of the result)
    StageProgram prog;
    uint32_t n_stages = 0, n_order = 0, n_rows = 0;
    bool trees = false, is_tree = false, has_tilted = false, tree2 = false;
    DevBuf d_table, d_scales, d_stages, d_rows;
    // LDS-tile path (vj_cv_tile.hip): tiles of one frame per LDS class, the rows of the scales that stay on
    // cv_profile_pass, and the per-frame reject / visited bitmap with one recurrence domain per window row
    uint32_t n_tile_scales = 0, n_rows_rest = 0, bits_frame_words = 0, n_bit_segs = 0;
    int row_blocks = 1;               // workgroups per CU of cv_profile_pass next to the tiles
    uint32_t tree_prefix = 0;         // stage trees: stages of the linear prefix the tiles run (0: the cascade is linear)
    CvChainDev chains = {};           // stage trees: the part after the prefix as chains (n = 0: not of that shape)
    int tq_shift = -1;                // stage trees on tiles: the prefix survivors' sub-queues hold 1 / 2^shift of the tile windows
                                      // (-1: the environment's start value; lowered for THIS plan when a sub-queue overflows)
    uint64_t tile_windows = 0;        // grid windows of the tile scales, per frame
    uint32_t class_first[3] = {0, 0, 0}, class_lds[2] = {0, 0};
    DevBuf d_tiles, d_rows_rest, d_bit_segs;
    uint64_t last_used = 0;
    void release_device() {
        for (DevBuf* b : {&d_table, &d_scales, &d_stages, &d_rows, &d_tiles, &d_rows_rest, &d_bit_segs}) b->release();
    }
};

}  // namespace vj

namespace vj {

// Everything ONE batch in flight owns: its frames on the device, its counters / detections and their host copies, and
// the events that time it.  vj_detect uses the environment's own lane; a vj_stream owns two, so that the upload of
// batch k+1 can run while the kernels of batch k do.  Integral images, survivor queues and plans are shared: the
// kernels of successive batches are ordered on the environment's stream.
// (vj_cv_profile.hip) the tilted integral as three banded prefix sums; diag: n_frames x bands x 2 x (W + H) dwords, col: n_frames x bands x (W + 1)
int launch_tilted_bands(const TiltedArgs& a, uint32_t* diag, uint32_t* col, void* stream);

struct Lane {
    DevBuf d_gray, d_counts, d_det;
    uint32_t det_cap = 0;
    void* h_pinned = nullptr;        // counters block + the first detections, read back asynchronously
    size_t h_pinned_bytes = 0;
    void* h_stage = nullptr;         // pinned staging of pageable host frames (streams)
    size_t h_stage_bytes = 0;
    hipEvent_t ev[4] = {};
    hipEvent_t pass_ev[VJ_MAX_PASSES + 1] = {};
    hipEvent_t launch_ev[2 * VJ_MAX_LAUNCHES] = {};   // start/stop per launch
    hipEvent_t upload_done = nullptr, integral_done = nullptr, done = nullptr;
    // the batch in flight (enqueue_batch -> finish_batch)
    bool pending = false;
    int nf = 0, first_frame = 0;
    // where the integral kernels read this batch's frames (the lane's d_gray or the caller's device batch), so that a
    // redo can recompute them: the lanes of a vj_stream share the environment's integral images, and by the time a
    // batch turns out to have overflowed its detection buffer the next batch's integrals have replaced its own
    const uint8_t* src_gray = nullptr;
    size_t src_frame_bytes = 0;
    int src_stride = 0, src_channels = 1;
    bool shared_integrals = false;
    size_t n_pass = 0;
    int launches = 0;
    bool count = false;
    uint32_t det_copied = 0;         // detections already copied to h_pinned by the asynchronous read-back
    std::vector<vj_launch> linfo;
    int create();
    void destroy();
};

}  // namespace vj

struct vj_env {
    int device = 0;
    hipStream_t stream = nullptr;
    vj::Lane lane0;                  // the batch of a plain vj_detect call
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    hipStream_t stream2 = nullptr;   // second chain of the first part of the cascade
    int max_subbatch = 0;      // > 0: cap on frames per sub-batch (tests)
    uint32_t det_cap_init = 1u << 16;  // initial capacity of the detection buffer (grows on overflow)
    int concurrent = 1;   // 1: the tile chain and the global-gather chain overlap on two streams
    int gather_waves = -1;              // waves per workgroup of the global-gather kernels (-1: 4 for calls of >= 8 frames, else 3)
    int gather_waves_for(int n_frames, bool general) const {
        if (general) return vj::WAVES_PER_BLOCK;
        const int w = gather_waves > 0 ? gather_waves : (n_frames >= 8 ? vj::GATHER_WAVES_MAX : vj::WAVES_PER_BLOCK);
        return std::min(std::max(w, 1), (int)vj::GATHER_WAVES_MAX);
    }
    int concurrent_blocks_per_cu = 1;   // workgroups per CU of the global-gather chain while it overlaps
    // scales' worth of tile work handed to the global-gather chain (largest tile scales first), by batch size: a single
    // frame is bound by the latency of the gather chain's thin queue pass (measured, 1080p / frontalface_alt: 1 frame
    // 1.20 / 1.46 / 1.46 ms at split 0 / 0.5 / 1.25; 4 frames 3.50 / 3.61 / 3.96; 16 frames 12.51 / 12.25 / 12.36;
    // 64 frames — / 48.0 / 54.0), so small batches keep everything they can on the tiles
    float tile_split = 2.0f;            // batches of >= 32 frames (round 4, band-major queue pass: 64 x 1080p 43.6 / 43.0 / 42.5 / 43.4 / 44.5 ms for 1.5 / 1.75 / 2 / 2.25 / 2.5; round 3: (four gather waves; 64 x 1080p: 46.70 / 46.24 / 45.57 / 45.16 / 45.55 / 46.72 ms for 0.75 / 1 / 1.25 / 1.5 / 1.75 / 2; 32: 23.39 / 23.16 / 22.80 / 22.62 / 22.78 / 23.38)
    float tile_split_mid = 1.75f;       // 8 .. 31 frames (round 4, band-major queue pass: 16 x 1080p 11.53 / 11.36 / 11.13 / 11.01 / 10.89 / 10.90 ms for 0.75 ... 2.0; round 3: (16 x 1080p: 11.81 / 11.67 / 11.47 / 11.52 / 11.86 for 0.75 ... 1.75; 8: 6.03 / 5.95 / 5.97 / 6.24); 5 .. 7 frames (three gather waves): at most 0.5
    float tile_split_small = 0.0f;      // <= 4 frames
    int one_pass_max_frames = 0;        // calls of at most this many frames (of 720p and more, stump cascades) run the gather chain in ONE pass; 0: never.
                                        // Content decides which is faster (profiles/r04_notes.md #15: noise -6 %, drawn faces +1-7 %), so it is off by default
    float split_for(int n_frames) const {
        return n_frames <= 4 ? tile_split_small : n_frames < 8 ? std::min(tile_split_mid, 0.5f) : n_frames < 32 ? tile_split_mid : tile_split;
    }
    // (the defaults are fractions of the last tile scale of a WHOLE pyramid; a share of the scales (vj_shard_scales) may end with a
    // scale of half a million windows: it starts without a move and lets the feedback find one)
    float split_for(int n_frames, const vj_params& p) const {
        return ((p.scale_mask[0] | p.scale_mask[1]) != 0 && !tile_split_set) ? 0.0f : split_for(n_frames);
    }
    int xcd_affinity = 1;               // global-gather first pass: one contiguous part of the work per XCD (L2 locality)
    int tile_segments = 1;              // stage trees: tiles run the chains after the prefix themselves
    int seg_cut2 = 0;                   // stage trees: a second cut inside a long chain after this many of its stages (0: none)
    int general_prefix = 1;             // stage trees: run their linear prefix on the linear kernels (0: one general pass)
    int grid_block_w = 32;              // width of the 2-D window blocks of the global-gather first pass (0: row runs)
    int global_blocks = 0;              // 1: large scales run as unstaged 2-D blocks in the tile kernel (stump cascades): 2.2x
                                        // faster than grid + queue passes on its own, but it overlaps the tile chain badly
    int tile_lds_reserve_kb = 16;       // LDS per CU the tile classes leave to the other chain (its 3-wave workgroup: 12 KiB + granule
                                        // rounding; with 14 the CU's 160 KiB do not take two class-0 blocks next to it any more)
    char name[256] = "";
    int n_cu = 0;
    // image buffers
    vj::DevBuf d_sum, d_sqsum, d_band_sum, d_band_sq, d_band_sqp;
    vj::DevBuf d_tilted;            // tilted integral images (OpenCV profile, cascades with tilted features)
    vj::DevBuf d_tilt_diag, d_tilt_col;   // ... its band totals (launch_tilted_bands)
    bool tilted_bands = true;       // the tilted integral as three banded prefix sums (0: the row-by-row recurrence, one workgroup per frame)
    vj::DevBuf d_out;               // scratch for device -> host results (vj_grayscale)
    int slack_w = 0, slack_h = 0, slack_frames = 0;   // layout whose slack rows are known to be zero
    void *slack_sum = nullptr, *slack_sq = nullptr;
    // survivor queues + counters + detections
    vj::DevBuf d_q[vj::MAX_PASSES];   // d_q[p]: windows waiting to enter pass p (p >= 1)
    vj::DevBuf d_q2[vj::MAX_PASSES];  // stage trees: the tiles' own queue set (enqueue_cascade: split_sets)
    vj::DevBuf d_skip_bits;           // P2 skip modes: visited-window bitmaps of the frames in flight
    vj::DevBuf d_run_table;           // band-major queue pass: where every first-pass unit's survivors sit in their sub-queue
    vj::DevBuf d_rois, d_roi_units, d_roi_det, d_roi_tiles;   // regions of interest on the device (vj_detect_chain)
    uint32_t roi_tile_cap = 0;
    int roi_tile_min_windows = 512;   // region pass: (region, scale) grids of at least this many windows run on LDS tiles (0: never)
    vj::DevBuf d_group;                          // scratch of the device-side grouping (vj_detect_chain, min_neighbors != 0)
    uint32_t roi_unit_cap = 0, roi_det_cap = 0;
    typedef std::tuple<uint64_t, int, int, int, int, int, int, uint32_t, uint64_t, uint64_t, uint32_t, uint32_t> PlanKey;
    std::map<PlanKey, std::unique_ptr<vj::Plan>> plans;
    // Chain balance per workload (cascade, frame size, parameters, batch-size CLASS): how much tile work goes to the
    // global-gather chain (Plan::tile_split) is found by a short hill climb on the measured cascade time of the workload's
    // first calls and then frozen; vj_env_configure("tile_split", ...) or ("auto_balance", "0") keep the static values.
    // The key names the cascade by CONTENT (two loads of one file share an entry; an exported table fits another process)
    // and the batch size by class (8-15, 16-31, 32-63, >= 64 frames; below 8 — single large frames — the exact count): a
    // service whose batch sizes vary searches four times, not once per size.  Only calls of the class's reference size
    // (n_ref: the first size seen, re-anchored when it stops coming) run candidates and feed the search — times are compared
    // per frame of ONE size —; every other call of the class runs the best split found so far.
    struct Balance {
        float cur = 0, best = 0, best_ms = 0, cand_ms = 0;   // (times: ms per frame)
        int phase = 0;        // 0: measuring the start value, 1: climbing up, 2: climbing down, 3: frozen, 4: measuring the other tile thresholds, 5: probing a whole scale further
        int samples = 0, moved = 0, calls = 0;
        int thr = 0;          // 0: the environment's tile thresholds; 1: scales whose tiles hold >= 384 windows go to tiles too
        bool thr_tried = false, far_tried = false;
        int n_ref = 0;        // frames per call of the calls that sample
        int off_ref = 0;      // calls of other sizes since n_ref was last seen
        bool first_slow = false;   // the candidate's unrated first call was already > 3 % slower than the best
        uint64_t last_used = 0;    // (least recently used entries go first when the table is full)
        uint32_t calls_total = 0, calls_on_candidate = 0;   // every call of the workload / those that ran a split other than the best known
    };
    typedef std::tuple<PlanKey, int> BalanceKey;    // the plan key with the cascade's content hash and split = 0, batch-size class
    std::map<BalanceKey, Balance> balance;
    uint64_t balance_tick = 0;
    bool balance_exact = false;   // key on the exact frame count, as round 3 did ("balance_exact": for the before / after of tools/balance_service.py)
    int balance_class(int n_frames) const {
        return balance_exact || n_frames < 8 ? n_frames : n_frames < 16 ? 8 : n_frames < 32 ? 16 : n_frames < 64 ? 32 : 64;
    }
    bool auto_balance = true, tile_split_set = false;
    typedef std::tuple<uint64_t, int, int, int, int, uint64_t, int> CvPlanKey;   // cascade uid, W, H, min size, bits of the scale factor, call of <= 4 frames
    std::map<CvPlanKey, std::unique_ptr<vj::CvPlan>> cv_plans;
    vj::DevBuf d_cv_det, d_cv_counts;   // vj_detect_opencv: detection list and counters
    vj::DevBuf d_cv_accept, d_cv_tq;    // ... stage trees on tiles: accept bitmap, the queue of the prefix's survivors
    vj::DevBuf d_cv_fail_rows, d_cv_fail_walk;   // ... per-wave fail lists of the chain sweeps (rows kernel / chain pass)
    uint64_t plan_tick = 0;
    int plan_cache_max = 48;      // plans kept per environment; the least recently used one is released beyond that
                                  // (a stream of ROI sizes — eyes inside faces of any size — would otherwise grow
                                  // device tables without bound)
    // tunables (env vars, read once)
    int integral_rows_mode = 2;      // band_rows: 0 one wave walks a band's chunks, 1 the chunks of a band side by side, 2 side by side except for batches
    int blocks_per_cu = 8;
    int tile_class_kb[vj::TILE_CLASSES] = {-2, -1, 0};  // image-tile LDS budget per class in KiB; -k = what lets k
                                                     // workgroups share a CU's 160 KiB; all 0 disables the tile path
    int tile_min_windows = 768;   // a class is acceptable for a scale when a tile holds at least this many windows
    int tile_max_dwords_per_window = 600;  // staging a tile must stay far cheaper than gathering its windows from L2
    bool tile_thresholds_set = false;   // the three thresholds were configured: no per-frame-size defaults (get_plan)
    int tile_accept_windows = 768;  // scales whose best tile holds fewer windows stay on the global-gather path
    int tile_end = 64;            // tile launches never enter a pass that begins at or beyond this stage
    int tile_min_lanes = 0;       // a tile leaves at a pass boundary when fewer windows than this survive in it
    unsigned long long tile_repack_mask = ~3ull;  // stages (2 and later) before which a tile re-packs its survivors
    int tile_sp_begin = 3;        // first stage at which a tile may switch to the stump-parallel finish (>= 64: never)
    int tile_sp_max = 192;        // ... once at most this many of its windows survive
    int tile_finish = 1;          // 0: stump-parallel finish, 1: wave-split finish
    int tile_ws_max = 512;        // windows a tile may carry into the wave-split finish
    int tile_ws_min = 48;         // ... below this many the stump-parallel finish takes over
    int tile_class_order = 1;     // 1: launch the tile classes largest-LDS first (measured: 48.3 -> 47.2 ms; the one-workgroup-per-CU
                                  // class suffers most from the gather chain, whose first pass is the heavier one)
    int tile_lds_nest = 1;        // LDS blocks of consecutive tile classes nest (k blocks of one = one block of the next)
    int tile_stage_x4 = 1;        // stage tile rows with 16-byte LDS-DMA loads (4x fewer texture-address instructions)
    int tile_deinterleave = 1;    // de-interleave the LDS tile rows of the step-2 scales
    int group_max = (int)vj::GROUP_MAX;   // vj_detect_chain groups up to this many raw candidates of one frame on the device (more: host path)
    bool tree_split_queues = true; // stage trees: the grid pass's survivors go down the tree while the tiles still run
    bool cv_tiles = true;         // OpenCV profile: small scales of stump cascades on LDS tiles (vj_cv_tile.hip)
    int cv_tile_ws_max = 512;     // ... windows a tile carries into its wave-split finish
    int cv_row_blocks = -1;       // ... workgroups per CU of cv_profile_pass while it runs next to the tiles (their LDS budget shrinks with it);
                                  // -1: 2 for stump cascades (round 4: the row kernel's pair / stump-parallel forms need fewer waves), 3 for multi-node trees
    int cv_row_blocks_tree = 2, cv_tile_min_windows_tree = 256;   // ... the same two for stage trees (swept: profiles/r03_cv_sweeps.log)
    bool cv_pairs = false;            // ... linear cascades' row kernel: two stumps per step in the sweeps of its queue
    int cv_tail_max = 64;             // ... a population of at most this many windows evaluates a stage stump-parallel (<= 64)
    int cv_tree_chunk = 64, cv_tree_chain_blocks = 2;   // ... windows per chunk and workgroups per CU of cv_tree_chain_pass
    bool cv_tiles_tilted = true;      // ... cascades with tilted features on LDS tiles too (the tilted integral's tile staged behind the sum's)
    bool cv_tree2 = true;             // ... cascades of two-node trees: the row kernel fetches both nodes of a tree at once
    int cv_row_band_px = 128;         // ... the row kernel's rows in band-major order, bands of this many pixels (0: scale after scale)
    bool cv_tree_chains = true;       // ... stage trees made of chains: compacting chain sweeps (0: the per-lane target-stage walk)
    int cv_tq_shift = 4;              // ... stage trees: the survivors' queue holds 1 / 2^shift of the tile windows (grows on overflow)
    int cv_tree_queue_cap = 0;        // ... stage trees: capacity of the prefix survivors' queue (0: a quarter of the tile windows)
    int cv_tile_min_windows0 = 2048;   // ... the same for the class with two tile workgroups per CU
    int cv_tile_min_windows = -1;     // ... a scale goes to tiles when a tile of at least this many windows fits the LDS (-1: 2048 for stump cascades, 1536 for trees)
    bool rois_on_device = true;   // vj_detect_rois: one region pass on the frames' integral images (0: one vj_detect per region size)
    int wide_tail = -1;           // queue passes: several windows in flight in the stump-parallel tail (-1: batches of <= 4 frames)
    int min_chunk = 32;           // queue passes: smallest chunk of windows a wave draws when there are fewer than 64 per wave
    int q_band_px = 128;          // first-pass units of a frame are ordered by image band of this height (then scale), and the queue pass of a
                                  // batch draws groups of units band-major (0: scale-major units, chunk-by-chunk queue pass)
    int q_group_units = 4;        // ... units per group (their survivors fill a wave's 512-entry queue about once: 2-4 equal, 5 / 6 / 8 / 16 lose 0.5 / 1 / 2-3 / 3-7 ms of 42.5)
    int q_band_min_frames = 8;    // ... batches of at least this many frames (a single frame keeps the thin-pass machinery)
    int q_slices = -1;            // queue passes: slices of a part handed out frame-major (-1: one per frame of the part's frame group)
    bool thin_pass_spread = true; // queue passes with fewer chunks than waves: only the first workgroups draw tickets
    int sp_tail_max = 48;         // global-gather sweeps switch to the stump-parallel tail when a wave holds at most this many windows (0: never)
    int gather_pairs = -1;        // global-gather sweeps evaluate two stumps per step, all their gathers in flight together: 0 never, 1 for
                                  // waves that hold a single chunk, 2 always, -1 = by batch size: 2 up to 4 frames (a single frame is bound
                                  // by the LATENCY of thin waves walking 16 stages: queue pass 0.78 -> 0.64 ms), 0 beyond (a faster gather chain
                                  // only takes issue slots from the tile chain: 64 x 1080p 47.98 / 48.21 / 49.21 ms for 0 / 1 / 2)
    uint32_t pairs_for(int n_frames) const { return gather_pairs >= 0 ? (uint32_t)gather_pairs : n_frames <= 4 ? 2u : 0u; }
    std::vector<int> split_override;
    std::vector<int> pass_cut_nodes{35};    // default pass cuts, in cumulative nodes (profiles/r03_notes.md #6c: 35 beats 150 on four cascades)
};

namespace vj {
// image staging and the integral launches (vj_env.cpp)
int image_channels(const vj_image& im);
uint32_t frame_elems_for(int W, int H);
int ensure_image_buffers(vj_env* e, int W, int H, int frames, bool need_gray, int channels = 1, Lane* lane = nullptr);
int enqueue_integral(vj_env* e, const uint8_t* d_gray, size_t frame_bytes, int stride, int W, int H, int frames,
                     int channels = 1);
int stage_frames(vj_env* e, const vj_image* frames, int n, int W, int H, const uint8_t** d_ptr, size_t* frame_bytes,
                 int* stride, Lane* lane = nullptr, hipStream_t copy_stream = nullptr);
int enqueue_tilted(vj_env* e, const uint8_t* d_gray, size_t frame_bytes, int stride, int W, int H, int frames, int channels);
}  // namespace vj
