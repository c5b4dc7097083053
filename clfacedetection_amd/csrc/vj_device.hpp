// Structures shared between the host driver (vj_env.cpp) and the gfx950 kernels
// (vj_kernels.hip).  All are plain PODs laid out for scalar (s_load) access: every
// field a wave needs is uniform across its 64 lanes.
#pragma once
#include <stdint.h>

namespace vj {

// One accepted scale (setupScale output, clod.cpp:371-415, in device units).
struct ScaleDev {
    float    step;         // window stride in pixels (f32, as the reference keeps it)
    uint32_t nx;           // windows per grid row
    uint32_t nwin;         // nx * ny
    uint32_t e_lt;         // equ_rect left-top, ELEMENT offset from the window origin
    uint32_t e_dw;         // equ_rect width in elements
    uint32_t e_dh;         // equ_rect height * stride in elements
    float    area;         // (float)scaled_window_area
    uint32_t table_first;  // first NodeRec of this scale in the table (image-stride offsets)
    uint32_t q_base;       // first entry of this scale's segment in the survivor queues
    uint32_t q_cap;        // capacity of ONE of the segment's Q_PARTS parts (= nwin * frames per part)
    uint32_t scale_idx;    // k of s_k (for the detection record)
    uint32_t ny;           // window grid rows
    // LDS-tile path (tile_rw == 0: this scale runs on the global-gather path)
    uint32_t tile_rw;      // != 0: this scale's first pass runs on the LDS-tile kernel
    uint32_t tile_pitch;   // dwords per row of the LDS image tile
    uint32_t tile_rows;    // rows of the LDS image tile
    uint32_t tile_table_first;  // NodeRec table built with stride = tile_pitch
    uint32_t te_lt;        // equ_rect left-top in tile-pitch elements
    uint32_t te_dh;        // equ_rect height * tile_pitch
    uint32_t tiles_x;      // tiles per grid row
    uint32_t tile_tw;      // windows per tile row
    uint32_t tile_th;      // window rows per tile (tile_tw * tile_th <= TILE_WAVES * TILE_WAVE_CAP)
    uint32_t tile_row_end; // window rows [0, tile_row_end) run on the LDS-tile kernel, the rest on the global-gather path
    uint32_t tile_class;   // LDS size class of the tile launch
    uint32_t tile_half;    // != 0: step is exactly 2 and the tile rows are de-interleaved: even image columns
                           // first, odd columns from element tile_half on (window origins are all even, so a
                           // wave's gathers of one corner touch CONSECUTIVE dwords: no bank conflicts)
    int32_t  te_dw;        // equ_rect left->right distance in tile elements (signed when de-interleaved)
    uint32_t tile_x4;      // rows are staged 16 bytes per lane (pitch is a multiple of 4 dwords)
    uint32_t skip_base;    // P2 skip modes: first word of this scale in a frame's visited-window bitmap
    uint32_t skip_wpr;     // ... words per window row (VJ_FLAG_SKIP_ROW); 0: one flattened bit list (VJ_FLAG_SKIP_LIST)
    float    scale_f;      // s_k itself, and the scaled window: what a region of interest needs to lay out its own grid
    uint32_t win_w, win_h; // ... (setupScale, clod.cpp:371-415, evaluated on the device per region: roi_plan_units)
    uint32_t pos_base;     // != 0: window positions of this scale come from CascadeArgs::pos_tab[pos_base + index] (VJ_FLAG_GRID_F64)
};
static_assert(sizeof(ScaleDev) == 128, "ScaleDev is 128 bytes");

// One cascade stage with its resolved successors (tempcv.cpp:834-861 flattened).
struct StageDev {
    uint32_t first_node;   // flat node index (same for every scale)
    uint32_t n_nodes;
    float    threshold;
    int32_t  on_pass;      // next stage, or -1 accept
    int32_t  on_fail;      // next stage, or -2 reject
    uint32_t n_trees;
    uint32_t order;        // stage-tree sweep: the i-th record holds the i-th stage to visit (topological)
    uint32_t sp_first;     // stump-parallel finish: index of the stage's first block in CascadeArgs::sp_blocks
    float    sp_delta;     // ... |tree-order stage sum - sequential stage sum| <= sp_delta for ANY window (see
                           //     tile_stump_parallel): 4 * n * 2^-24 * sum_k max(|left_k|, |right_k|), rounded up
    uint32_t cv_f64;       // OpenCV profile: this stage multiplies in f64 (two_rects stump stage, tempcv.cpp:872-888)
    uint32_t pad[2];
};
static_assert(sizeof(StageDev) == 48, "StageDev is 48 bytes");

// One unit of first-pass work inside a frame: a run of consecutive windows of one scale.
struct UnitDev {
    uint32_t scale;        // index into ScaleDev[]
    uint32_t first;        // first window index (row-major in the scale's grid), or ix0 | iy0 << 16 of a 2-D block
    uint32_t count;        // <= UNIT_WINDOWS (2-D block: width * height)
    uint32_t bw;           // 0: a run of consecutive windows; else the width of a 2-D block of windows
};

// Survivor record handed from one pass to the next: byte offset of the window origin
// in the batch sum image (frame included) and the window's variance.  It is the
// on-device twin of CLODSubwindowData (clod.cpp:33-38) without x/y (recoverable from
// the offset).
struct QEntry {
    uint32_t off;
    float    var;
};

// A raw detection: window origin offset + scale slot.
struct DetEntry {
    uint32_t off;
    uint32_t scale;        // index into ScaleDev[]
};

// A region of interest on the device (vj_roi's layout: frame, x, y, w, h) and one unit of work inside it.
struct RoiDev {
    int32_t frame, x, y, w, h;
};
struct RoiUnit {
    uint32_t roi;          // index into the region list
    uint32_t slot;         // scale (index into ScaleDev[])
    uint32_t first;        // first window of the unit, row-major in the region's own grid of this scale
    uint32_t count;        // <= UNIT_WINDOWS
    uint32_t nx, ny;       // the region's grid of this scale
};
struct RoiDet {
    uint32_t off;          // window origin, byte offset in the batch sum image
    uint32_t slot;
    uint32_t roi;
};
struct RoiTile {           // one LDS tile inside a region (cascade_tile_roi_pass)
    uint32_t roi;
    uint32_t slot;
    uint32_t first;        // ix0 | iy0 << 16 in the region's own grid of this scale
    uint32_t nxy;          // the region's grid: nx | ny << 16
    uint32_t twh;          // this tile's shape in windows, tw | th << 16: the region's grid cut into EQUAL parts no larger than the scale's
                           // tile (a 51 x 51 grid in 64 x 32 tiles: two tiles of 51 x 26, not 51 x 32 + 51 x 19), and only the rows / columns
                           // of the image such a tile needs are staged
};

// One block of <= 64 consecutive nodes of a stage (stump-parallel finish of the tile kernel).
struct SpBlock {
    uint32_t first_node;   // flat node index
    uint32_t desc;         // jn | jb << 8 | b << 16 | nb << 20 | stage << 24
};

constexpr int UNIT_WINDOWS = 512;   // windows per wave-unit == per-wave LDS queue capacity
constexpr int WAVES_PER_BLOCK = 3;    // waves per workgroup of the global-gather kernels: with one workgroup per CU next to the tile
                                      // chain, 2 starve the gather chain (+12 ms); batches run 4 (GATHER_WAVES_MAX: the gather chain is
                                      // 11 % faster, the tiles 1 % slower, and the balance moves three quarters of a scale further:
                                      // profiles/r03_notes.md #6g), single frames and stage trees 3 (CascadeArgs::gather_waves)
constexpr int GATHER_WAVES_MAX = 4;   // (the workgroup's queues: 4 KiB of LDS per wave, 16 KiB = what the tile classes leave free)
constexpr int MAX_SCALES = 128;
constexpr int MAX_PASSES = 8;         // == VJ_MAX_PASSES
constexpr uint32_t Q_PARTS = 8;       // parts of a scale's survivor-queue segment, by frame group (one per XCD to drain)
constexpr int TILE_WAVES = 8;        // waves per workgroup of the LDS-tile kernel (16 measured slower)
constexpr int TILE_W = 64;           // windows per tile row (= lanes of a wave)
constexpr int TILE_WAVE_CAP = 256;    // windows (= LDS queue entries) per wave of the tile kernel
constexpr int TILE_CLASSES = 3;       // LDS size classes, one launch each
constexpr int TILE_LDS_HEADER = (TILE_WAVES * TILE_WAVE_CAP * 2 + 32) * 4;  // queues + per-wave counts, bytes
constexpr int TILE_SP_MAX_WINDOWS = 256;  // windows a tile may carry into the finish: entries + verdict masks + partial sums fit the 16 KiB queue area
constexpr int TILE_WS_MAX_WINDOWS = 512;  // wave-split finish: 8 chunks of packed entries (4 KiB) + 8 x 320 dwords of sums and verdict words
constexpr int TILE_SEG_MAX_WINDOWS = 256;  // stage-tree chains inside a tile: population + reject list (256 entries each) around the scratch
constexpr int TILE_SP_MAX_BLOCKS = 4;     // blocks of 64 stumps per stage at most (stages of <= 256 nodes)
constexpr int TILE_SP_BLOCK = 64;         // stumps evaluated per round and window (= lanes of a wave)
constexpr int TILE_SP_FIELDS = 14;        // dwords of a node record kept in the LDS copy of a stage's table

struct CascadeArgs {
    const uint32_t* sum;        // batch sum images, frame f at f * frame_elems
    const uint64_t* sqsum;      // batch squared-sum images, same geometry
    const uint32_t* table;      // NodeRec[] viewed as dwords (16 per node)
    const ScaleDev* scales;
    const StageDev* stages;
    const UnitDev*  units;      // first-pass units of ONE frame (global-gather scales)
    uint32_t n_units;           // units per frame
    const UnitDev*  tile_units; // first-pass tiles of ONE frame (LDS-tile scales): first = ix0 | iy0 << 16
    uint32_t n_tile_units;
    uint32_t tile_lds_bytes;    // dynamic LDS of the tile kernel
    uint32_t* tile_ticket;      // eight ticket counters of this tile launch, one per part of the tile list (zeroed before the launch)
    uint32_t n_frames;
    uint32_t n_scales;
    uint32_t frame_elems;       // elements per frame in sum / sqsum
    uint32_t sum_bytes;         // n_frames * frame_elems * 4 (< 2^32)
    uint32_t stride;            // elements per image row (W + 1)
    uint32_t stage_begin, stage_end;  // stages [begin, end) evaluated by this pass
    uint32_t total_waves;       // gridDim.x * gather_waves
    uint32_t gather_waves;      // waves per workgroup of cascade_pass / cascade_roi_pass (3 or 4; stage trees: 3)
    uint32_t xcd_affinity;      // grid pass: waves of one XCD share a contiguous part of the (frame, unit) list
    const QEntry*   q_in;       // survivor queue read by this pass (passes > 0)
    const uint32_t* q_in_count; // entry counts of q_in, [scale][Q_PARTS]
    uint32_t* q_ticket;         // Q_PARTS chunk-ticket counters of this queue pass (zeroed before the launch)
    uint32_t wide_tail;         // queue passes: the stump-parallel tail keeps several windows' gathers in flight (small batches)
    uint32_t min_chunk;         // queue passes: smallest chunk of windows a wave draws (1..64)
    uint32_t thin_pass_spread;  // queue passes with fewer chunks than waves: only the first workgroups draw tickets (even load per CU)
    uint32_t q_slices;          // queue passes: a part's chunks are handed out in this many slices of every scale's range (frame-major order)
    // Band-major queue pass (batches; linear cascades): the first pass records where every unit's survivors landed in their
    // (scale, part) sub-queue (run_table[frame * n_units + unit] = {first entry, count}); the queue pass then draws GROUPS of
    // consecutive units of one (image band, scale) — the host orders a frame's units by band first — in (frame, group) order:
    // an XCD's waves work on one band of one frame across all scales before they move on (their gathers share an L2-sized
    // part of the sum image), and nothing has to be sorted.
    uint32_t* run_table;        // [n_frames * n_units][2] (null: not recorded)
    const UnitDev* q_groups;    // {scale, first unit, units, band} (null: the queue pass walks the sub-queues chunk by chunk)
    uint32_t n_q_groups;        // groups per frame
    QEntry*   q_out;            // survivor queue written by this pass (not the last)
    uint32_t* q_out_count;
    QEntry*   q_fail;           // stage trees: queue of the chain that takes this pass's rejects (else null)
    uint32_t* q_fail_count;
    // Tile launches: one survivor queue per pass boundary.  A wave sweeps the cascade one
    // pass segment at a time and leaves at boundary p — appending to queue p — as soon as
    // fewer than tile_min_lanes windows survive (or the boundary is >= tile_end).
    uint32_t  n_pass;                       // segments: [pass_begin[p], pass_begin[p+1])
    uint32_t  pass_begin[MAX_PASSES + 1];
    QEntry*   q_pass[MAX_PASSES];           // q_pass[p]: windows waiting to enter segment p (p >= 1)
    uint32_t* q_pass_count[MAX_PASSES];
    uint32_t  tile_end;                     // deepest stage a tile launch may enter
    uint32_t  tile_min_lanes;               // leave at a pass boundary when the whole tile has fewer survivors
    unsigned long long tile_repack_mask;    // bit s: re-pack the tile's survivors across its waves before stage s
    // Stump-parallel finish (stump cascades): once a tile is down to <= TILE_SP_MAX_WINDOWS windows at a
    // re-pack point at or after tile_sp_begin, its 512 lanes evaluate (window, stump) pairs in parallel
    // and one lane per window adds the stump values in cascade order — through the last stage.
    uint32_t  tile_sp_begin;                // >= number of stages: disabled
    uint32_t  tile_sp_pad;                  // dwords of LDS reserved for the finish: two record blocks + leaf values (0 = off)
    uint32_t  tile_sp_max;                  // enter the finish when at most this many windows are left
    uint32_t  identity_order;               // StageDev::order is 0, 1, 2, ... (every linear cascade)
    uint32_t  n_seg;                        // stage tree: chains after the linear prefix that a tile may run itself (0: none)
    uint32_t  seg_end[4];                   // ... end position (sweep order) of chain k; it starts where chain k-1 (or the prefix) ends
    uint32_t  seg_chain;                    // ... bit k: the rejects of chain k are the population of chain k+1
    uint32_t  tree2;                        // every tree has exactly two nodes, the second one the child of the first
    uint32_t  tile_finish;                  // 0: stump-parallel finish, 1: wave-split finish (tile_wave_split)
    uint32_t  tile_ws_min;                  // ... and hand over to the stump-parallel finish below this many
    uint32_t  tile_ws_max;                  // enter the wave-split finish when at most this many windows are left
    const SpBlock* sp_blocks;               // per block of <= 64 stumps, all stages in order
    uint32_t  n_sp_blocks;
    DetEntry* det;              // detections (last pass)
    uint32_t* det_count;
    uint32_t  det_cap;
    uint32_t  signed_mean;      // VJ_FLAG_SIGNED_MEAN
    unsigned long long* stage_entered;  // [VJ_MAX_STAGES] when counting, else null
    unsigned long long* tree_ctr;       // counting, multi-node trees: [0] nodes below the root that a window's walk visited, [1] their rectangles
    // P2 skip modes (VJ_FLAG_SKIP_LIST / VJ_FLAG_SKIP_ROW; clod.cpp:729-732, :1430): a bitmap of the grid windows the
    // reference's sequential CPU loops visit, built by skip_fail_bits + skip_resolve before the cascade passes;
    // null = every grid window (the OpenCL kernel's contract)
    unsigned long long* skip_bits;      // [n_frames][skip_frame_words]
    uint32_t  skip_frame_words;
    uint32_t  sp_tail_max;              // global-gather sweeps: at most this many windows left in a wave -> stump-parallel tail (0: off; <= 48)
    uint32_t  max_stage_nodes;          // nodes of the cascade's largest stage
    uint32_t  gather_pairs;             // global-gather sweeps evaluate two stumps per step: 0 never, 1 when the wave holds one chunk, 2 always
    uint32_t  pos_mode;                 // window_pos(): bit 0 = round half away from zero (clod.cpp:1416) instead of lrint (:514); bit 1: the block variant's f64 grid (ScaleDev::pos_base tables)
    const uint32_t* pos_tab;            // positions of the f64 grids (VJ_FLAG_GRID_F64), per scale from ScaleDev::pos_base; entry 0 is unused
    const UnitDev* skip_units;          // one per bitmap word of a frame: {scale, first window (flattened index, or ix0 | iy << 16), valid bits, word}
    uint32_t  n_skip_units;
    const UnitDev* skip_segs;           // one per recurrence domain (a window row, or a scale's whole list): {scale, first word, words, -}
    uint32_t  n_skip_segs;
};

struct IntegralArgs {
    const uint8_t* gray;        // batch of frames
    uint64_t gray_frame_bytes;  // distance between frames
    uint32_t gray_stride;       // bytes per row
    uint32_t channels;          // 1: gray; 3: BGR, 4: BGRA (converted on the fly)
    uint32_t width, height;
    uint32_t n_frames;
    uint32_t n_bands;           // ceil(height / BAND_ROWS)
    uint32_t band_pitch;        // elements per band row in the band arrays (>= width, multiple of 4)
    uint32_t* band_sum;         // [frames][bands][band_pitch]  column sums of the band / exclusive prefix
    uint32_t* band_sq;          // [frames][bands][band_pitch]  column sums of squares of the band
    uint64_t* band_sq_prefix;   // [frames][bands][band_pitch]  exclusive prefix of band_sq over bands
    uint32_t* sum;              // [frames][frame_elems]
    uint64_t* sqsum;
    uint32_t frame_elems;
    uint32_t rows_mode;         // band_rows: 0 one wave walks a band's chunks; 1 the chunks side by side (band_rows_par); 2 side by side unless the call is a batch
};
constexpr int BAND_ROWS = 8;

// Launch wrappers (defined in vj_kernels.hip); stream is a hipStream_t.
int launch_integral(const IntegralArgs& a, void* stream);
int launch_cascade_pass(const CascadeArgs& a, bool from_grid, bool trees, bool last, bool count, bool general,
                        int n_blocks, void* stream);
int launch_cascade_tile_pass(const CascadeArgs& a, bool trees, bool count, bool staged, int n_blocks, void* stream);

// Regions of interest on the device (vj_detect_chain).
struct RoiArgs {
    // dets_to_rois: raw detections of a first cascade -> regions
    const DetEntry* det_in;
    const uint32_t* det_in_count;
    uint32_t det_in_cap;
    const ScaleDev* scales_in;   // the first cascade's scales (win_w / win_h)
    uint32_t frame_bytes;        // frame_elems * 4
    uint32_t stride;             // W + 1
    RoiDev* rois;                // region list (written by dets_to_rois, or supplied by the caller)
    uint32_t* n_rois;            // on the device
    uint32_t max_rois;
    // roi_plan_units: every region lays out the second cascade's grid inside itself
    uint32_t n_frames;
    int32_t  frame_w, frame_h;
    int32_t  win_w0, win_h0;     // the second cascade's base window
    int32_t  min_w, min_h, max_w, max_h;
    RoiUnit* units;
    uint32_t* n_units;
    uint32_t max_units;
    uint32_t* ticket;            // unit ticket counter of cascade_roi_pass
    // cascade_roi_pass
    RoiDet* det;
    uint32_t* det_count;
    uint32_t det_cap;
    // cascade_tile_roi_pass: (region, scale) grids of at least tile_min_windows windows on two-per-CU tile scales (null: none)
    RoiTile* tiles;
    uint32_t* n_tiles;
    uint32_t max_tiles;
    uint32_t tile_min_windows;
    uint32_t tile_blocks;        // workgroups of the launch
};
int launch_roi_chain(const RoiArgs& r, const CascadeArgs& a, bool from_dets, bool trees, bool count, bool general /* stage tree */,
                     int n_blocks, void* stream, void* stream2 /* may be null */, void* fork_ev, void* join_ev);

// Grouping of a first cascade's raw candidates on the device (vj_detect_chain with min_neighbors != 0): the grouped
// rectangles become the region list of the second cascade without a host round trip (vj_group_dev.hip).
constexpr uint32_t GROUP_MAX = 2048;   // raw candidates of ONE frame the device kernel groups; beyond that the host path runs
struct GroupArgs {
    const DetEntry* det;         // the first cascade's detection buffer (device order)
    const uint32_t* det_count;
    uint32_t det_cap;
    const ScaleDev* scales;      // the first cascade's scales (win_w / win_h)
    uint32_t frame_bytes;        // frame_elems * 4
    uint32_t stride;             // W + 1
    uint32_t n_frames;
    int32_t  threshold;          // MAX(min_neighbors, 1)
    double   eps;                // 0.2 (clod.cpp:11)
    uint32_t* frame_count;       // [n_frames]      raw candidates per frame          } one block, zeroed by the
    uint32_t* frame_cursor;      // [n_frames]      scatter cursors                   } launcher
    uint32_t* grouped_count;     // [n_frames]      grouped rectangles per frame      }
    uint32_t* overflow;          // [1]             frames with more than GROUP_MAX   }
    uint32_t* frame_first;       // [n_frames + 1]  exclusive prefix of frame_count
    uint64_t* keys;              // [det_cap]       candidates bucketed by frame: scale slot << 32 | element index in the frame
    RoiDev*   grouped;           // [det_cap]       per-frame segments (at frame_first[frame])
    uint32_t* grouped_weight;    // [det_cap]
    RoiDev*   rois;              // out: the region list, frames in order
    uint32_t* roi_weight;        // out: members of every group
    uint32_t* n_rois;            // out (device)
    uint32_t  max_rois;
    uint32_t  group_max;         // <= GROUP_MAX
};
int launch_group_rois(const GroupArgs& g, void* stream);
int prepare_group_kernels();   // per device: dynamic-LDS cap of group_frame
int launch_skip_bitmap(const CascadeArgs& a, bool trees, int n_blocks, void* stream);   // fail bits, then the visited bitmap
int prepare_tile_kernels();   // per device: raise the dynamic-LDS cap of the tile kernel's instantiations

}  // namespace vj

// ------------------------------------------------------------ OpenCV arithmetic profile (vj_cv.cpp)
namespace vj {

// One scale of cvHaarDetectObjects' scale-cascade loop (tempcv.cpp:1359-1417) with what
// cvSetImagesForHaarClassifierCascade (tempcv.cpp:549-632) derives for it.
struct CvScaleDev {
    double   ystep;        // max(2, factor)
    double   inv_area;     // weight_scale = 1 / (equ_w * equ_h)
    uint32_t win_w, win_h; // cvRound(orig * factor)
    uint32_t end_x, end_y; // window grid: ix in [0, end_x), iy in [0, end_y)
    uint32_t q0, q1, q2, q3;   // the four corners of equRect, element offsets from the window origin
    uint32_t table_first;  // first NodeRec of this scale
    uint32_t scale_idx;    // index of the factor in the enumeration (skipped scales keep their number)
    // LDS-tile path of the profile (vj_cv_tile.hip; tile_th == 0: the scale runs on cv_profile_pass)
    uint32_t tile_tw, tile_th;   // windows per tile row (64, 32 or 16: a tile row never straddles a bitmap word) and rows per tile (<= 32)
    uint32_t tile_pitch;         // dwords per row of the LDS image tile (a multiple of 4)
    uint32_t tile_rows;          // rows of the LDS image tile
    uint32_t tile_table_first;   // first CvNodeRec of this scale built with the tile's pitch
    uint32_t bits_base;          // first word of this scale in a frame's reject / visited bitmap ((end_x + 63) / 64 words per window row)
    // stage trees on tiles: the survivors of the tree's linear prefix wait in one sub-queue per scale (a chunk of the walk
    // then holds windows of ONE scale: full lanes, scalar table).  Sub-queue of this scale for n_frames frames: entries
    // [tq_first(n_frames, shift), + tq_cap(n_frames, shift)) — see cv_tq_first / cv_tq_cap
    uint32_t tq_win_first;       // grid windows of the tile scales before this one, per frame
    uint32_t tq_slot;            // this scale's number among the tile scales
};
static_assert(sizeof(CvScaleDev) == 88, "CvScaleDev is 88 bytes");
// one sub-queue: 1 / 2^shift of the scale's grid windows in the batch, at least 4096 entries
__host__ __device__ inline uint64_t cv_tq_cap(uint32_t end_x, uint32_t end_y, uint32_t n_frames, uint32_t shift) {
    return (((uint64_t)end_x * end_y * n_frames) >> shift) + 4096u;
}
__host__ __device__ inline uint64_t cv_tq_first(uint32_t tq_win_first, uint32_t tq_slot, uint32_t n_frames, uint32_t shift) {
    return (((uint64_t)tq_win_first * n_frames) >> shift) + (uint64_t)tq_slot * 4096u;
}

struct CvDet {
    uint32_t x, y, slot, frame;
};

// Node record of the OpenCV profile (64 bytes, fetched through the scalar cache like NodeRec): corner q of
// rectangle k sits at lt[k] + {0, da[k], db[k], da[k] + db[k]} bytes from the window origin — upright rectangles
// da = width, db = height * stride; tilted ones (tempcv.cpp:743-750) da = height * (stride - 1), db = width *
// (stride + 1), in the tilted integral image.
struct alignas(16) CvNodeRec {
    uint32_t lt[3], da[3], db[3];
    float    w[3];         // w[2] == 0: two rectangles
    float    thr;
    uint32_t left, right;  // f32 leaf value bits, or child node index (flags)
    uint32_t flags;        // NODE_LEFT_IS_NODE | NODE_RIGHT_IS_NODE | NODE_TREE_LAST | CV_NODE_TILTED
};
static_assert(sizeof(CvNodeRec) == 64, "CvNodeRec must be 64 bytes");
constexpr uint32_t CV_NODE_TILTED = 8u;

// A stage tree whose part after the linear prefix is a sequence of CHAINS (tempcv.cpp:834-861 on frontalface_alt_tree: stages
// 0-4, then 5, 7, ..., 39, whose rejects — anywhere — start 6, 8, ..., 46): chain k covers positions [begin[k], end[k]) of the
// sweep order; passing a stage leads to the next position, passing the chain's last stage accepts; a reject inside chain k
// continues at the first stage of chain k + 1 when bit k of `chained` is set, else it is final.  n == 0: the tree does not
// have this shape (the per-lane target-stage walk is used).
struct CvChainDev {
    uint32_t n;
    uint32_t begin[4], end[4];
    uint32_t chained;
    uint32_t tail_max;           // see CV_TAIL_MAX
};
constexpr uint32_t CV_TAIL_MAX = 64;        // a population of at most CvChainDev::tail_max <= this many windows evaluates a stage stump-parallel (lane = stump)
constexpr uint32_t CV_TAIL_BLOCKS = 8;      // ... stages of up to 8 x 64 nodes
constexpr uint32_t CV_TQ_CHUNK = 256;       // windows per chunk of cv_tree_chain_pass at most (LDS: 24 bytes each; CvTreeArgs::chunk <= this)

constexpr int CV_WAVES_PER_BLOCK = 4;
constexpr int VJ_MAX_STAGES_DEV = 64;  // == VJ_MAX_STAGES
constexpr int CV_QCAP = 320;           // survivors of stage 0 a wave collects before it sweeps the later stages

struct CvArgs {
    const uint32_t* sum;
    const uint32_t* tilted;      // tilted integral images, same geometry as sum (null: no tilted features)
    const uint64_t* sqsum;
    const uint32_t* table;       // NodeRec[] (offsets in bytes, f32 weights per tempcv.cpp:700-768)
    const CvScaleDev* scales;
    const StageDev* stages;      // threshold = stage threshold - 0.0001f (tempcv.cpp:262, 419)
    const UnitDev* rows;         // one unit per (scale, window row): {scale slot, iy}
    uint32_t n_rows;             // per frame
    uint32_t n_frames;
    uint32_t n_stages;
    uint32_t n_order;            // stage trees: stages reachable from stage 0, swept in StageDev::order
    uint32_t frame_elems;
    uint32_t stride;             // W + 1
    uint32_t sum_h;              // H + 1
    uint32_t total_waves;
    CvDet* det;
    uint32_t* det_count;
    uint32_t det_cap;
    unsigned long long* stage_entered;   // [VJ_MAX_STAGES] + [VJ_MAX_STAGES] = windows visited (border ones included)
    CvChainDev chains;           // stage trees made of chains (else n = 0)
    void* fail_scratch;          // ... CV_QCAP x 16 bytes per wave: where a chain's rejects wait for the next chain
    uint32_t tail_max;           // linear cascades: a wave's queue of at most this many windows evaluates a stage stump-parallel (<= CV_TAIL_MAX)
    uint32_t pairs;              // ... larger populations evaluate two stumps per step (all gathers in flight)
    uint32_t tree2;              // every tree is a root + its only node child, upright (frontalface_alt2): both nodes' gathers in flight
};

int launch_cv_profile_pass(const CvArgs& a, bool trees, bool count, bool stage_tree, int n_blocks, void* stream);

// The profile's LDS-tile kernel (vj_cv_tile.hip): small scales of stump cascades with linear stages and upright features.
constexpr int CVT_WAVES = 8;            // waves per workgroup
constexpr int CVT_WAVE_CAP = 256;       // windows (queue entries) per wave: four tile rows of <= 64 windows
constexpr int CVT_WS_MAX = 512;         // windows a tile may carry into the wave-split finish
constexpr int CVT_LDS_HEADER = CVT_WAVES * CVT_WAVE_CAP * 12 + 512;   // offset queue (u32) + norm-factor queue (f64) + counters + 32 reject words, bytes
struct CvTileArgs {
    const uint32_t* sum;
    const uint32_t* tilted;      // tilted integral images, same geometry (null: the cascade has no tilted features): staged behind the sum tile
    const uint64_t* sqsum;
    const uint32_t* table;       // CvNodeRec[] (the tile scales' records carry offsets in the tile's pitch)
    const CvScaleDev* scales;
    const StageDev* stages;      // as CvArgs::stages, plus sp_delta (order-independence bound of a stage's leaf sum)
    const UnitDev* tiles;        // tiles of ONE frame in this launch's LDS class: {scale slot, ix0 | iy0 << 16}
    uint32_t n_tiles;
    uint32_t* ticket;            // eight ticket counters of this launch (zeroed before it)
    uint32_t n_frames, n_stages, frame_elems;
    uint32_t stride;             // W + 1
    uint32_t sum_h;              // H + 1
    unsigned long long* bits;    // [n_frames][bits_frame_words]: stage-0 reject bits (pass 0), visited bits after skip_resolve
    uint32_t bits_frame_words;
    uint32_t lds_bytes;          // dynamic LDS of the launch
    unsigned long long repack_mask;   // bit s: pool the tile's survivors across its waves before stage s
    uint32_t ws_begin, ws_max;   // wave-split finish from this stage on, once at most ws_max (<= CVT_WS_MAX) windows are left
    CvDet* det;
    uint32_t* det_count;
    uint32_t det_cap;
    unsigned long long* stage_entered;   // as CvArgs
    // stage trees (mode 2): the survivors of the tree's linear prefix wait here for cv_tree_walk
    struct CvTreeEntry* tq;
    uint32_t* tq_count;          // one counter per tile scale (CvScaleDev::tq_slot), + [64]: entries that did not fit
    uint32_t tq_cap;             // total entries of the queue buffer (all sub-queues)
    uint32_t tq_shift;
};
// A window that passed the linear prefix of a stage tree inside a tile and walks the rest of the tree with global gathers.
struct CvTreeEntry {
    uint32_t off;        // byte offset of the window origin in the batch sum image
    uint32_t word;       // its word in the reject / accept bitmaps (frame included)
    uint32_t bit_slot;   // bit | scale slot << 8
    uint32_t pad;
    double   vnf;        // variance norm factor
};
int launch_cv_tile_pass(const CvTileArgs& a, int mode /* 0: reject bits of stage 0, 1: the cascade on the visited windows, 2: stage trees:
                        the linear prefix on every grid window */, bool count, bool tree2 /* every tree: a root and one node child */, int n_blocks, void* stream);
// Stage trees on tiles: the rest of the tree for the prefix's survivors (reject / accept bits), then — after skip_resolve — the
// accepted windows the walk visits become detections (vj_cv_profile.hip).
struct CvTreeArgs {
    const uint32_t* sum;
    const uint32_t* table;       // frame-stride CvNodeRec tables
    const CvScaleDev* scales;
    const StageDev* stages;
    uint32_t n_order, prefix;    // sweep positions [prefix, n_order) are walked
    uint32_t sum_bytes;          // n_frames * frame_elems * 4
    const CvTreeEntry* tq;
    const uint32_t* tq_count;
    uint32_t tq_cap;
    unsigned long long* reject;  // [n_frames][bits_frame_words]; visited bits after skip_resolve
    unsigned long long* accept;
    uint32_t bits_frame_words, n_frames;
    const UnitDev* segs;         // one per window row of a tile scale: {scale slot, first word, words per row}
    uint32_t n_segs;
    CvDet* det;
    uint32_t* det_count;
    uint32_t det_cap;
    // per-scale sub-queues + chains (cv_tree_chain_pass)
    uint32_t tq_shift;           // sub-queue capacities: cv_tq_cap(.., n_frames, tq_shift)
    uint32_t n_scales;
    uint32_t* ticket;            // chunk ticket counter (zeroed before the launch)
    CvChainDev chains;
    void* fail_scratch;          // CV_TQ_CHUNK x 24 bytes per wave
    uint32_t total_waves;
    uint32_t chunk;              // windows a wave draws at a time (64 .. CV_TQ_CHUNK)
};
int launch_cv_tree_walk(const CvTreeArgs& a, int n_blocks, void* stream);
int launch_cv_tree_chain_pass(const CvTreeArgs& a, int n_blocks, void* stream);
int launch_cv_tree_emit(const CvTreeArgs& a, int n_blocks, void* stream);
int prepare_cv_tile_kernels();   // per device: raise the dynamic-LDS cap
int launch_skip_resolve(const CascadeArgs& a, int n_blocks, void* stream);   // reject bits -> visited bits (vj_kernels.hip)

struct TiltedArgs {
    const uint8_t* gray;        // batch of frames
    uint64_t gray_frame_bytes;
    uint32_t gray_stride;
    uint32_t channels;
    uint32_t width, height;
    uint32_t n_frames;
    uint32_t frame_elems;
    uint32_t* tilted;           // [frames][frame_elems], rows of W + 1
};
int launch_tilted_integral(const TiltedArgs& a, void* stream);
int launch_grayscale(const TiltedArgs& a, uint8_t* dst, uint32_t dst_stride, void* stream);

}  // namespace vj
