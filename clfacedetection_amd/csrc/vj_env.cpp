// Host driver of the HIP path: device buffers, per-(cascade, size, params) plans and
// the launch sequence of one vj_detect call.  It replaces clodInitEnvironment /
// clodInitBuffers / clodDetectObjectsOpenCL (clod.cpp:72-163, 1176-1336): where the
// reference does, per frame, one blocking launch + count read-back per (scale, stage)
// — up to 924 round trips at 1080p — this driver enqueues 3 integral launches and a
// handful of cascade passes for the WHOLE batch and synchronises once.
#include "vj_internal.hpp"
#include "vj_device.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>

#include "vj_env_internal.hpp"

using namespace vj;

namespace vj {

static uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static constexpr uint32_t kSlackRows = 2;  // zero rows after row H (defined reads for the
                                           // one-column feature overshoot)

uint32_t frame_elems_for(int W, int H) {
    uint64_t e = (uint64_t)(W + 1) * (uint64_t)(H + 1 + kSlackRows);
    e = (e + 63u) & ~(uint64_t)63u;  // keep every frame 256-byte aligned
    return (uint32_t)e;
}

// Default pass split: stage boundaries after which the survivor population is small
// enough that re-packing it across the whole chip pays for the extra launch.
static std::vector<uint32_t> default_pass_bounds(const vj_cascade& c, const StageProgram& prog,
                                                 const std::vector<int>& override_, const std::vector<int>& cut_nodes) {
    const uint32_t n = (uint32_t)c.stages.size();
    std::vector<uint32_t> b{0};
    if (!override_.empty()) {
        for (int v : override_)
            if (v > (int)b.back() && v < (int)n) b.push_back((uint32_t)v);
    } else {
        // cuts after given numbers of cumulative nodes (default: one, after 35 — frontalface_alt: before stage 3 —
        // measured best on batches of four cascades: [0,3) global first pass, one queue pass for the rest whose chunks hold
        // the survivors of the whole batch re-packed, both on the global-gather chain while the tile chain runs the whole
        // cascade next to it; round 1 cut after ~150 nodes, which the frame-major chunk order of round 3 made too late)
        uint32_t acc = 0;
        size_t ci = 0;
        for (uint32_t s = 0; s < n && ci < cut_nodes.size(); ++s) {
            acc += prog.n_nodes[s];
            if (acc >= (uint32_t)cut_nodes[ci] && s + 1 < n) {
                b.push_back(s + 1);
                while (ci < cut_nodes.size() && acc >= (uint32_t)cut_nodes[ci]) ++ci;
            }
        }
    }
    b.push_back(n);
    while (b.size() > (size_t)VJ_MAX_PASSES + 1) b.erase(b.end() - 2);  // at most VJ_MAX_PASSES launches
    return b;
}

// Which scales go to LDS tiles: a class is acceptable for a scale when a tile holds at least min_windows windows, a scale
// whose best tile holds fewer than accept_windows stays on the global-gather path, and staging a tile may cost at most
// max_dwords_per_window.
struct TileThresholds {
    int min_windows, accept_windows, max_dwords_per_window;
};

static int build_plan(vj_env* e, const vj_cascade& c, int W, int H, const vj_params& p, Plan* pl, float tile_split,
                      const TileThresholds& thresholds, bool one_pass = false) {
    pl->tile_split = tile_split;
    if ((int)c.stages.size() > VJ_MAX_STAGES) {
        set_error("cascade has %zu stages; at most %d are supported", c.stages.size(), VJ_MAX_STAGES);
        return VJ_ERR_LIMIT;
    }
    if (!(p.flags & VJ_FLAG_TILTED_AS_UPRIGHT))
        for (const auto& nd : c.nodes)
            if (nd.tilted) {
                set_error("the cascade has tilted features: the clod profile would evaluate them as upright rectangles like the reference "
                          "(clod.cpp:448-492 never reads the flag) — pass VJ_FLAG_TILTED_AS_UPRIGHT for that, or use vj_detect_opencv");
                return VJ_ERR_UNSUPPORTED;
            }
    pl->prog = build_stage_program(c);
    const uint32_t skip_mode = p.flags & (VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW);
    pl->skip_mode = skip_mode;
    // how a grid index becomes a pixel position (window_pos): bit 0 = round half away from zero, bit 1 = f64 product.
    // precomputeWindows / the OpenCL path: lrint of the f32 product (clod.cpp:514); plain CPU loop: round() of it (:1416);
    // block variant: lrint of the f64 product in its row loop (:941-942), round() of it in its per-stage lists (:1034)
    pl->pos_mode = (p.flags & VJ_FLAG_GRID_F64) ? (skip_mode == VJ_FLAG_SKIP_LIST ? 3u : 2u)
                                                : (skip_mode == VJ_FLAG_SKIP_ROW ? 1u : 0u);
    for (const auto& t : c.trees)
        if (t.n_nodes != 1) pl->trees = true;
    if (pl->trees) {
        // two-node trees (frontalface_alt2): the tile kernel's wave-split finish knows this shape
        pl->tree2 = true;
        for (const auto& t : c.trees) {
            if (t.n_nodes != 2) { pl->tree2 = false; break; }
            const vj_node_desc& n0 = c.nodes[t.first_node];
            const vj_node_desc& n1 = c.nodes[t.first_node + 1];
            const int kids = (n0.left > 0) + (n0.right > 0);
            if (kids != 1 || (n0.left > 0 ? n0.left : n0.right) != 1 || n1.left > 0 || n1.right > 0) {
                pl->tree2 = false;
                break;
            }
        }
    }
    for (size_t s = 0; s < c.stages.size(); ++s) {
        const bool linear = pl->prog.on_fail[s] == STAGE_REJECT &&
                            (pl->prog.on_pass[s] == (int)s + 1 ||
                             (pl->prog.on_pass[s] == STAGE_ACCEPT && s + 1 == c.stages.size()));
        if (!linear) pl->general = true;
    }
    // Topological order of the pass/fail graph rooted at stage 0: the general path sweeps the stages once in it.
    std::vector<uint32_t> order;
    if (!stage_sweep_order(pl->prog, &order)) {
        set_error("stage links form a cycle");
        return VJ_ERR_UNSUPPORTED;
    }
    if (pl->general && e->general_prefix) {
        // the linear head of a stage tree (frontalface_alt_tree: stages 0..4 before the two chains split)
        uint32_t P = 0;
        while (P + 1 < order.size() && order[P] == P && pl->prog.on_fail[P] == STAGE_REJECT &&
               pl->prog.on_pass[P] == (int)(P + 1) && order[P + 1] == P + 1)
            ++P;
        if (P >= 2) pl->general_prefix = P;
    }
    pl->scales_all = plan_scales(c, W, H, p);
    pl->frame_elems = frame_elems_for(W, H);
    const uint32_t stride = (uint32_t)W + 1u;
    const size_t n_nodes = c.nodes.size();

    // stump-parallel finish: LDS room for two blocks of node records (field-major) and the leaf values of
    // the largest stage (dwords); stages of more than TILE_SP_MAX_BLOCKS * 64 nodes rule it out
    if (!pl->trees && !pl->general && e->tile_sp_begin < (int)c.stages.size()) {
        uint32_t mx = 0;
        for (size_t s = 0; s < c.stages.size(); ++s) mx = std::max(mx, pl->prog.n_nodes[s]);
        if (mx <= (uint32_t)TILE_SP_MAX_BLOCKS * TILE_SP_BLOCK)
            pl->sp_pad = (2u * TILE_SP_FIELDS * (TILE_SP_BLOCK + 1u) + 2u * mx + 3u) & ~3u;
    }
    const uint32_t tile_header_bytes = TILE_LDS_HEADER + pl->sp_pad * 4u;
    std::vector<NodeRec> table;
    std::vector<uint32_t> pos_tab;
    for (const vj_scale_info& si : pl->scales_all) {
        if (!si.accepted || si.nx <= 0 || si.ny <= 0) continue;
        if ((p.scale_mask[0] | p.scale_mask[1]) != 0) {  // scale subset (multi-GPU sharding of one frame)
            const int k = si.scale_idx;
            if (k >= 127 || !((p.scale_mask[k >> 6] >> (k & 63)) & 1ull)) continue;   // bit 127 = VJ_SCALE_MASK_NONE
        }
        if (pl->scales.size() >= (size_t)MAX_SCALES) {
            set_error("more than %d scales", MAX_SCALES);
            return VJ_ERR_LIMIT;
        }
        ScaleDev sd;
        memset(&sd, 0, sizeof(sd));
        sd.step = si.step;
        sd.nx = (uint32_t)si.nx;
        sd.nwin = (uint32_t)si.nx * (uint32_t)si.ny;
        sd.e_lt = (uint32_t)si.equ_y * stride + (uint32_t)si.equ_x;
        sd.e_dw = (uint32_t)si.equ_w;
        sd.e_dh = (uint32_t)si.equ_h * stride;
        sd.area = (float)si.area;
        sd.table_first = (uint32_t)table.size();
        sd.scale_idx = (uint32_t)si.scale_idx;
        sd.scale_f = si.scale;
        sd.win_w = (uint32_t)si.win_w;
        sd.win_h = (uint32_t)si.win_h;
        table.resize(table.size() + n_nodes);
        int rc = build_node_table(c, W, si, table.data() + sd.table_first);
        if (rc) return rc;
        // furthest element any gather of this scale can touch, relative to frame start
        // (positions are lrint(i * step); the other roundings of window_pos differ by at most one: one more covers them)
        const uint32_t x_max = (uint32_t)std::lrint((double)((float)(si.nx - 1) * si.step)) + (pl->pos_mode != 0u ? 1u : 0u);
        const uint32_t y_max = (uint32_t)std::lrint((double)((float)(si.ny - 1) * si.step)) + (pl->pos_mode != 0u ? 1u : 0u);
        uint32_t reach = sd.e_lt + sd.e_dh + sd.e_dw;
        for (size_t k = 0; k < n_nodes; ++k) {
            const NodeRec& r = table[sd.table_first + k];
            const uint32_t dw[3] = {r.dw01 & 0xffffu, r.dw01 >> 16, r.dw2_flags & 0xffffu};
            for (int q = 0; q < 3; ++q)
                if (q < 2 || r.w[2] != 0.0f) reach = std::max(reach, (r.lt[q] + r.dh[q] + dw[q]) / 4u);
        }
        reach += y_max * stride + x_max;
        pl->max_reach_elems = std::max(pl->max_reach_elems, reach);
        pl->windows_per_frame += sd.nwin;
        sd.ny = (uint32_t)si.ny;
        if (pl->pos_mode & 2u) {
            // the block variant's positions, index by index, with the reference's own expressions: lrint(index * step) in
            // its row loop (clod.cpp:941-942), round(index * step) in its per-stage lists (:1034), `step` a double (:862)
            if (pos_tab.empty()) pos_tab.push_back(0u);   // (base 0 means "no table")
            sd.pos_base = (uint32_t)pos_tab.size();
            const double stepd = std::max(2.0, (double)si.scale);
            const int n_pos = std::max(si.nx, si.ny) + 1;
            for (int i = 0; i < n_pos; ++i)
                pos_tab.push_back((pl->pos_mode & 1u) ? (uint32_t)std::round((double)i * stepd) : (uint32_t)std::lrint((double)i * stepd));
        }

        // LDS-tile path: does a 64 x (TILE_WAVES * rw) window tile's footprint fit the budget?
        uint32_t reach_x = (uint32_t)(si.equ_x + si.equ_w), reach_y = (uint32_t)(si.equ_y + si.equ_h);
        for (size_t k = 0; k < n_nodes; ++k) {
            const NodeRec& r = table[sd.table_first + k];
            const uint32_t dw[3] = {r.dw01 & 0xffffu, r.dw01 >> 16, r.dw2_flags & 0xffffu};
            for (int q = 0; q < 3; ++q)
                if (q < 2 || r.w[2] != 0.0f) {
                    const uint32_t lt = r.lt[q] / 4u;
                    reach_x = std::max(reach_x, lt % stride + dw[q] / 4u);
                    reach_y = std::max(reach_y, lt / stride + (r.dh[q] / 4u) / stride);
                }
        }
        if ((!pl->general || pl->general_prefix) && si.ny < 65536 && si.nx < 65536) {
            // candidate tile shapes; per class the shape with the most windows that fits wins
            static const uint32_t kTw[] = {64, 48, 32, 24, 16, 12, 8}, kTh[] = {32, 24, 16, 12, 8, 6, 4};
            uint32_t best_cls = TILE_CLASSES, best_n = 0, b_tw = 0, b_th = 0, b_pitch = 0, b_rows = 0;
            for (uint32_t cls = 0; cls < TILE_CLASSES && best_n < (uint32_t)thresholds.min_windows; ++cls) {
                const int kb = e->tile_class_kb[cls];
                const uint32_t lds_cu = (160u - (uint32_t)e->tile_lds_reserve_kb) * 1024u;
                // (2 KiB of the CU's share stay free so that the nested class blocks below can be rounded up to whole
                // allocation granules)
                const uint64_t budget = kb < 0 ? ((lds_cu - 2048u) / (uint32_t)(-kb) - tile_header_bytes) & ~63ull
                                               : std::min<uint64_t>((uint64_t)kb * 1024u, 160u * 1024u - tile_header_bytes);
                for (uint32_t tw : kTw)
                    for (uint32_t th : kTh) {
                        const uint32_t nwt = tw * th;
                        if (nwt > TILE_WAVES * TILE_WAVE_CAP || nwt < 64 || nwt <= best_n) continue;
                        // rows staged 16 bytes per lane (tile_stage_x4; not the de-interleaved step-2 tiles, whose
                        // sources are 8 bytes apart): pitch a multiple of 4 dwords, not of 32 (rows would share banks);
                        // otherwise an odd pitch
                        const bool x4 = e->tile_stage_x4 && !(e->tile_deinterleave && si.step == 2.0f);
                        uint32_t pitch = (uint32_t)std::ceil((double)(tw - 1) * (double)si.step) + 3u + reach_x;
                        if (x4) {
                            pitch = (pitch + 3u) & ~3u;
                            if (pitch % 32u == 0u) pitch += 4u;
                        } else {
                            pitch |= 1u;
                        }
                        const uint32_t rows = (uint32_t)std::ceil((double)(th - 1) * (double)si.step) + 3u + reach_y;
                        if ((uint64_t)pitch * rows * 4u > budget) continue;
                        // staging a tile must stay far cheaper than gathering its windows from L2
                        if ((uint64_t)pitch * rows > (uint64_t)thresholds.max_dwords_per_window * nwt) continue;
                        best_cls = cls; best_n = nwt; b_tw = tw; b_th = th; b_pitch = pitch; b_rows = rows;
                    }
            }
            if (best_n >= (uint32_t)thresholds.accept_windows) {
                sd.tile_rw = 1;
                sd.tile_tw = b_tw;
                sd.tile_th = b_th;
                sd.tile_pitch = b_pitch;
                sd.tile_rows = b_rows;
                sd.tile_class = best_cls;
            }
        }
        if (sd.tile_rw) {
            sd.tile_table_first = (uint32_t)table.size();
            table.resize(table.size() + n_nodes);
            // step exactly 2: every window origin is an even column -> de-interleave the tile rows
            sd.tile_half = (e->tile_deinterleave && si.step == 2.0f) ? (sd.tile_pitch + 1u) / 2u : 0u;
            sd.tile_x4 = (e->tile_stage_x4 && !sd.tile_half && sd.tile_pitch % 4u == 0u) ? 1u : 0u;
            if (getenv("VJ_DEBUG_PLAN"))
                fprintf(stderr, "vj plan: scale %d s=%.3f step=%.2f nx=%u ny=%u class %u tile %ux%u = %u windows, pitch %u rows %u (%u B)%s%s\n",
                        si.scale_idx, si.scale, si.step, sd.nx, sd.ny, sd.tile_class, sd.tile_tw, sd.tile_th, sd.tile_tw * sd.tile_th,
                        sd.tile_pitch, sd.tile_rows, sd.tile_pitch * sd.tile_rows * 4u, sd.tile_half ? " deinterleaved" : "",
                        sd.tile_x4 ? " x4" : "");
            rc = build_node_table_stride(c, sd.tile_pitch, si, table.data() + sd.tile_table_first, sd.tile_half);
            if (rc) return rc;
            auto col = [&](uint32_t cx) { return sd.tile_half ? (cx & 1u) * sd.tile_half + (cx >> 1) : cx; };
            sd.te_lt = (uint32_t)si.equ_y * sd.tile_pitch + col((uint32_t)si.equ_x);
            sd.te_dw = (int32_t)col((uint32_t)(si.equ_x + si.equ_w)) - (int32_t)col((uint32_t)si.equ_x);
            sd.te_dh = (uint32_t)si.equ_h * sd.tile_pitch;
            sd.tiles_x = (sd.nx + sd.tile_tw - 1) / sd.tile_tw;
            sd.tile_row_end = sd.ny;
        } else {
            // unstaged in the tile kernel (global_blocks): 2-D blocks of <= 2048 windows, about as wide as high
            sd.tile_tw = std::min<uint32_t>(sd.nx, 48u);
            sd.tile_th = std::min<uint32_t>(sd.ny, (uint32_t)(TILE_WAVES * TILE_WAVE_CAP) / sd.tile_tw);
            sd.tile_row_end = 0;
        }
        pl->scales.push_back(sd);
        pl->scales_info.push_back(si);
    }
    if (pl->max_reach_elems >= pl->frame_elems) {
        set_error("feature reach %u exceeds the frame allocation %u", pl->max_reach_elems, pl->frame_elems);
        return VJ_ERR_LIMIT;
    }
    for (size_t s = 0; s < c.stages.size(); ++s) {
        StageDev sd;
        memset(&sd, 0, sizeof(sd));
        sd.first_node = pl->prog.first_node[s];
        sd.n_nodes = pl->prog.n_nodes[s];
        sd.threshold = c.stages[s].threshold;
        sd.on_pass = pl->prog.on_pass[s];
        sd.on_fail = pl->prog.on_fail[s];
        sd.n_trees = (uint32_t)c.stages[s].n_trees;
        sd.order = s < order.size() ? order[s] : 0u;
        {   // rigorous bound on how far two summation orders of this stage's leaf values can differ:
            // every order satisfies |fl_sum - exact| <= gamma_{n-1} * sum|a_k| (Higham), gamma_k = k u / (1 - k u),
            // u = 2^-24, and |a_k| <= max(|left_k|, |right_k|); the factor 4 (instead of 2) also covers the
            // rounding of the comparison itself
            double amax = 0.0;
            const vj_stage_desc& st = c.stages[s];
            for (int t = 0; t < st.n_trees; ++t) {
                const vj_tree_desc& td = c.trees[st.first_tree + t];
                double m = 0.0;
                for (int k = 0; k <= td.n_nodes; ++k) m = std::max(m, (double)std::fabs(c.alpha[td.first_alpha + k]));
                amax += m;
            }
            const double n = (double)pl->prog.n_nodes[s];
            sd.sp_delta = (float)(4.0 * n * std::ldexp(1.0, -24) * amax * 1.001 + 1e-30);
        }
        pl->stages.push_back(sd);
    }
    pl->n_order = (uint32_t)order.size();
    for (size_t s2 = 0; s2 < c.stages.size(); ++s2) pl->max_stage_nodes = std::max(pl->max_stage_nodes, pl->prog.n_nodes[s2]);
    // blocks of <= 64 consecutive nodes per stage, balanced, for the stump-parallel finish of the tile kernel
    std::vector<SpBlock> sp_blocks;
    for (size_t s = 0; s < c.stages.size(); ++s) {
        const uint32_t S = pl->prog.n_nodes[s], nb = (S + TILE_SP_BLOCK - 1u) / TILE_SP_BLOCK;
        pl->stages[s].sp_first = (uint32_t)sp_blocks.size();
        uint32_t jb = 0;
        for (uint32_t b = 0; b < nb && nb <= (uint32_t)TILE_SP_MAX_BLOCKS && s < 256; ++b) {
            const uint32_t jn = S / nb + (b < S % nb ? 1u : 0u);
            sp_blocks.push_back(SpBlock{pl->prog.first_node[s] + jb, jn | (jb << 8) | (b << 16) | (nb << 20) | ((uint32_t)s << 24)});
            jb += jn;
        }
    }
    pl->n_sp_blocks = (uint32_t)sp_blocks.size();
    // Balance between the two chains: tile_split scales' worth of tile work (counted from the largest tile
    // scale down, fractions by window rows) moves to the global-gather chain, which overlaps the tile chain.
    {
        double remaining = std::max(0.0, (double)tile_split);
        for (size_t k = pl->scales.size(); k-- > 0 && remaining > 0.0;) {
            ScaleDev& sd = pl->scales[k];
            if (!sd.tile_rw) continue;
            const double move = std::min(1.0, remaining);
            remaining -= move;
            const uint32_t keep_rows = (uint32_t)((double)sd.ny * (1.0 - move));
            sd.tile_row_end = keep_rows / sd.tile_th * sd.tile_th;
        }
    }
    // first-pass units of the global-gather path: whole scales, and the rows of split scales the tiles leave
    for (uint32_t slot = 0; slot < pl->scales.size(); ++slot) {
        const ScaleDev& sd = pl->scales[slot];
        const uint32_t row0 = sd.tile_row_end;
        if (row0 >= sd.ny) continue;
        if (e->grid_block_w > 0 && sd.nx < 65536 && sd.ny < 65536) {
            // 2-D blocks of <= UNIT_WINDOWS windows: a compact footprint in the sum image per wave (a row run
            // of 512 windows drags a full-width band of 20 s rows through the L2)
            const uint32_t bw = std::min<uint32_t>((uint32_t)e->grid_block_w, sd.nx);
            const uint32_t bh = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)UNIT_WINDOWS / bw, sd.ny));
            for (uint32_t iy0 = row0; iy0 < sd.ny; iy0 += bh)
                for (uint32_t ix0 = 0; ix0 < sd.nx; ix0 += bw)
                    pl->units.push_back(UnitDev{slot, ix0 | (iy0 << 16), bw * bh, bw});
        } else {
            for (uint32_t f = row0 * sd.nx; f < sd.nwin; f += UNIT_WINDOWS)
                pl->units.push_back(UnitDev{slot, f, std::min<uint32_t>(UNIT_WINDOWS, sd.nwin - f), 0});
        }
    }
    // Band-major order: the units of a frame sorted by the image band their top row lies in, then by scale — the waves of an
    // XCD (which take a contiguous run of the (frame, unit) list) then gather from ONE band of the sum image across all scales
    // instead of sweeping the frame once per scale — and, for the queue pass, groups of consecutive units of one (band, scale).
    if (e->q_band_px > 0 && !pl->units.empty()) {
        auto band_of = [&](const UnitDev& u) {
            const ScaleDev& sd = pl->scales[u.scale];
            const uint32_t iy0 = u.bw != 0u ? (u.first >> 16) : u.first / std::max(1u, sd.nx);
            return (uint32_t)std::lrint((double)((float)iy0 * sd.step)) / (uint32_t)e->q_band_px;
        };
        std::vector<std::pair<uint32_t, UnitDev>> keyed;
        keyed.reserve(pl->units.size());
        for (const UnitDev& u : pl->units) keyed.push_back({band_of(u), u});
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) {
            return std::make_tuple(a.first, a.second.scale) < std::make_tuple(b.first, b.second.scale);   // (stable: blocks stay row-major inside)
        });
        const uint32_t G = (uint32_t)std::max(1, e->q_group_units);
        for (size_t i = 0; i < keyed.size(); ++i) {
            pl->units[i] = keyed[i].second;
            if (pl->unit_groups.empty() || pl->unit_groups.back().bw != keyed[i].first || pl->unit_groups.back().scale != keyed[i].second.scale ||
                pl->unit_groups.back().count >= G)
                pl->unit_groups.push_back(UnitDev{keyed[i].second.scale, (uint32_t)i, 0u, keyed[i].first});
            ++pl->unit_groups.back().count;
        }
    }
    pl->pass_bounds = default_pass_bounds(c, pl->prog, e->split_override, e->pass_cut_nodes);
    // A call of a few frames may run the gather chain in ONE pass (one_pass_max_frames; the first pass runs every stage, thin waves finishing
    // stump-parallel): one dependent launch fewer — a noise frame at 1080p 1.23 -> 1.15 ms, a 4096 x 4096 one 7.81 -> 7.02 — but frames with
    // many deep survivors (blocks, drawn faces) lose as much without the queue pass's re-packing (profiles/r04_notes.md #15): off by default
    // (stump cascades: a tree cascade's thin waves have no stump-parallel form and lose — frontalface_alt2, one 1080p frame, 1.30 -> 1.41 ms)
    if (one_pass && !pl->general && !pl->trees && e->split_override.empty()) pl->pass_bounds = {0u, (uint32_t)c.stages.size()};
    if (pl->general) {   // positions in StageDev::order: [linear prefix | the rest]
        if (pl->general_prefix) pl->pass_bounds = {0u, pl->general_prefix, pl->n_order};
        else pl->pass_bounds = {0u, pl->n_order};
        // Can the rest be cut into linear segments?  A segment is a run of positions whose pass edges follow the
        // order, whose rejects all go to the first position of a LATER segment (or are final), and whose end accepts.
        if (pl->general_prefix) {
            struct Seg { uint32_t b, e; int fail_pos; };
            std::vector<Seg> segs;
            std::vector<int> pos_of(c.stages.size(), -1);
            for (uint32_t i = 0; i < pl->n_order; ++i) pos_of[order[i]] = (int)i;
            bool ok = true;
            uint32_t b = pl->general_prefix;
            while (b < pl->n_order && ok) {
                const int f = pl->prog.on_fail[order[b]];
                uint32_t e2 = b;
                while (true) {
                    const uint32_t sid = order[e2];
                    if (pl->prog.on_fail[sid] != f) { ok = false; break; }
                    const int np = pl->prog.on_pass[sid];
                    ++e2;
                    if (np == STAGE_ACCEPT) break;
                    if (e2 >= pl->n_order || np != (int)order[e2]) { ok = false; break; }
                }
                if (!ok) break;
                segs.push_back(Seg{b, e2, f == STAGE_REJECT ? -1 : pos_of[f]});
                b = e2;
            }
            for (const Seg& sg : segs)   // a reject target must be the start of a later segment
                if (sg.fail_pos >= 0) {
                    bool found = false;
                    for (const Seg& t : segs) found |= t.b == (uint32_t)sg.fail_pos && t.b >= sg.e;
                    ok &= found;
                }
            // passes: the prefix, then every segment in two parts (its first stages see most of its windows)
            std::vector<uint32_t> bounds{0u, pl->general_prefix};
            std::vector<uint8_t> last{0}, failq{0};   // index = pass; entry 0 = the prefix pass
            std::vector<int> seg_first_pass;
            for (const Seg& sg : segs) {
                seg_first_pass.push_back((int)bounds.size() - 1);
                // cuts inside a chain: after its third stage (most of its windows are gone by then) and, when the chain is
                // long and launches are left, once more after its seg_cut2-th — the survivors are re-packed chip-wide at
                // every cut instead of riding on in thin waves
                const uint32_t cut = sg.e - sg.b > 4 ? sg.b + 3 : sg.e;
                if (cut < sg.e) { bounds.push_back(cut); last.push_back(0); failq.push_back(0); }
                const uint32_t cut2 = e->seg_cut2 > 3 && sg.e - sg.b > (uint32_t)e->seg_cut2 + 4u && bounds.size() + 2 * (segs.size() - seg_first_pass.size()) + 1 < (size_t)VJ_MAX_PASSES
                                          ? sg.b + (uint32_t)e->seg_cut2 : sg.e;
                if (cut2 < sg.e) { bounds.push_back(cut2); last.push_back(0); failq.push_back(0); }
                bounds.push_back(sg.e); last.push_back(1); failq.push_back(0);
            }
            if (ok && !segs.empty() && bounds.size() - 1 <= (size_t)VJ_MAX_PASSES) {
                for (size_t k = 0; k < segs.size(); ++k) {
                    if (segs[k].fail_pos < 0) continue;
                    int target_pass = -1;
                    for (size_t t = 0; t < segs.size(); ++t)
                        if (segs[t].b == (uint32_t)segs[k].fail_pos) target_pass = seg_first_pass[t];
                    const int p_end = k + 1 < segs.size() ? seg_first_pass[k + 1] : (int)bounds.size() - 1;
                    for (int ps = seg_first_pass[k]; ps < p_end; ++ps) failq[(size_t)ps] = (uint8_t)target_pass;
                }
                pl->pass_bounds = bounds;
                pl->seg_last = last;
                pl->seg_fail = failq;
                bool chain_ok = segs.size() <= 4;
                for (size_t k = 0; k < segs.size() && chain_ok; ++k)
                    if (segs[k].fail_pos >= 0 && (k + 1 >= segs.size() || segs[k + 1].b != (uint32_t)segs[k].fail_pos)) chain_ok = false;
                if (chain_ok) {
                    pl->tile_n_seg = (uint32_t)segs.size();
                    for (size_t k = 0; k < segs.size(); ++k) {
                        pl->tile_seg_end[k] = segs[k].e;
                        if (segs[k].fail_pos >= 0) pl->tile_seg_chain |= 1u << k;
                    }
                }
            }
        }
    }
    // tile launches run deeper than the global-gather first pass (LDS gathers are ~10x cheaper)
    pl->tile_end = std::min<uint32_t>((uint32_t)c.stages.size(), std::max<uint32_t>((uint32_t)e->tile_end, pl->pass_bounds[1]));
    if (pl->pass_bounds.size() == 2) pl->tile_end = pl->pass_bounds[1];
    for (uint32_t cls = 0; cls < TILE_CLASSES; ++cls) {
        pl->class_first[cls] = (uint32_t)pl->tile_units.size();
        for (uint32_t slot = 0; slot < pl->scales.size(); ++slot) {
            ScaleDev& sd = pl->scales[slot];
            if (!sd.tile_rw || sd.tile_class != cls) continue;
            for (uint32_t iy0 = 0; iy0 < sd.tile_row_end; iy0 += sd.tile_th)
                for (uint32_t ix0 = 0; ix0 < sd.nx; ix0 += sd.tile_tw)
                    pl->tile_units.push_back(UnitDev{slot, ix0 | (iy0 << 16), 0, 0});
            pl->class_lds[cls] = std::max(pl->class_lds[cls], sd.tile_pitch * sd.tile_rows * 4u);
        }
        if (pl->class_lds[cls]) pl->class_lds[cls] += tile_header_bytes;
    }
    pl->class_first[TILE_CLASSES] = (uint32_t)pl->tile_units.size();
    // Nest the LDS blocks of consecutive classes: the tile launches follow each other on CUs where the gather
    // chain's workgroup holds a block somewhere in the middle of the LDS, so the k workgroups of one class have to
    // fit exactly into the hole the workgroup(s) of the other class leave behind (measured: a class-0 pair 2 KB
    // larger than the class-1 block it follows leaves one tile workgroup per CU idle, + 17 % on that launch).
    if (e->tile_lds_nest) {
        const uint32_t lds_cu = (160u - (uint32_t)e->tile_lds_reserve_kb) * 1024u;
        for (uint32_t cls = 0; cls + 1 < TILE_CLASSES; ++cls) {
            const int ka = e->tile_class_kb[cls], kb = e->tile_class_kb[cls + 1];
            if (ka >= 0 || kb >= 0 || !pl->class_lds[cls] || !pl->class_lds[cls + 1]) continue;
            const uint32_t na = (uint32_t)(-ka), nb = (uint32_t)(-kb);   // workgroups per CU: na > nb
            if (na <= nb || na % nb != 0u) continue;
            const uint32_t ratio = na / nb;
            uint32_t big = std::max(pl->class_lds[cls + 1], ratio * pl->class_lds[cls]);
            // whole allocation granules (the hardware hands LDS out in 512-byte granules; 1 KiB is safe)
            big = (big + ratio * 1024u - 1u) / (ratio * 1024u) * (ratio * 1024u);
            if ((uint64_t)big * nb > lds_cu) continue;
            pl->class_lds[cls + 1] = big;
            pl->class_lds[cls] = big / ratio;
        }
    }
    pl->block_first = (uint32_t)pl->tile_units.size();
    if (!pl->general && !pl->trees)
        for (uint32_t slot = 0; slot < pl->scales.size(); ++slot) {
            const ScaleDev& sd = pl->scales[slot];
            for (uint32_t iy0 = sd.tile_row_end; iy0 < sd.ny; iy0 += sd.tile_th)
                for (uint32_t ix0 = 0; ix0 < sd.nx; ix0 += sd.tile_tw)
                    pl->tile_units.push_back(UnitDev{slot, ix0 | (iy0 << 16), 0, 0});
        }
    pl->n_block_units = (uint32_t)pl->tile_units.size() - pl->block_first;
    pl->block_lds = tile_header_bytes;

    // P2 skip modes: one bitmap word per 64 consecutive windows of a recurrence domain — a window row
    // (VJ_FLAG_SKIP_ROW, clod.cpp:1430) or a scale's whole row-major list (VJ_FLAG_SKIP_LIST, clod.cpp:729-732)
    std::vector<UnitDev> skip_units, skip_segs;
    if (skip_mode) {
        if (pl->general) {
            set_error("the skip modes restate the reference's CPU loops, which run linear cascades only");
            return VJ_ERR_UNSUPPORTED;
        }
        uint32_t word = 0;
        for (uint32_t slot = 0; slot < pl->scales.size(); ++slot) {
            ScaleDev& sd = pl->scales[slot];
            sd.skip_base = word;
            if (skip_mode == VJ_FLAG_SKIP_ROW) {
                sd.skip_wpr = (sd.nx + 63u) / 64u;
                for (uint32_t iy = 0; iy < sd.ny; ++iy) {
                    skip_segs.push_back(UnitDev{slot, word, sd.skip_wpr, 0});
                    for (uint32_t ix0 = 0; ix0 < sd.nx; ix0 += 64u)
                        skip_units.push_back(UnitDev{slot, ix0 | (iy << 16), std::min<uint32_t>(64u, sd.nx - ix0), word++});
                }
            } else {
                sd.skip_wpr = 0;
                const uint32_t n_words = (sd.nwin + 63u) / 64u;
                skip_segs.push_back(UnitDev{slot, word, n_words, 0});
                for (uint32_t i0 = 0; i0 < sd.nwin; i0 += 64u)
                    skip_units.push_back(UnitDev{slot, i0, std::min<uint32_t>(64u, sd.nwin - i0), word++});
            }
        }
        pl->skip_frame_words = word;
        pl->n_skip_units = (uint32_t)skip_units.size();
        pl->n_skip_segs = (uint32_t)skip_segs.size();
    }

    int rc;
    if (!skip_units.empty()) {
        if ((rc = pl->d_skip_units.ensure(skip_units.size() * sizeof(UnitDev)))) return rc;
        if ((rc = pl->d_skip_segs.ensure(skip_segs.size() * sizeof(UnitDev)))) return rc;
        HIP_TRY(hipMemcpy(pl->d_skip_units.p, skip_units.data(), skip_units.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(pl->d_skip_segs.p, skip_segs.data(), skip_segs.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    }
    if ((rc = pl->d_table.ensure(std::max<size_t>(table.size(), 1) * sizeof(NodeRec)))) return rc;
    if ((rc = pl->d_scales.ensure(std::max<size_t>(pl->scales.size(), 1) * sizeof(ScaleDev)))) return rc;
    if ((rc = pl->d_stages.ensure(pl->stages.size() * sizeof(StageDev)))) return rc;
    if ((rc = pl->d_units.ensure(std::max<size_t>(pl->units.size(), 1) * sizeof(UnitDev)))) return rc;
    if ((rc = pl->d_tile_units.ensure(std::max<size_t>(pl->tile_units.size(), 1) * sizeof(UnitDev)))) return rc;
    if (!pos_tab.empty()) {
        if ((rc = pl->d_pos_tab.ensure(pos_tab.size() * sizeof(uint32_t)))) return rc;
        HIP_TRY(hipMemcpy(pl->d_pos_tab.p, pos_tab.data(), pos_tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!pl->unit_groups.empty()) {
        if ((rc = pl->d_unit_groups.ensure(pl->unit_groups.size() * sizeof(UnitDev)))) return rc;
        HIP_TRY(hipMemcpy(pl->d_unit_groups.p, pl->unit_groups.data(), pl->unit_groups.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    }
    if ((rc = pl->d_sp_blocks.ensure(std::max<size_t>(sp_blocks.size(), 1) * sizeof(SpBlock)))) return rc;
    if (!sp_blocks.empty())
        HIP_TRY(hipMemcpy(pl->d_sp_blocks.p, sp_blocks.data(), sp_blocks.size() * sizeof(SpBlock), hipMemcpyHostToDevice));
    if (!pl->tile_units.empty())
        HIP_TRY(hipMemcpy(pl->d_tile_units.p, pl->tile_units.data(), pl->tile_units.size() * sizeof(UnitDev),
                          hipMemcpyHostToDevice));
    if (!table.empty())
        HIP_TRY(hipMemcpy(pl->d_table.p, table.data(), table.size() * sizeof(NodeRec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl->d_stages.p, pl->stages.data(), pl->stages.size() * sizeof(StageDev), hipMemcpyHostToDevice));
    if (!pl->units.empty())
        HIP_TRY(hipMemcpy(pl->d_units.p, pl->units.data(), pl->units.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    return VJ_OK;
}

// (Re)lay out the per-scale queue segments for `frames` frames in flight.
static int layout_queues(Plan* pl, int frames, uint64_t* total_entries) {
    // a scale's segment: Q_PARTS parts, part x takes the windows of frames with frame * Q_PARTS / frames == x,
    // i.e. at most ceil(frames / Q_PARTS) frames (fewer frames than parts: one part per frame, the rest unused)
    const uint64_t frames_per_part = frames >= (int)Q_PARTS ? ((uint64_t)frames + Q_PARTS - 1) / Q_PARTS : 1;
    uint64_t base = 0;
    for (ScaleDev& sd : pl->scales) {
        sd.q_base = (uint32_t)base;
        sd.q_cap = (uint32_t)((uint64_t)sd.nwin * frames_per_part);
        base += (uint64_t)sd.nwin * frames_per_part * std::min<uint64_t>(Q_PARTS, (uint64_t)frames);
    }
    *total_entries = base;
    if (base > 0xffffffffull) {
        set_error("survivor queue needs %llu entries", (unsigned long long)base);
        return VJ_ERR_LIMIT;
    }
    if (pl->frames_q != frames && !pl->scales.empty()) {
        HIP_TRY(hipMemcpy(pl->d_scales.p, pl->scales.data(), pl->scales.size() * sizeof(ScaleDev),
                          hipMemcpyHostToDevice));
        pl->frames_q = frames;
    }
    return VJ_OK;
}

// One or two SMALL frames per call are bound by the latency of the gather chain (most of their scales are "large" for the
// batch thresholds: a 320 x 240 frame keeps two scales on tiles), and a tile with few windows is still far cheaper than
// the thin waves of the queue pass: 320 x 240 0.78 -> 0.46 ms with tiles of >= 64 windows, 640 x 480 0.80 -> 0.76 with
// >= 256 (tools/small_frames.py).  Batches keep the thresholds that make their tiles efficient (tools/many_small.py).
static int small_frame_class(const vj_env* e, int W, int H, int n_frames) {
    if (e->tile_thresholds_set || n_frames > 2) return 0;
    const long long px = (long long)W * H;
    return px <= 115200 ? 2 : px <= 460800 ? 1 : 0;
}

// The workload whose chain balance is being found (or was found) by feedback: batches of >= 8 frames through vj_detect.
static vj_env::BalanceKey balance_key(const vj_env* e, const vj_cascade* c, int W, int H, const vj_params& p, int n_frames) {
    return vj_env::BalanceKey(vj_env::PlanKey(c->content_hash, W, H, p.min_w, p.min_h, p.max_w, p.max_h, f2u(p.scale_factor), p.scale_mask[0],
                                              p.scale_mask[1], p.flags & (VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW | VJ_FLAG_GRID_F64 | VJ_FLAG_TILTED_AS_UPRIGHT), 0u),
                              e->balance_class(n_frames));
}

static vj_env::Balance* balance_of(vj_env* e, const vj_cascade* c, int W, int H, const vj_params& p, int n_frames, bool create) {
    // (calls of at least eight frames or sixteen megapixels: below that a call is a millisecond and its time says little)
    if (!e->auto_balance || e->tile_split_set || !e->concurrent || n_frames >= (1 << 20) ||
        (n_frames < 8 && (uint64_t)W * (uint64_t)H * (uint64_t)n_frames < 16000000ull))
        return nullptr;
    const vj_env::BalanceKey key = balance_key(e, c, W, H, p, n_frames);
    auto it = e->balance.find(key);
    if (it != e->balance.end()) {
        it->second.last_used = ++e->balance_tick;
        return &it->second;
    }
    if (!create) return nullptr;
    while (e->balance.size() >= 256) {   // bounded: the least recently used workload goes (a frozen result is cheap to find again)
        auto lru = e->balance.begin();
        for (auto i = e->balance.begin(); i != e->balance.end(); ++i)
            if (i->second.last_used < lru->second.last_used) lru = i;
        e->balance.erase(lru);
    }
    vj_env::Balance b;
    b.cur = b.best = e->split_for(n_frames, p);
    b.n_ref = n_frames;
    b.last_used = ++e->balance_tick;
    return &(e->balance[key] = b);
}

// What a call of n_frames frames runs: the candidate under test when it is one of the workload's sampling calls (the class's
// reference size, search not finished), else the best split known.
struct BalanceChoice {
    float split;
    int thr;
    bool candidate;
};
static BalanceChoice balance_choice(const vj_env::Balance* b, int n_frames) {
    if (b->phase != 3 && n_frames == b->n_ref) return BalanceChoice{b->cur, b->thr, b->cur != b->best || b->phase == 4};
    return BalanceChoice{b->best, b->phase == 4 ? 0 : b->thr, false};
}

// Called once per vj_detect / vj_detect_chain call of a balanced workload, before its plan is looked up: bookkeeping, and the
// reference size follows the traffic (a class whose reference size stopped coming — 32 calls of other sizes — restarts the
// measurement of its current best at the size that does come; what was found so far stays the starting point).
static void balance_touch(vj_env::Balance* b, int n_frames) {
    ++b->calls_total;
    if (b->phase == 3) return;
    if (n_frames == b->n_ref) {
        b->off_ref = 0;
    } else if (++b->off_ref > 32) {
        b->n_ref = n_frames;
        b->off_ref = 0;
        b->cur = b->best;
        b->phase = 0;
        b->samples = 0;
        b->moved = 0;
    }
    if (balance_choice(b, n_frames).candidate) ++b->calls_on_candidate;
}

// One more measured call (cascade kernels' time per frame, uncounted variants, the class's reference batch size) of a workload
// that is still being balanced.  Three calls per candidate — the first unrated (a new plan's tables were just uploaded: the
// device idled), the faster of the other two counts — or two when both the unrated call and the first rated one were already
// more than 3 % slower than the best: such a candidate is dropped at once.  Steps of a quarter of a scale; a move needs 0.7 %;
// at most 40 calls per workload.
static void balance_report(vj_env::Balance* b, float ms, bool try_thresholds) {
    if (!b || b->phase == 3 || !(ms > 0.0f)) return;
    ++b->calls;
    bool early_drop = false;
    if (++b->samples == 1) {
        b->cand_ms = 1e30f;
        b->first_slow = b->phase != 0 && ms > b->best_ms * 1.03f;
        return;
    }
    b->cand_ms = std::min(b->cand_ms, ms);
    if (b->samples == 2 && b->first_slow && b->phase != 0 && ms > b->best_ms * 1.03f) early_drop = true;
    if (b->samples < 3 && !early_drop) return;
    b->samples = 0;
    const float step = 0.25f, max_split = 3.0f;
    const int max_calls = 40;
    // the split is settled: one more candidate — the same split with the lower tile thresholds (how many scales the tile chain
    // can take at all: the better value depends on the frames' content, profiles/r03_notes.md #6e) — then the search ends
    // — and before that one probe a whole scale further: the response is not always convex (a scale split between the two
    // chains can cost more than the same scale moved entirely), and a climb in quarters stops at the first rise
    auto freeze = [&]() {
        b->cur = b->best;
        if (!b->far_tried && b->best + 1.0f <= max_split && b->calls <= max_calls) {
            b->far_tried = true;
            b->cur = b->best + 1.0f;
            b->phase = 5;
        } else if (try_thresholds && !b->thr_tried && b->calls <= max_calls) {
            b->thr_tried = true;
            b->thr = 1;
            b->phase = 4;
        } else {
            b->phase = 3;
        }
    };
    if (b->phase == 4) {
        if (b->cand_ms < b->best_ms * 0.993f) b->best_ms = b->cand_ms;
        else b->thr = 0;
        b->phase = 3;
        return;
    }
    if (b->phase == 5) {
        if (b->cand_ms < b->best_ms * 0.993f) {   // the far side is better: climb on from there
            b->best = b->cur;
            b->best_ms = b->cand_ms;
            b->moved = 0;
            b->phase = 1;
            if (b->best + step <= max_split) b->cur = b->best + step;
            else { b->phase = 2; b->cur = b->best - step; }
        } else {
            freeze();
        }
        return;
    }
    if (b->phase == 0) {
        b->best_ms = b->cand_ms;
        b->phase = 1;
        if (b->best + step <= max_split) b->cur = b->best + step;
        else if (b->best >= step) { b->phase = 2; b->cur = b->best - step; }
        else freeze();
        return;
    }
    const bool better = b->cand_ms < b->best_ms * 0.993f;
    if (better) {
        b->best = b->cur;
        b->best_ms = b->cand_ms;
        b->moved = 1;
        const float next = b->phase == 1 ? b->cur + step : b->cur - step;
        if (next < 0.0f || next > max_split || b->calls > max_calls) freeze();
        else b->cur = next;
    } else if (b->phase == 1 && !b->moved && b->best >= step) {
        b->phase = 2;
        b->cur = b->best - step;
    } else {
        freeze();
    }
}

// The found balances as text, one workload per line, so that another environment (or process) can start from them:
//   vjbal1 <cascade content hash> W H min_w min_h max_w max_h <scale factor bits> <mask0> <mask1> <flags> <class> <split> <thr>
static int balance_export(const vj_env* e, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) {
        set_error("cannot write %s", path);
        return VJ_ERR_IO;
    }
    for (const auto& kv : e->balance) {
        const vj_env::PlanKey& k = std::get<0>(kv.first);
        const vj_env::Balance& b = kv.second;
        // (the last four fields are for the reader of the file: search state — 0 = nothing measured yet, 3 = finished —, calls of
        // the workload, those that ran a split other than the best known, calls the search measured; import skips state 0)
        fprintf(f, "vjbal1 %llx %d %d %d %d %d %d %x %llx %llx %x %d %.4f %d %d %u %u %d\n", (unsigned long long)std::get<0>(k), std::get<1>(k),
                std::get<2>(k), std::get<3>(k), std::get<4>(k), std::get<5>(k), std::get<6>(k), std::get<7>(k), (unsigned long long)std::get<8>(k),
                (unsigned long long)std::get<9>(k), std::get<10>(k), std::get<1>(kv.first), (double)b.best, b.phase == 4 ? 0 : b.thr, b.phase,
                b.calls_total, b.calls_on_candidate, b.calls);
    }
    fclose(f);
    return VJ_OK;
}

static int balance_import(vj_env* e, const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) {
        set_error("cannot read %s", path);
        return VJ_ERR_IO;
    }
    char line[512];
    int n_bad = 0;
    while (fgets(line, sizeof(line), f)) {
        unsigned long long hash, m0, m1;
        int W, H, a, b2, c2, d, cls, thr, state = 3;
        unsigned sf, flags;
        double split;
        if (line[0] == '#' || line[0] == '\n') continue;
        if (sscanf(line, "vjbal1 %llx %d %d %d %d %d %d %x %llx %llx %x %d %lf %d %d", &hash, &W, &H, &a, &b2, &c2, &d, &sf, &m0, &m1, &flags, &cls,
                   &split, &thr, &state) < 14 || !(split >= 0.0 && split <= 3.0) || cls < 1) {
            ++n_bad;
            continue;
        }
        if (state == 0) continue;
        vj_env::Balance b;
        b.cur = b.best = (float)split;
        b.thr = thr ? 1 : 0;
        b.phase = 3;          // frozen: an imported workload does not search again ("auto_balance" "reset" forgets it)
        b.thr_tried = b.far_tried = true;
        b.last_used = ++e->balance_tick;
        e->balance[vj_env::BalanceKey(vj_env::PlanKey((uint64_t)hash, W, H, a, b2, c2, d, (uint32_t)sf, (uint64_t)m0, (uint64_t)m1, (uint32_t)flags, 0u), cls)] = b;
    }
    fclose(f);
    if (n_bad) {
        set_error("%s: %d lines are not balance records", path, n_bad);
        return VJ_ERR_PARSE;
    }
    return VJ_OK;
}

static int get_plan(vj_env* e, const vj_cascade* c, int W, int H, const vj_params& p, Plan** out, int n_frames = 1 << 20) {
    const vj_env::Balance* bal = balance_of(e, c, W, H, p, n_frames, false);
    const BalanceChoice choice = bal ? balance_choice(bal, n_frames) : BalanceChoice{e->split_for(n_frames, p), 0, false};
    const float split = choice.split;
    int small = small_frame_class(e, W, H, n_frames);
    if (small == 0 && choice.thr == 1 && !e->tile_thresholds_set) small = 3;   // the feedback's lower thresholds
    // (frames of 720p and more: below that the queue pass costs nothing — 640 x 480 0.473 / 0.476 ms with / without it, 320 x 240 the same)
    const bool one_pass = n_frames <= e->one_pass_max_frames && (uint64_t)W * (uint64_t)H >= 800000ull;
    vj_env::PlanKey key(c->uid, W, H, p.min_w, p.min_h, p.max_w, p.max_h, f2u(p.scale_factor), p.scale_mask[0],
                        p.scale_mask[1], (p.flags & (VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW | VJ_FLAG_GRID_F64 | VJ_FLAG_TILTED_AS_UPRIGHT)) | ((uint32_t)small << 8) | ((uint32_t)one_pass << 12), f2u(split));
    auto it = e->plans.find(key);
    if (it != e->plans.end()) {
        it->second->last_used = ++e->plan_tick;
        *out = it->second.get();
        return VJ_OK;
    }
    // bounded cache: release the least recently used plans first (their kernels may still be running)
    if ((int)e->plans.size() >= std::max(2, e->plan_cache_max)) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->stream2) HIP_TRY(hipStreamSynchronize(e->stream2));
        while ((int)e->plans.size() >= std::max(2, e->plan_cache_max)) {
            auto lru = e->plans.begin();
            for (auto i = e->plans.begin(); i != e->plans.end(); ++i)
                if (i->second->last_used < lru->second->last_used) lru = i;
            lru->second->release_device();
            e->plans.erase(lru);
        }
    }
    auto pl = std::make_unique<Plan>();
    TileThresholds th{e->tile_min_windows, e->tile_accept_windows, e->tile_max_dwords_per_window};
    if (small == 2) th = TileThresholds{64, 64, 8000};
    else if (small == 1) th = TileThresholds{256, 256, 2000};
    else if (small == 3) th = TileThresholds{std::min(384, th.min_windows), std::min(384, th.accept_windows), th.max_dwords_per_window};
    int rc = build_plan(e, *c, W, H, p, pl.get(), split, th, one_pass);
    if (rc) {
        pl->release_device();
        return rc;
    }
    pl->last_used = ++e->plan_tick;
    *out = pl.get();
    e->plans[key] = std::move(pl);
    return VJ_OK;
}

int Lane::create() {
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    for (auto& x : pass_ev) HIP_TRY(hipEventCreate(&x));
    for (auto& x : launch_ev) HIP_TRY(hipEventCreate(&x));
    HIP_TRY(hipEventCreateWithFlags(&upload_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&integral_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    h_pinned_bytes = (size_t)1 << 20;   // counters block (~100 KiB) + the first detections
    HIP_TRY(hipHostMalloc(&h_pinned, h_pinned_bytes, hipHostMallocDefault));
    return VJ_OK;
}

void Lane::destroy() {
    for (DevBuf* b : {&d_gray, &d_counts, &d_det}) b->release();
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (h_stage) (void)hipHostFree(h_stage);
    h_pinned = h_stage = nullptr;
    for (auto& x : ev) if (x) (void)hipEventDestroy(x);
    for (auto& x : pass_ev) if (x) (void)hipEventDestroy(x);
    for (auto& x : launch_ev) if (x) (void)hipEventDestroy(x);
    for (hipEvent_t* x : {&upload_done, &integral_done, &done}) if (*x) (void)hipEventDestroy(*x);
}

int ensure_image_buffers(vj_env* e, int W, int H, int frames, bool need_gray, int channels, Lane* lane) {
    if (!lane) lane = &e->lane0;
    const size_t fe = frame_elems_for(W, H);
    const uint32_t n_bands = ((uint32_t)H + BAND_ROWS - 1) / BAND_ROWS;
    const uint32_t band_pitch = ((uint32_t)W + 3u) & ~3u;
    int rc;
    const size_t sum_bytes = fe * 4 * (size_t)frames, sq_bytes = fe * 8 * (size_t)frames;
    const bool grow = sum_bytes > e->d_sum.cap || sq_bytes > e->d_sqsum.cap;
    if ((rc = e->d_sum.ensure(sum_bytes))) return rc;
    if ((rc = e->d_sqsum.ensure(sq_bytes))) return rc;
    (void)grow;
    if (need_gray) {
        const size_t gstride = ((size_t)W * (size_t)channels + 3) & ~(size_t)3;
        if ((rc = lane->d_gray.ensure(gstride * (size_t)H * (size_t)frames))) return rc;
    }
    const size_t band_elems = (size_t)frames * n_bands * band_pitch;
    if ((rc = e->d_band_sum.ensure(band_elems * 4))) return rc;
    if ((rc = e->d_band_sq.ensure(band_elems * 4))) return rc;
    if ((rc = e->d_band_sqp.ensure(band_elems * 8))) return rc;
    return VJ_OK;
}

// Enqueue the three integral kernels for `frames` frames already on the device.
int enqueue_integral(vj_env* e, const uint8_t* d_gray, size_t frame_bytes, int stride, int W, int H, int frames,
                     int channels) {
    IntegralArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.channels = (uint32_t)channels;
    ia.gray = d_gray;
    ia.gray_frame_bytes = frame_bytes;
    ia.gray_stride = (uint32_t)stride;
    ia.width = (uint32_t)W;
    ia.height = (uint32_t)H;
    ia.n_frames = (uint32_t)frames;
    ia.n_bands = ((uint32_t)H + BAND_ROWS - 1) / BAND_ROWS;
    ia.band_pitch = ((uint32_t)W + 3u) & ~3u;
    ia.band_sum = (uint32_t*)e->d_band_sum.p;
    ia.band_sq = (uint32_t*)e->d_band_sq.p;
    ia.band_sq_prefix = (uint64_t*)e->d_band_sqp.p;
    ia.sum = (uint32_t*)e->d_sum.p;
    ia.sqsum = (uint64_t*)e->d_sqsum.p;
    ia.frame_elems = frame_elems_for(W, H);
    ia.rows_mode = (uint32_t)e->integral_rows_mode;
    // the slack rows after row H (and the alignment tail) must read as zero
    // (the kernels never write there, so once per buffer layout is enough)
    const size_t used = (size_t)(W + 1) * (size_t)(H + 1);
    const bool zeroed = e->slack_w == W && e->slack_h == H && e->slack_frames >= frames && e->slack_sum == e->d_sum.p &&
                        e->slack_sq == e->d_sqsum.p;
    for (int f = 0; f < frames && !zeroed; ++f) {
        HIP_TRY(hipMemsetAsync((uint32_t*)e->d_sum.p + (size_t)f * ia.frame_elems + used, 0,
                               (ia.frame_elems - used) * 4, e->stream));
        HIP_TRY(hipMemsetAsync((uint64_t*)e->d_sqsum.p + (size_t)f * ia.frame_elems + used, 0,
                               (ia.frame_elems - used) * 8, e->stream));
    }
    if (!zeroed) {
        e->slack_w = W;
        e->slack_h = H;
        e->slack_frames = frames;
        e->slack_sum = e->d_sum.p;
        e->slack_sq = e->d_sqsum.p;
    }
    int hrc = launch_integral(ia, e->stream);
    if (hrc) {
        set_error("integral launch failed: %s", hipGetErrorString((hipError_t)hrc));
        return VJ_ERR_HIP;
    }
    return VJ_OK;
}

// The tilted integral of `frames` frames already on the device, into e->d_tilted (same geometry as d_sum).
int enqueue_tilted(vj_env* e, const uint8_t* d_gray, size_t frame_bytes, int stride, int W, int H, int frames, int channels) {
    const uint32_t fe = frame_elems_for(W, H);
    int rc;
    if ((rc = e->d_tilted.ensure((size_t)fe * 4u * (size_t)frames))) return rc;
    // rows past H (the slack rows) must read as zero, like d_sum's
    HIP_TRY(hipMemsetAsync(e->d_tilted.p, 0, (size_t)fe * 4u * (size_t)frames, e->stream));
    TiltedArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.gray = d_gray;
    ta.gray_frame_bytes = frame_bytes;
    ta.gray_stride = (uint32_t)stride;
    ta.channels = (uint32_t)channels;
    ta.width = (uint32_t)W;
    ta.height = (uint32_t)H;
    ta.n_frames = (uint32_t)frames;
    ta.frame_elems = fe;
    ta.tilted = (uint32_t*)e->d_tilted.p;
    int hrc;
    if (e->tilted_bands) {
        const size_t n_bands = ((size_t)H + 7u) / 8u;
        if ((rc = e->d_tilt_diag.ensure((size_t)frames * n_bands * 2u * (size_t)(W + H) * 4u))) return rc;
        if ((rc = e->d_tilt_col.ensure((size_t)frames * n_bands * (size_t)(W + 1) * 4u))) return rc;
        hrc = launch_tilted_bands(ta, (uint32_t*)e->d_tilt_diag.p, (uint32_t*)e->d_tilt_col.p, e->stream);
    } else {
        hrc = launch_tilted_integral(ta, e->stream);
    }
    if (hrc) {
        set_error("tilted integral launch failed (width %d): %s", W, hipGetErrorString((hipError_t)hrc));
        return hrc == (int)hipErrorInvalidValue ? VJ_ERR_LIMIT : VJ_ERR_HIP;
    }
    return VJ_OK;
}

// Upload host frames (or gather strided device frames) into d_gray with a 4-byte
// aligned pitch.  Returns the device pointer / pitch the integral kernels should use.
int image_channels(const vj_image& im) { return im.channels <= 1 ? 1 : im.channels; }

int stage_frames(vj_env* e, const vj_image* frames, int n, int W, int H, const uint8_t** d_ptr, size_t* frame_bytes,
                 int* stride, Lane* lane, hipStream_t copy_stream) {
    if (!lane) lane = &e->lane0;
    hipStream_t cs = copy_stream ? copy_stream : e->stream;
    const size_t row_bytes = (size_t)W * (size_t)image_channels(frames[0]);
    bool all_dev = true, contiguous = true;
    for (int i = 0; i < n; ++i) {
        if (!frames[i].on_device) all_dev = false;
        if (i > 0 && (frames[i].stride != frames[0].stride ||
                      frames[i].data != frames[0].data + (size_t)i * (size_t)frames[0].stride * (size_t)H))
            contiguous = false;
    }
    if (all_dev && contiguous) {  // use the caller's device batch in place
        *d_ptr = frames[0].data;
        *stride = frames[0].stride;
        *frame_bytes = (size_t)frames[0].stride * (size_t)H;
        return VJ_OK;
    }
    const size_t gstride = (row_bytes + 3) & ~(size_t)3;
    bool all_host = true;
    for (int i = 0; i < n; ++i) all_host = all_host && !frames[i].on_device;
    if (all_host && contiguous && (size_t)frames[0].stride == gstride && n > 1) {
        // a batch that is one block on the host with the device's pitch (a numpy array of n frames): ONE copy instead of n
        // (2048 frames of 100 x 100: 24 ms of per-frame copy calls)
        const size_t bytes = gstride * (size_t)H * (size_t)n;
        const void* src = frames[0].data;
        if (copy_stream) {   // streams: a true DMA — page-locked memory as it is, pageable memory through the staging buffer
            hipPointerAttribute_t at;
            const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
            if (!pinned) {
                (void)hipGetLastError();
                if (lane->h_stage_bytes < bytes) {
                    if (lane->h_stage) (void)hipHostFree(lane->h_stage);
                    lane->h_stage = nullptr;
                    lane->h_stage_bytes = 0;
                    HIP_TRY(hipHostMalloc(&lane->h_stage, bytes, hipHostMallocDefault));
                    lane->h_stage_bytes = bytes;
                }
                memcpy(lane->h_stage, src, bytes);
                src = lane->h_stage;
            }
        }
        HIP_TRY(hipMemcpyAsync(lane->d_gray.p, src, bytes, hipMemcpyHostToDevice, cs));
        *d_ptr = (const uint8_t*)lane->d_gray.p;
        *stride = (int)gstride;
        *frame_bytes = gstride * (size_t)H;
        return VJ_OK;
    }
    if (all_host && n >= 4 && gstride * (size_t)H <= ((size_t)256 << 10)) {
        // many small frames in separate host buffers: gather them in the page-locked staging buffer (at the device's pitch)
        // and send one block — a copy call per frame costs more than copying a few KB twice
        const size_t fb = gstride * (size_t)H, bytes = fb * (size_t)n;
        if (lane->h_stage_bytes < bytes) {
            if (lane->h_stage) (void)hipHostFree(lane->h_stage);
            lane->h_stage = nullptr;
            lane->h_stage_bytes = 0;
            HIP_TRY(hipHostMalloc(&lane->h_stage, bytes, hipHostMallocDefault));
            lane->h_stage_bytes = bytes;
        }
        for (int i = 0; i < n; ++i) {
            uint8_t* st = (uint8_t*)lane->h_stage + (size_t)i * fb;
            if ((size_t)frames[i].stride == gstride) memcpy(st, frames[i].data, fb);
            else
                for (int y = 0; y < H; ++y) memcpy(st + (size_t)y * gstride, frames[i].data + (size_t)y * (size_t)frames[i].stride, row_bytes);
        }
        HIP_TRY(hipMemcpyAsync(lane->d_gray.p, lane->h_stage, bytes, hipMemcpyHostToDevice, cs));
        *d_ptr = (const uint8_t*)lane->d_gray.p;
        *stride = (int)gstride;
        *frame_bytes = fb;
        return VJ_OK;
    }
    for (int i = 0; i < n; ++i) {
        uint8_t* dst = (uint8_t*)lane->d_gray.p + (size_t)i * gstride * (size_t)H;
        const uint8_t* src = frames[i].data;
        size_t src_stride = (size_t)frames[i].stride;
        if (!frames[i].on_device && (copy_stream || src_stride != gstride)) {
            // streams: the copy must be a true DMA to overlap the kernels — page-locked memory (vj_host_alloc) goes as
            // it is, pageable memory through this lane's pinned staging buffer.  Blocking calls: a pageable frame whose
            // row stride is not the device pitch (a width that is not a multiple of 4) would be a 2-D copy from pageable
            // memory, which the runtime does row by row (1921 x 1081: 7.5 ms instead of 0.2): re-pitch it in the
            // staging buffer and send it as one block
            hipPointerAttribute_t at;
            const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
            if (!pinned) {
                (void)hipGetLastError();
                const size_t need = gstride * (size_t)H * (size_t)n;
                if (lane->h_stage_bytes < need) {
                    if (lane->h_stage) (void)hipHostFree(lane->h_stage);
                    lane->h_stage = nullptr;
                    lane->h_stage_bytes = 0;
                    HIP_TRY(hipHostMalloc(&lane->h_stage, need, hipHostMallocDefault));
                    lane->h_stage_bytes = need;
                }
                uint8_t* st = (uint8_t*)lane->h_stage + (size_t)i * gstride * (size_t)H;
                for (int y = 0; y < H; ++y) memcpy(st + (size_t)y * gstride, src + (size_t)y * src_stride, row_bytes);
                src = st;
                src_stride = gstride;
            }
        }
        HIP_TRY(hipMemcpy2DAsync(dst, gstride, src, src_stride, row_bytes, (size_t)H,
                                 frames[i].on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, cs));
    }
    *d_ptr = (const uint8_t*)lane->d_gray.p;
    *stride = (int)gstride;
    *frame_bytes = gstride * (size_t)H;
    return VJ_OK;
}

struct RawDet {
    int frame;
    uint32_t slot, x, y;
};

// Where the counters of one batch live: [MAX_PASSES][MAX_SCALES][Q_PARTS] queue counts | det_count | pad | one
// stage_entered[VJ_MAX_STAGES] (u64) array per kernel launch (array 0 is spare) | the counters of a region-of-interest
// pass (vj_detect_chain): regions, units, invalid regions, unit ticket, detections, pad | its stage_entered array | the queue
// counts of the tiles' own queue set (stage trees, enqueue_cascade).
struct CountsLayout {
    static constexpr size_t q_counts = (size_t)MAX_SCALES * Q_PARTS;   // counters of one queue: [scale][part]
    static constexpr size_t stage_off_u32 = MAX_PASSES * q_counts + 2;
    static constexpr size_t roi_off_u32 = stage_off_u32 + (size_t)(1 + VJ_MAX_LAUNCHES) * VJ_MAX_STAGES * 2;
    static constexpr size_t q2_off_u32 = roi_off_u32 + 8 + (size_t)VJ_MAX_STAGES * 2;   // second set of queue counters (stage trees)
    static constexpr size_t bytes = (q2_off_u32 + MAX_PASSES * q_counts) * sizeof(uint32_t);
};

// Upload (or adopt) the frames of one batch and enqueue its integral images.  `copy_stream` != null: the upload runs
// on that stream and the integral waits for it (vj_stream); else everything is ordered on the environment's stream.
static int enqueue_prepare(vj_env* e, Lane* L, Plan* pl, const vj_image* frames, int nf, int W, int H, bool fixed_queue_layout,
                           hipStream_t copy_stream) {
    int rc;
    const int channels = image_channels(frames[0]);
    if ((rc = ensure_image_buffers(e, W, H, nf, true, channels, L))) return rc;
    uint64_t q_entries = 0;
    if (!fixed_queue_layout) {
        if ((rc = layout_queues(pl, nf, &q_entries))) return rc;
        const size_t n_pass = pl->pass_bounds.size() - 1;
        for (size_t ps = 1; ps < n_pass; ++ps)
            if ((rc = e->d_q[ps].ensure(std::max<uint64_t>(q_entries, 1) * sizeof(QEntry)))) return rc;
    }
    if ((rc = L->d_counts.ensure(CountsLayout::bytes))) return rc;
    if (L->det_cap == 0) {
        L->det_cap = e->det_cap_init;
        if ((rc = L->d_det.ensure((size_t)L->det_cap * sizeof(DetEntry)))) return rc;
    }
    const uint8_t* d_gray;
    size_t gray_frame_bytes;
    int gray_stride;
    if ((rc = stage_frames(e, frames, nf, W, H, &d_gray, &gray_frame_bytes, &gray_stride, L, copy_stream))) return rc;
    if (copy_stream) {
        HIP_TRY(hipEventRecord(L->upload_done, copy_stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, L->upload_done, 0));
    }
    L->src_gray = d_gray;
    L->src_frame_bytes = gray_frame_bytes;
    L->src_stride = gray_stride;
    L->src_channels = channels;
    HIP_TRY(hipEventRecord(L->ev[0], e->stream));
    if ((rc = enqueue_integral(e, d_gray, gray_frame_bytes, gray_stride, W, H, nf, channels))) return rc;
    HIP_TRY(hipEventRecord(L->ev[1], e->stream));
    HIP_TRY(hipEventRecord(L->integral_done, e->stream));   // the lane's frame buffer may be overwritten from here on
    L->nf = nf;
    return VJ_OK;
}

// Enqueue every cascade launch of the batch whose integral images are (being) computed, then the asynchronous
// read-back of its counters and first detections.  Nothing here waits for the device.
static int enqueue_cascade(vj_env* e, Lane* L, Plan* pl, int W, int /*H*/, const vj_params& p) {
    int rc;
    const int nf = L->nf;
    const size_t n_pass = pl->pass_bounds.size() - 1;
    const size_t counts_bytes = CountsLayout::bytes;
    constexpr size_t q_counts = CountsLayout::q_counts;
    uint32_t* d_qcount[MAX_PASSES];
    for (int ps = 0; ps < MAX_PASSES; ++ps) d_qcount[ps] = (uint32_t*)L->d_counts.p + ps * q_counts;
    uint32_t* d_qcount2[MAX_PASSES];   // the tiles' own queue set (stage trees: see split_sets below)
    for (int ps = 0; ps < MAX_PASSES; ++ps) d_qcount2[ps] = (uint32_t*)L->d_counts.p + CountsLayout::q2_off_u32 + ps * q_counts;
    uint32_t* d_det_count = (uint32_t*)L->d_counts.p + MAX_PASSES * q_counts;
    unsigned long long* d_stage_entered = (unsigned long long*)((uint32_t*)L->d_counts.p + CountsLayout::stage_off_u32);
    const bool count = (p.flags & VJ_FLAG_COUNTERS) != 0;
    const int n_blocks = std::max(1, e->n_cu * e->blocks_per_cu);
    {
        HIP_TRY(hipMemsetAsync(L->d_counts.p, 0, counts_bytes, e->stream));
        HIP_TRY(hipEventRecord(L->ev[2], e->stream));
        CascadeArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.sum = (const uint32_t*)e->d_sum.p;
        ca.sqsum = (const uint64_t*)e->d_sqsum.p;
        ca.table = (const uint32_t*)pl->d_table.p;
        ca.scales = (const ScaleDev*)pl->d_scales.p;
        ca.stages = (const StageDev*)pl->d_stages.p;
        ca.units = (const UnitDev*)pl->d_units.p;
        ca.n_units = (uint32_t)pl->units.size();
        ca.tile_units = (const UnitDev*)pl->d_tile_units.p;
        ca.n_tile_units = pl->block_first;   // staged tiles; the unstaged blocks follow them in the list
        ca.n_frames = (uint32_t)nf;
        ca.n_scales = (uint32_t)pl->scales.size();
        ca.frame_elems = pl->frame_elems;
        ca.sum_bytes = (uint32_t)((uint64_t)pl->frame_elems * 4u * (uint64_t)nf);
        ca.stride = (uint32_t)W + 1u;
        // waves per workgroup of the gather chain: four for batches, three for single frames and for stage trees (their
        // workgroups carry a second LDS array); the launchers clamp, the kernels read blockDim
        ca.gather_waves = (uint32_t)e->gather_waves_for(nf, pl->general);
        ca.total_waves = (uint32_t)n_blocks * ca.gather_waves;
        ca.det = (DetEntry*)L->d_det.p;
        ca.det_count = d_det_count;
        ca.det_cap = L->det_cap;
        ca.signed_mean = (p.flags & VJ_FLAG_SIGNED_MEAN) ? 1u : 0u;
        ca.stage_entered = d_stage_entered;
        ca.tree_ctr = d_stage_entered;       // array 0 is spare: [0], [1] = the first cascade's visited child nodes / their rectangles
        ca.n_pass = (uint32_t)n_pass;
        for (size_t ps = 0; ps <= n_pass; ++ps) ca.pass_begin[ps] = pl->pass_bounds[ps];
        for (size_t ps = 1; ps < n_pass; ++ps) {
            ca.q_pass[ps] = (QEntry*)e->d_q[ps].p;
            ca.q_pass_count[ps] = d_qcount[ps];
        }
        ca.tile_end = pl->general ? pl->general_prefix : (uint32_t)e->tile_end;   // stage trees: tiles run the linear prefix only
        ca.tile_min_lanes = (uint32_t)e->tile_min_lanes;
        ca.tile_repack_mask = e->tile_repack_mask;
        ca.tile_sp_begin = (!pl->general && (pl->sp_pad || pl->tree2)) ? (uint32_t)e->tile_sp_begin : 0xffffffffu;   // the finishes walk positions linearly: never on a stage tree
        ca.tree2 = pl->tree2 ? 1u : 0u;
        ca.identity_order = pl->general ? 0u : 1u;
        ca.n_seg = e->tile_segments ? pl->tile_n_seg : 0u;
        for (int k = 0; k < 4; ++k) ca.seg_end[k] = pl->tile_seg_end[k];
        ca.seg_chain = pl->tile_seg_chain;
        ca.tile_sp_pad = pl->sp_pad;
        ca.sp_blocks = (const SpBlock*)pl->d_sp_blocks.p;
        ca.n_sp_blocks = pl->n_sp_blocks;
        ca.tile_sp_max = (uint32_t)std::min(e->tile_sp_max, (int)TILE_SP_MAX_WINDOWS);
        ca.xcd_affinity = (uint32_t)e->xcd_affinity;
        ca.tile_finish = (uint32_t)e->tile_finish;
        ca.tile_ws_min = pl->sp_pad != 0u ? (uint32_t)e->tile_ws_min : 0u;   // no stump-parallel tables: wave-split to the end
        ca.tile_ws_max = (uint32_t)std::min(e->tile_ws_max, (int)TILE_WS_MAX_WINDOWS);
        ca.pos_mode = pl->pos_mode;
        ca.pos_tab = (const uint32_t*)pl->d_pos_tab.p;
        ca.gather_pairs = e->pairs_for(nf);
        ca.sp_tail_max = (uint32_t)std::max(0, std::min(e->sp_tail_max, 48));
        ca.max_stage_nodes = pl->max_stage_nodes;
        if (pl->skip_mode && pl->n_skip_units) {
            // the windows the reference's sequential CPU loop visits, as a bitmap: stage-0 verdict of every grid window,
            // then the parity recurrence; the passes below drop the unvisited windows while they enumerate the grid
            if ((rc = e->d_skip_bits.ensure((size_t)pl->skip_frame_words * 8u * (size_t)nf))) return rc;
            ca.skip_bits = (unsigned long long*)e->d_skip_bits.p;
            ca.skip_frame_words = pl->skip_frame_words;
            ca.skip_units = (const UnitDev*)pl->d_skip_units.p;
            ca.n_skip_units = pl->n_skip_units;
            ca.skip_segs = (const UnitDev*)pl->d_skip_segs.p;
            ca.n_skip_segs = pl->n_skip_segs;
            const int hrc = launch_skip_bitmap(ca, pl->trees, std::max(1, e->n_cu * 4), e->stream);
            if (hrc) {
                set_error("skip bitmap launch failed: %s", hipGetErrorString((hipError_t)hrc));
                return VJ_ERR_HIP;
            }
        }
        int launches = 0;
        std::vector<vj_launch>& linfo = L->linfo;
        linfo.clear();
        // every launch is bracketed by its own pair of events on the stream it runs on
        auto begin_launch = [&](int kind, int cls, uint32_t sb, uint32_t se, uint32_t lds, hipStream_t st) -> int {
            if (linfo.size() < VJ_MAX_LAUNCHES) HIP_TRY(hipEventRecord(L->launch_ev[2 * linfo.size()], st));
            vj_launch li;
            memset(&li, 0, sizeof(li));
            li.kind = kind;
            li.lds_class = cls;
            li.stage_begin = (int32_t)sb;
            li.stage_end = (int32_t)se;
            li.lds_bytes = lds;
            for (const ScaleDev& sd : pl->scales) {
                const bool in = kind == VJ_LAUNCH_QUEUE ||
                                (kind == VJ_LAUNCH_TILE && sd.tile_rw && sd.tile_row_end > 0 && (int)sd.tile_class == cls) ||
                                ((kind == VJ_LAUNCH_GRID || kind == VJ_LAUNCH_BLOCK) && sd.tile_row_end < sd.ny);
                if (in && sd.scale_idx < 128) li.scale_mask[sd.scale_idx >> 6] |= 1ull << (sd.scale_idx & 63);
            }
            linfo.push_back(li);
            return VJ_OK;
        };
        // the launch just begun counts into its own array
        auto launch_counters = [&]() -> unsigned long long* {
            return d_stage_entered + std::min<size_t>(linfo.size(), VJ_MAX_LAUNCHES) * VJ_MAX_STAGES;
        };
        auto end_launch = [&](hipStream_t st) -> int {
            if (linfo.size() <= VJ_MAX_LAUNCHES) HIP_TRY(hipEventRecord(L->launch_ev[2 * linfo.size() - 1], st));
            return VJ_OK;
        };
        auto pass_is_last = [&](size_t ps) { return pl->seg_last.empty() ? ps + 1 == n_pass : pl->seg_last[ps] != 0; };
        const bool general_kernel = pl->general && pl->seg_last.empty();   // run_stages_general finishes the tree
        // Band-major queue pass: a batch of a linear cascade whose gather chain is grid pass + ONE queue pass that only the grid
        // pass feeds (the tiles run the whole cascade themselves)
        const bool banded = e->q_band_px > 0 && !pl->unit_groups.empty() && !pl->general && n_pass == 2 && nf >= e->q_band_min_frames &&
                            e->tile_min_lanes == 0 && (uint32_t)e->tile_end >= pl->pass_bounds[1] && !(e->global_blocks && pl->n_block_units > 0 && pl->sp_pad != 0);
        if (banded) {
            if ((rc = e->d_run_table.ensure((size_t)nf * pl->units.size() * 8u))) return rc;
            ca.run_table = (uint32_t*)e->d_run_table.p;
        }
        auto queue_args = [&](size_t ps, int set = 0) {
            CascadeArgs qa = ca;
            DevBuf* dq = set ? e->d_q2 : e->d_q;
            uint32_t** qc = set ? d_qcount2 : d_qcount;
            qa.stage_begin = pl->pass_bounds[ps];
            qa.stage_end = pl->pass_bounds[ps + 1];
            const bool last = pass_is_last(ps);
            // pass ps reads queue ps (filled by pass ps-1 and by tiles that left at this boundary)
            // and appends its survivors to queue ps+1
            qa.q_in = (const QEntry*)dq[ps].p;
            qa.q_in_count = qc[ps];
            qa.q_ticket = qc[0] + ps * Q_PARTS;   // queue 0 does not exist: its counters serve as tickets
            qa.thin_pass_spread = e->thin_pass_spread ? 1u : 0u;
            // frame-major order inside a part: one slice per frame of the part's frame group (-1), or as configured
            qa.q_slices = e->q_slices >= 0 ? (uint32_t)std::max(1, e->q_slices)
                                           : (uint32_t)std::max(1, std::min(16, (nf + (int)Q_PARTS - 1) / (int)Q_PARTS));
            qa.min_chunk = (uint32_t)e->min_chunk;
            qa.wide_tail = (e->wide_tail < 0 ? nf <= 4 : e->wide_tail != 0) ? 1u : 0u;
            qa.q_out = last ? nullptr : (QEntry*)dq[ps + 1].p;
            qa.q_out_count = last ? nullptr : qc[ps + 1];
            if (banded && ps == 1) {
                qa.q_groups = (const UnitDev*)pl->d_unit_groups.p;
                qa.n_q_groups = (uint32_t)pl->unit_groups.size();
            }
            if (!pl->seg_fail.empty() && pl->seg_fail[ps] != 0) {   // stage tree: this segment's rejects continue
                qa.q_fail = (QEntry*)dq[pl->seg_fail[ps]].p;
                qa.q_fail_count = qc[pl->seg_fail[ps]];
            }
            return qa;
        };
        if (ca.n_units + ca.n_tile_units > 0) {
            int hrc = 0;
            // Two chains share the first part of the cascade and run CONCURRENTLY on two streams:
            //   A (e->stream) : the LDS-tile launches (LDS-bound)
            //   B (e->stream2): the global-gather first pass and the queue passes that only it feeds
            //                   (texture-address-bound) — every pass that begins before the stage at
            //                   which tiles hand over (tiles leave at one boundary when tile_min_lanes = 0)
            // then B joins A and the remaining queue passes run on A.
            const uint32_t handover = pl->general ? pl->general_prefix : std::max<uint32_t>(ca.tile_end, pl->pass_bounds[1]);
            size_t first_joint_pass = 1;
            if (e->tile_min_lanes == 0)
                while (first_joint_pass < n_pass && pl->pass_bounds[first_joint_pass] < handover) ++first_joint_pass;
            const bool use_blocks = e->global_blocks && pl->n_block_units > 0 && pl->sp_pad != 0;
            const bool two_streams = e->concurrent && pl->block_first > 0 && ca.n_units > 0;
            // Stage tree in chains (seg_last): the chains' queue passes would have to wait for BOTH the grid pass and the
            // tiles, because a crowded tile hands its windows to the same queues — and the tiles take longer than the grid
            // pass (4096 x 4096: 7.8 vs 4.8 ms, then 7.8 ms of queue passes).  So the tiles get a queue set of their own:
            // the grid pass's survivors go down the tree on stream B while the tiles still run, and after the join the same
            // passes run once more on what the tiles left (usually little: thin passes).
            const bool split_sets = two_streams && pl->general && !pl->seg_last.empty() && e->tree_split_queues && !use_blocks;
            if (split_sets) {
                for (size_t ps = 1; ps < n_pass; ++ps) {
                    if ((rc = e->d_q2[ps].ensure(e->d_q[ps].cap))) return rc;
                    ca.q_pass[ps] = (QEntry*)e->d_q2[ps].p;        // (the grid and queue passes take theirs from queue_args)
                    ca.q_pass_count[ps] = d_qcount2[ps];
                }
                first_joint_pass = n_pass;
            }
            hipStream_t sB = two_streams ? e->stream2 : e->stream;
            if (two_streams) {
                HIP_TRY(hipEventRecord(e->fork_ev, e->stream));
                HIP_TRY(hipStreamWaitEvent(e->stream2, e->fork_ev, 0));
            }
            HIP_TRY(hipEventRecord(L->pass_ev[0], e->stream));
            // chain A: tile launches
            auto chain_a = [&]() -> int {
            for (uint32_t ci = 0; ci < TILE_CLASSES && !hrc && ca.n_tile_units > 0; ++ci) {
                const uint32_t cls = e->tile_class_order ? TILE_CLASSES - 1u - ci : ci;
                const uint32_t n_cls = pl->class_first[cls + 1] - pl->class_first[cls];
                if (!n_cls) continue;
                CascadeArgs ta = ca;
                ta.tile_units = (const UnitDev*)pl->d_tile_units.p + pl->class_first[cls];
                ta.n_tile_units = n_cls;
                ta.tile_lds_bytes = pl->class_lds[cls];
                ta.tile_ticket = d_qcount[0] + (q_counts - 8u * (cls + 1u));   // queue 0 does not exist: its counters are free
                // workgroups per CU: what the LDS allows (160 KiB per CU), at most 4 x 8 waves
                const int per_cu = std::max(1, std::min(32 / TILE_WAVES, (int)(160u * 1024u / ta.tile_lds_bytes)));
                const int tb = (int)std::min<uint64_t>((uint64_t)n_cls * (uint64_t)nf, (uint64_t)e->n_cu * (uint64_t)per_cu);
                const uint32_t deepest = std::min<uint32_t>((uint32_t)pl->stages.size(), handover);
                if ((rc = begin_launch(VJ_LAUNCH_TILE, (int)cls, 0, deepest, ta.tile_lds_bytes, e->stream))) return rc;
                ta.stage_entered = launch_counters();
                hrc = launch_cascade_tile_pass(ta, pl->trees, count, true, std::max(1, tb), e->stream);
                if ((rc = end_launch(e->stream))) return rc;
            }
                return VJ_OK;
            };
            // chain B: grid pass, then the queue passes fed by it alone.  When the chains overlap it goes
            // first, with one small workgroup per CU (the texture-address unit it is bound by saturates at
            // one wave per SIMD), so that the tile workgroups find their LDS share next to it.
            const int b_blocks = two_streams ? std::max(1, e->n_cu * e->concurrent_blocks_per_cu) : n_blocks;
            auto chain_b = [&]() -> int {
            if (!hrc && use_blocks) {
                CascadeArgs ta = ca;
                ta.tile_units = (const UnitDev*)pl->d_tile_units.p + pl->block_first;
                ta.n_tile_units = pl->n_block_units;
                ta.tile_lds_bytes = pl->block_lds;
                ta.tile_ticket = d_qcount[0] + (q_counts - 8u * (TILE_CLASSES + 1u));
                const int per_cu = two_streams ? e->concurrent_blocks_per_cu : 2;
                const int tb = (int)std::min<uint64_t>((uint64_t)pl->n_block_units * (uint64_t)nf, (uint64_t)e->n_cu * (uint64_t)per_cu);
                const uint32_t deepest = std::min<uint32_t>((uint32_t)pl->stages.size(), handover);
                if ((rc = begin_launch(VJ_LAUNCH_BLOCK, 0, 0, deepest, ta.tile_lds_bytes, sB))) return rc;
                ta.stage_entered = launch_counters();
                hrc = launch_cascade_tile_pass(ta, false, count, false, std::max(1, tb), sB);
                if ((rc = end_launch(sB))) return rc;
            } else if (!hrc && ca.n_units > 0) {
                CascadeArgs ga = queue_args(0);
                ga.total_waves = (uint32_t)b_blocks * ga.gather_waves;
                if ((rc = begin_launch(VJ_LAUNCH_GRID, 0, ga.stage_begin, ga.stage_end, 0, sB))) return rc;
                ga.stage_entered = launch_counters();
                hrc = launch_cascade_pass(ga, true, pl->trees, n_pass == 1, count, general_kernel && n_pass == 1, b_blocks, sB);
                if ((rc = end_launch(sB))) return rc;
                for (size_t ps = 1; ps < first_joint_pass && !hrc; ++ps) {
                    CascadeArgs qa = queue_args(ps);
                    // (next to the tiles one workgroup per CU; a stage tree's chains are latency-bound: all of them)
                    const int qb = split_sets ? n_blocks : b_blocks;
                    qa.total_waves = (uint32_t)qb * qa.gather_waves;
                    if ((rc = begin_launch(VJ_LAUNCH_QUEUE, 0, qa.stage_begin, qa.stage_end, 0, sB))) return rc;
                    qa.stage_entered = launch_counters();
                    hrc = launch_cascade_pass(qa, false, pl->trees, pass_is_last(ps), count, false, qb, sB);
                    if ((rc = end_launch(sB))) return rc;
                }
            } else {
                first_joint_pass = 1;
            }
                return VJ_OK;
            };
            if (two_streams) {
                if ((rc = chain_b())) return rc;
                if ((rc = chain_a())) return rc;
            } else {
                if ((rc = chain_a())) return rc;
                if ((rc = chain_b())) return rc;
            }
            if (two_streams) {
                HIP_TRY(hipEventRecord(e->join_ev, e->stream2));
                HIP_TRY(hipStreamWaitEvent(e->stream, e->join_ev, 0));
            }
            for (size_t ps = 1; ps < n_pass && ps < VJ_MAX_PASSES && ps <= first_joint_pass; ++ps)
                HIP_TRY(hipEventRecord(L->pass_ev[ps], e->stream));
            // joint passes (split_sets: all of them once more, on the tiles' queue set)
            for (size_t ps = split_sets ? 1 : first_joint_pass; ps < n_pass && !hrc; ++ps) {
                if (ps > first_joint_pass && ps < VJ_MAX_PASSES) HIP_TRY(hipEventRecord(L->pass_ev[ps], e->stream));
                CascadeArgs qa = queue_args(ps, split_sets ? 1 : 0);
                if ((rc = begin_launch(VJ_LAUNCH_QUEUE, 0, qa.stage_begin, qa.stage_end, 0, e->stream))) return rc;
                qa.stage_entered = launch_counters();
                hrc = launch_cascade_pass(qa, false, pl->trees, pass_is_last(ps), count, general_kernel, n_blocks, e->stream);
                if ((rc = end_launch(e->stream))) return rc;
            }
            if (hrc) {
                set_error("cascade launch failed: %s", hipGetErrorString((hipError_t)hrc));
                return VJ_ERR_HIP;
            }
            launches = (int)n_pass;
        }
        if (n_pass <= VJ_MAX_PASSES && launches) HIP_TRY(hipEventRecord(L->pass_ev[n_pass], e->stream));
        HIP_TRY(hipEventRecord(L->ev[3], e->stream));
        L->launches = launches;
    }
    L->n_pass = n_pass;
    L->count = count;
    // read back the counters block and the first detections (nearly always all of them)
    HIP_TRY(hipMemcpyAsync(L->h_pinned, L->d_counts.p, counts_bytes, hipMemcpyDeviceToHost, e->stream));
    const size_t room = (L->h_pinned_bytes - ((counts_bytes + 255) & ~(size_t)255)) / sizeof(DetEntry);
    L->det_copied = (uint32_t)std::min<size_t>(room, L->det_cap);
    HIP_TRY(hipMemcpyAsync((char*)L->h_pinned + ((counts_bytes + 255) & ~(size_t)255), L->d_det.p, (size_t)L->det_copied * sizeof(DetEntry),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipEventRecord(L->done, e->stream));
    L->pending = true;
    return VJ_OK;
}

// Wait for the batch on lane L and decode it: detections appended to `dets` with frame numbers offset by f0.  When the
// detection buffer overflowed it is grown and the cascade passes run again — on the integral images still in place for a
// blocking call, after recomputing them from the lane's frames for a stream lane (the two lanes of a vj_stream share
// the environment's integral images, and the next batch's have replaced this one's by now).  The redo is ordered on the
// environment's stream behind everything already queued, and finished synchronously here.
static int finish_batch(vj_env* e, Lane* L, Plan* pl, int f0, int W, int H, const vj_params& p, std::vector<RawDet>* dets,
                        vj_counters* ctr, vj_timing* tm, std::vector<DetEntry>* raw_out = nullptr) {
    int rc;
    constexpr size_t q_counts = CountsLayout::q_counts;
    const size_t counts_bytes = CountsLayout::bytes;
    for (int attempt = 0; attempt < 3; ++attempt) {
        HIP_TRY(hipEventSynchronize(L->done));
        L->pending = false;
        const size_t n_pass = L->n_pass;
        const int launches = L->launches;
        const bool count = L->count;
        const std::vector<vj_launch>& linfo = L->linfo;
        const uint32_t n_det = ((const uint32_t*)L->h_pinned)[MAX_PASSES * q_counts];
        float ms_i = 0, ms_c = 0, ms_t = 0;
        HIP_TRY(hipEventElapsedTime(&ms_i, L->ev[0], L->ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms_c, L->ev[2], L->ev[3]));
        HIP_TRY(hipEventElapsedTime(&ms_t, L->ev[0], L->ev[3]));
        if (n_det > L->det_cap) {  // detections overflowed: grow and redo the cascade passes
            uint32_t want = L->det_cap;
            while (want < n_det) want *= 2;
            if ((rc = L->d_det.ensure((size_t)want * sizeof(DetEntry)))) return rc;
            L->det_cap = want;
            if (L->shared_integrals) {
                HIP_TRY(hipEventRecord(L->ev[0], e->stream));
                if ((rc = enqueue_integral(e, L->src_gray, L->src_frame_bytes, L->src_stride, W, H, L->nf, L->src_channels))) return rc;
                HIP_TRY(hipEventRecord(L->ev[1], e->stream));
            }
            if ((rc = enqueue_cascade(e, L, pl, W, H, p))) return rc;
            continue;
        }
        if (attempt == 0) tm->integral_ms += ms_i;
        tm->cascade_ms += ms_c;
        tm->total_ms += ms_t;
        if (launches && n_pass <= VJ_MAX_PASSES)
            for (size_t ps = 0; ps < n_pass; ++ps) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, L->pass_ev[ps], L->pass_ev[ps + 1]));
                tm->pass_ms[ps] += ms;
                tm->pass_stage_begin[ps] = (int32_t)pl->pass_bounds[ps];
                tm->pass_stage_end[ps] = (int32_t)pl->pass_bounds[ps + 1];
            }
        tm->n_cascade_launches = std::max(tm->n_cascade_launches, launches);
        const unsigned long long* se_all = (const unsigned long long*)((const uint32_t*)L->h_pinned + CountsLayout::stage_off_u32);
        if (linfo.size() <= VJ_MAX_LAUNCHES) {
            for (size_t i = 0; i < linfo.size(); ++i) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, L->launch_ev[2 * i], L->launch_ev[2 * i + 1]));
                vj_launch acc = tm->launch[i];   // sums over sub-batches: time and counters
                tm->launch[i] = linfo[i];
                tm->launch[i].ms = acc.ms + ms;
                const unsigned long long* se = se_all + (1 + i) * VJ_MAX_STAGES;
                for (size_t s = 0; s < (size_t)VJ_MAX_STAGES; ++s)
                    tm->launch[i].stage_entered[s] = acc.stage_entered[s] + (count ? se[s] : 0ull);
            }
            tm->n_launches = (int32_t)linfo.size();
        }
#ifdef VJ_STAMPS
        if (getenv("VJ_DEBUG_STAMPS")) {  // diagnostic build (-DVJ_STAMPS=1): phase cycle sums of the tile kernel
            const unsigned long long* se = se_all;
            for (size_t l = 0; l < linfo.size(); ++l)
                fprintf(stderr, "vj launch %zu kind %d class %d: max resident workgroups %llu\n", l, linfo[l].kind, linfo[l].lds_class,
                        se[(1 + l) * VJ_MAX_STAGES + 39]);
            for (size_t l = 0; l < linfo.size(); ++l)   // queue passes: chunk time sum / max / chunks; stump-parallel tail: A, B, pairs, stages
                if (linfo[l].kind == 1)
                    fprintf(stderr, "vj queue launch %zu [%d,%d): chunks %llu sum %llu max %llu | tail A %llu B %llu pairs %llu stages %llu\n", l,
                            linfo[l].stage_begin, linfo[l].stage_end, se[(1 + l) * VJ_MAX_STAGES + 50], se[(1 + l) * VJ_MAX_STAGES + 48],
                            se[(1 + l) * VJ_MAX_STAGES + 49], se[(1 + l) * VJ_MAX_STAGES + 51], se[(1 + l) * VJ_MAX_STAGES + 52],
                            se[(1 + l) * VJ_MAX_STAGES + 53], se[(1 + l) * VJ_MAX_STAGES + 54]);
            fprintf(stderr, "vj stamps:");
            for (int i = 40; i < 60; ++i) {
                unsigned long long v = 0;
                for (int l = 1; l <= VJ_MAX_LAUNCHES; ++l) v += se[l * VJ_MAX_STAGES + i];
                fprintf(stderr, " %llu", v);
            }
            fprintf(stderr, "\n");
        }
#endif
        if (count) {
            for (size_t s = 0; s < pl->stages.size(); ++s)
                for (int l = 1; l <= VJ_MAX_LAUNCHES; ++l) ctr->stage_entered[s] += se_all[(size_t)l * VJ_MAX_STAGES + s];
            // multi-node trees: nodes below the roots that the walks visited, and their rectangles — parked in the two derived
            // fields until fill_counters prices the roots (every entering window evaluates those) and adds these
            ctr->stump_evals += se_all[0];
            ctr->gather_bytes += se_all[1];
        }
        std::vector<DetEntry> raw(n_det);
        const uint32_t have = std::min(n_det, L->det_copied);
        if (have) memcpy(raw.data(), (const char*)L->h_pinned + ((counts_bytes + 255) & ~(size_t)255), (size_t)have * sizeof(DetEntry));
        if (n_det > have)
            HIP_TRY(hipMemcpy(raw.data() + have, (const DetEntry*)L->d_det.p + have, (size_t)(n_det - have) * sizeof(DetEntry),
                              hipMemcpyDeviceToHost));
        const uint32_t stride = (uint32_t)W + 1u;
        const uint64_t fbytes = (uint64_t)pl->frame_elems * 4u;
        for (const DetEntry& d : raw) {
            const uint32_t f = (uint32_t)(d.off / fbytes);
            const uint32_t el = (uint32_t)((d.off - (uint64_t)f * fbytes) / 4u);
            dets->push_back(RawDet{f0 + (int)f, d.scale, el % stride, el / stride});
        }
        if (raw_out) *raw_out = std::move(raw);
        return VJ_OK;
    }
    set_error("detection buffer overflow persisted");
    return VJ_ERR_LIMIT;
}

// One sub-batch: frames [f0, f0+nf).  Appends decoded detections to `dets`.
static int detect_subbatch(vj_env* e, Plan* pl, const vj_image* frames, int f0, int nf, int W, int H,
                           const vj_params& p, std::vector<RawDet>* dets, vj_counters* ctr, vj_timing* tm) {
    int rc;
    Lane* L = &e->lane0;
    if ((rc = enqueue_prepare(e, L, pl, frames + f0, nf, W, H, false, nullptr))) return rc;
    if ((rc = enqueue_cascade(e, L, pl, W, H, p))) return rc;
    return finish_batch(e, L, pl, f0, W, H, p, dets, ctr, tm);
}

// Sort, widen and (optionally) group the decoded detections of a whole call; fill the counters.
static void fill_counters(Plan* pl, int n_frames, const vj_params& p, vj_result* out) {
    if (!(p.flags & VJ_FLAG_COUNTERS)) return;
    vj_counters& k = out->counters;
    k.windows = pl->windows_per_frame * (uint64_t)n_frames;
    // node evaluations as the oracle counts them: every node a window's walk visits.  Stumps: every node of an
    // entered stage.  Multi-node trees: the root of every tree of an entered stage + the nodes below the roots that the
    // counted kernels saw visited (finish_batch parked those in the two derived fields).
    const uint64_t child_nodes = k.stump_evals, child_rects = k.gather_bytes;
    uint64_t rect_evals = pl->trees ? child_rects : 0;
    k.stump_evals = pl->trees ? child_nodes : 0;
    for (size_t s = 0; s < pl->stages.size(); ++s) {
        k.stump_evals += k.stage_entered[s] * (pl->trees ? pl->prog.n_roots[s] : pl->prog.n_nodes[s]);
        rect_evals += k.stage_entered[s] * (pl->trees ? pl->prog.n_root_rects[s] : pl->prog.n_rects[s]);
    }
    k.gather_bytes = 48ull * k.windows + 16ull * rect_evals;
}

static int build_result(Plan* pl, std::vector<RawDet>& dets, int n_frames, const vj_params& p, vj_result* out) {
    // deterministic order: (frame, scale_idx, y, x)
    std::sort(dets.begin(), dets.end(), [](const RawDet& a, const RawDet& b) {
        return std::tie(a.frame, a.slot, a.y, a.x) < std::tie(b.frame, b.slot, b.y, b.x);
    });
    out->count = (uint32_t)dets.size();
    if (!dets.empty()) {
        out->rects = (vj_rect*)malloc(dets.size() * sizeof(vj_rect));
        if (!out->rects) return VJ_ERR_NOMEM;
        for (size_t i = 0; i < dets.size(); ++i) {
            const vj_scale_info& si = pl->scales_info[dets[i].slot];
            out->rects[i] = vj_rect{(int32_t)dets[i].x, (int32_t)dets[i].y, si.win_w, si.win_h, 0.0f, dets[i].frame,
                                    si.scale_idx};
        }
    }
    if (p.min_neighbors != 0 && out->count) {  // clod.cpp:1325-1326: filterResult(matches, n, MAX(min_neighbors, 1), EPS)
        const int rc = vj_group_rectangles(out->rects, &out->count, (int)std::max<uint32_t>(p.min_neighbors, 1u), 0.2);
        if (rc) return rc;
    }
    fill_counters(pl, n_frames, p, out);
    return VJ_OK;
}

// Frames per sub-batch so that 32-bit byte offsets into the batch sum image never wrap and the worst-case survivor
// queues stay within a fixed budget.
static uint64_t max_frames_per_subbatch(const vj_env* e, const Plan* pl) {
    const uint64_t frame_bytes = (uint64_t)pl->frame_elems * 4u;
    uint64_t max_frames = (0xffffffffull - (uint64_t)pl->max_reach_elems * 4u - 16u) / frame_bytes;
    const uint64_t q_budget = 6ull << 30;  // bytes per queue
    // a queue holds Q_PARTS parts of ceil(frames / Q_PARTS) frames' worth of windows each (layout_queues)
    auto fit_parts = [](uint64_t frames_worth) {
        return frames_worth >= Q_PARTS ? frames_worth / Q_PARTS * Q_PARTS : std::max<uint64_t>(1, frames_worth);
    };
    if (pl->windows_per_frame) {
        max_frames = std::min<uint64_t>(max_frames, fit_parts(q_budget / (pl->windows_per_frame * sizeof(QEntry))));
        max_frames = std::min<uint64_t>(max_frames, fit_parts(0xffffffffull / pl->windows_per_frame));
    }
    if (e->max_subbatch > 0) max_frames = std::min<uint64_t>(max_frames, (uint64_t)e->max_subbatch);
    return max_frames;
}

static int check_frames(const vj_image* frames, int n_frames, int* W_, int* H_, int* CH_) {
    const int W = frames[0].width, H = frames[0].height;
    if (W <= 0 || H <= 0) return VJ_ERR_ARG;
    const int CH = image_channels(frames[0]);
    if (CH != 1 && CH != 3 && CH != 4) {
        set_error("frames must have 1 (gray), 3 (BGR) or 4 (BGRA) channels");
        return VJ_ERR_ARG;
    }
    for (int i = 0; i < n_frames; ++i) {
        if (!frames[i].data || frames[i].width != W || frames[i].height != H || image_channels(frames[i]) != CH ||
            frames[i].stride < W * CH) {
            set_error("frame %d: all frames of a batch must be non-null and of equal size and channel count", i);
            return VJ_ERR_ARG;
        }
    }
    if ((uint64_t)(W + 1) * (uint64_t)(H + 3) >= (1ull << 30)) {
        set_error("image too large");
        return VJ_ERR_LIMIT;
    }
    *W_ = W;
    *H_ = H;
    *CH_ = CH;
    return VJ_OK;
}

}  // namespace vj

// What every entry point that takes a vj_params checks before it plans anything (vj_detect, vj_detect_rois,
// vj_detect_chain for both parameter sets, vj_stream_create): the same VJ_ERR_ARG everywhere, so that no path can reach a
// kernel with a flag combination the plan does not describe (VJ_FLAG_GRID_F64 alone would give the region pass a position
// table its launcher never binds).
static int check_params(const vj_params& p) {
    if (!(p.scale_factor > 1.0f)) {
        set_error("scale_factor must be > 1");
        return VJ_ERR_ARG;
    }
    if ((p.flags & VJ_FLAG_SKIP_LIST) && (p.flags & VJ_FLAG_SKIP_ROW)) {
        set_error("VJ_FLAG_SKIP_LIST and VJ_FLAG_SKIP_ROW are two different loops of the reference: pick one");
        return VJ_ERR_ARG;
    }
    if ((p.flags & VJ_FLAG_GRID_F64) && !(p.flags & (VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW))) {
        set_error("VJ_FLAG_GRID_F64 names the block variant's two loops: combine it with VJ_FLAG_SKIP_ROW or VJ_FLAG_SKIP_LIST");
        return VJ_ERR_ARG;
    }
    return VJ_OK;
}

extern "C" {
static void drop_plans(vj_env* e);   // (defined inside the extern "C" block below: the same language linkage here)
}

// Arguments of the region pass (roi_plan_units + cascade_roi_pass): `second` inside regions of the nf frames whose integral
// images sit in e->d_sum / e->d_sqsum.  The region list is e->d_rois (max_rois entries), its length on the device at
// roi_counts[0]; the counters live in the lane's counts block.
static void fill_region_args(vj_env* e, Lane* L, Plan* pl2, const vj_cascade* second, const vj_params& p_second, int W, int H, int nf,
                             uint32_t max_rois, RoiArgs* ra_, CascadeArgs* ca_) {
    uint32_t* roi_counts = (uint32_t*)L->d_counts.p + CountsLayout::roi_off_u32;
    const uint32_t stride = (uint32_t)W + 1u;
    RoiArgs& ra = *ra_;
    memset(&ra, 0, sizeof(ra));
    ra.frame_bytes = pl2->frame_elems * 4u;
    ra.stride = stride;
    ra.rois = (RoiDev*)e->d_rois.p;
    ra.n_rois = roi_counts + 0;
    ra.max_rois = max_rois;
    ra.n_frames = (uint32_t)nf;
    ra.frame_w = W;
    ra.frame_h = H;
    ra.win_w0 = second->win_w;
    ra.win_h0 = second->win_h;
    ra.units = (RoiUnit*)e->d_roi_units.p;
    ra.n_units = roi_counts + 1;   // (+2: invalid regions)
    ra.max_units = e->roi_unit_cap;
    ra.ticket = roi_counts + 3;
    ra.det = (RoiDet*)e->d_roi_det.p;
    ra.det_count = roi_counts + 4;
    ra.det_cap = e->roi_det_cap;
    CascadeArgs& ca = *ca_;
    memset(&ca, 0, sizeof(ca));
    ca.sum = (const uint32_t*)e->d_sum.p;
    ca.sqsum = (const uint64_t*)e->d_sqsum.p;
    ca.table = (const uint32_t*)pl2->d_table.p;
    ca.scales = (const ScaleDev*)pl2->d_scales.p;
    ca.stages = (const StageDev*)pl2->d_stages.p;
    ca.n_frames = (uint32_t)nf;
    ca.n_scales = (uint32_t)pl2->scales.size();
    ca.frame_elems = pl2->frame_elems;
    ca.sum_bytes = (uint32_t)((uint64_t)pl2->frame_elems * 4u * (uint64_t)nf);
    ca.stride = stride;
    ca.gather_waves = (uint32_t)e->gather_waves_for(nf, pl2->general);
    ca.stage_begin = 0;
    ca.stage_end = pl2->general ? pl2->pass_bounds.back() : (uint32_t)pl2->stages.size();   // stage trees: positions in the sweep order
    ca.identity_order = pl2->general ? 0u : 1u;
    ca.tree2 = 0u;
    ca.signed_mean = (p_second.flags & VJ_FLAG_SIGNED_MEAN) ? 1u : 0u;
    ca.pos_mode = pl2->pos_mode;   // (0 on this path: check_params sends the f64 grids through the per-size plans; bound all the same)
    ca.pos_tab = (const uint32_t*)pl2->d_pos_tab.p;
    ca.gather_pairs = 2u;   // regions are small: thin waves, latency-bound
    ca.sp_tail_max = (uint32_t)std::max(0, std::min(e->sp_tail_max, 48));
    ca.max_stage_nodes = pl2->max_stage_nodes;
    ca.stage_entered = (unsigned long long*)(roi_counts + 8);
    ca.tree_ctr = (unsigned long long*)((uint32_t*)L->d_counts.p + CountsLayout::stage_off_u32) + 2;   // the spare array's [2], [3]: the region pass's
    // Regions with large grids at the small scales (many raw candidates, big faces) run those grids on the tile kernel
    // (cascade_tile_roi_pass: tiles of the second cascade's two-per-CU tile scales, laid inside the region): stump cascades
    // whose plan has such tiles.  The caller sizes e->d_roi_tiles and zeroes the eight ticket counters.
    if (e->roi_tile_min_windows > 0 && !pl2->general && !pl2->trees && pl2->sp_pad != 0u && pl2->class_first[1] > pl2->class_first[0] &&
        e->roi_tile_cap != 0u) {
        ra.tiles = (RoiTile*)e->d_roi_tiles.p;
        ra.n_tiles = roi_counts + 6;
        ra.max_tiles = e->roi_tile_cap;
        ra.tile_min_windows = (uint32_t)e->roi_tile_min_windows;
        ca.tile_lds_bytes = pl2->class_lds[0];
        const int per_cu = std::max(1, std::min(32 / TILE_WAVES, (int)(160u * 1024u / ca.tile_lds_bytes)));
        ra.tile_blocks = (uint32_t)std::max(1, e->n_cu * per_cu);
        ca.tile_ticket = (uint32_t*)L->d_counts.p + (CountsLayout::q_counts - 40u);
        ca.n_pass = 1;                       // the whole cascade inside the tile: survivors are detections
        ca.pass_begin[0] = 0;
        ca.pass_begin[1] = (uint32_t)pl2->stages.size();
        ca.tile_end = (uint32_t)pl2->stages.size();
        ca.tile_min_lanes = 0;
        ca.tile_repack_mask = e->tile_repack_mask;
        ca.tile_sp_begin = (uint32_t)e->tile_sp_begin;
        ca.tile_sp_pad = pl2->sp_pad;
        ca.sp_blocks = (const SpBlock*)pl2->d_sp_blocks.p;
        ca.n_sp_blocks = pl2->n_sp_blocks;
        ca.tile_sp_max = (uint32_t)std::min(e->tile_sp_max, (int)TILE_SP_MAX_WINDOWS);
        ca.xcd_affinity = (uint32_t)e->xcd_affinity;
        ca.tile_finish = (uint32_t)e->tile_finish;
        ca.tile_ws_min = (uint32_t)e->tile_ws_min;
        ca.tile_ws_max = (uint32_t)std::min(e->tile_ws_max, (int)TILE_WS_MAX_WINDOWS);
    }
}

// vj_detect_rois on frames of one size with a linear cascade: one integral per frame and ONE region pass for regions of
// any sizes (the machinery of vj_detect_chain's second half, the region list uploaded instead of built on the device) —
// not one plan and one vj_detect call per distinct region size.
static int detect_rois_on_device(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames, const vj_roi* rois, int n_rois,
                                 const vj_params* p, int W, int H, vj_result* out) {
    int rc;
    Plan* pl2;
    if ((rc = get_plan(e, c, W, H, *p, &pl2))) return rc;
    const uint64_t max_frames = max_frames_per_subbatch(e, pl2);
    if (max_frames == 0) {
        set_error("a single frame exceeds the 32-bit offset range");
        return VJ_ERR_LIMIT;
    }
    {
        uint64_t dummy = 0;
        if (pl2->frames_q == 0 && (rc = layout_queues(pl2, 1, &dummy))) return rc;
    }
    Lane* L = &e->lane0;
    const bool count2 = (p->flags & VJ_FLAG_COUNTERS) != 0;
    const uint32_t stride = (uint32_t)W + 1u;
    const uint64_t fbytes = (uint64_t)pl2->frame_elems * 4u;
    struct Det2 { int roi; uint32_t slot, x, y; };
    std::vector<Det2> dets2;
    uint64_t child2_nodes = 0, child2_rects = 0;   // counted calls, multi-node trees: visited nodes below the roots (second cascade)
    for (int f0 = 0; f0 < n_frames; f0 += (int)max_frames) {
        const int nf = (int)std::min<uint64_t>(max_frames, (uint64_t)(n_frames - f0));
        std::vector<RoiDev> regions;
        std::vector<int> index;   // position in `regions` -> the caller's region
        for (int i = 0; i < n_rois; ++i)
            if (rois[i].frame >= f0 && rois[i].frame < f0 + nf) {
                regions.push_back(RoiDev{rois[i].frame - f0, rois[i].x, rois[i].y, rois[i].w, rois[i].h});
                index.push_back(i);
            }
        if (regions.empty()) continue;
        if ((rc = enqueue_prepare(e, L, pl2, frames + f0, nf, W, H, false, nullptr))) return rc;
        if (e->roi_unit_cap == 0) e->roi_unit_cap = 1u << 18;
        if (e->roi_det_cap == 0) e->roi_det_cap = 1u << 16;
        const uint32_t n_reg = (uint32_t)regions.size();
        if ((rc = e->d_rois.ensure((size_t)n_reg * sizeof(RoiDev)))) return rc;
        HIP_TRY(hipMemcpyAsync(e->d_rois.p, regions.data(), (size_t)n_reg * sizeof(RoiDev), hipMemcpyHostToDevice, e->stream));
        uint32_t n_units = 0, n_det2 = 0;
        float ms_roi = 0;
        bool ok = false;
        for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
            if ((rc = e->d_roi_units.ensure((size_t)e->roi_unit_cap * sizeof(RoiUnit)))) return rc;
            if ((rc = e->d_roi_det.ensure((size_t)e->roi_det_cap * sizeof(RoiDet)))) return rc;
            if (e->roi_tile_cap == 0) e->roi_tile_cap = 1u << 16;
            if ((rc = e->d_roi_tiles.ensure((size_t)e->roi_tile_cap * sizeof(RoiTile)))) return rc;
            HIP_TRY(hipMemsetAsync((uint32_t*)L->d_counts.p + (CountsLayout::q_counts - 40u), 0, 8 * 4, e->stream));   // tile tickets of the region pass
            uint32_t* roi_counts = (uint32_t*)L->d_counts.p + CountsLayout::roi_off_u32;
            HIP_TRY(hipMemsetAsync(roi_counts, 0, (8 + (size_t)VJ_MAX_STAGES * 2) * 4, e->stream));
            HIP_TRY(hipMemsetAsync((uint32_t*)L->d_counts.p + CountsLayout::stage_off_u32 + 4, 0, 16, e->stream));   // the region pass's tree counters
            HIP_TRY(hipMemcpyAsync(roi_counts, &n_reg, 4, hipMemcpyHostToDevice, e->stream));
            RoiArgs ra;
            CascadeArgs ca;
            fill_region_args(e, L, pl2, c, *p, W, H, nf, n_reg, &ra, &ca);
            HIP_TRY(hipEventRecord(L->launch_ev[2 * VJ_MAX_LAUNCHES - 2], e->stream));
            const int hrc = launch_roi_chain(ra, ca, false, pl2->trees, count2, pl2->general, std::max(1, e->n_cu * e->blocks_per_cu), e->stream,
                                             e->concurrent ? e->stream2 : nullptr, e->fork_ev, e->join_ev);
            if (hrc) {
                set_error("region pass launch failed: %s", hipGetErrorString((hipError_t)hrc));
                return VJ_ERR_HIP;
            }
            HIP_TRY(hipEventRecord(L->launch_ev[2 * VJ_MAX_LAUNCHES - 1], e->stream));
            HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::roi_off_u32, roi_counts, (8 + (size_t)VJ_MAX_STAGES * 2) * 4,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::stage_off_u32 + 4, (uint32_t*)L->d_counts.p + CountsLayout::stage_off_u32 + 4, 16,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipEventRecord(L->done, e->stream));
            HIP_TRY(hipEventSynchronize(L->done));
            const uint32_t* hc = (const uint32_t*)L->h_pinned;
            n_units = hc[CountsLayout::roi_off_u32 + 1];
            n_det2 = hc[CountsLayout::roi_off_u32 + 4];
            if (hc[CountsLayout::roi_off_u32 + 2] != 0) {
                set_error("internal error: %u regions outside their frames", hc[CountsLayout::roi_off_u32 + 2]);
                return VJ_ERR_HIP;
            }
            const uint32_t n_tiles2 = hc[CountsLayout::roi_off_u32 + 6];
            if (n_units > e->roi_unit_cap || n_det2 > e->roi_det_cap || n_tiles2 > e->roi_tile_cap) {
                while (e->roi_unit_cap < n_units) e->roi_unit_cap *= 2;
                while (e->roi_det_cap < n_det2) e->roi_det_cap *= 2;
                while (e->roi_tile_cap < n_tiles2) e->roi_tile_cap *= 2;
                continue;
            }
            HIP_TRY(hipEventElapsedTime(&ms_roi, L->launch_ev[2 * VJ_MAX_LAUNCHES - 2], L->launch_ev[2 * VJ_MAX_LAUNCHES - 1]));
            ok = true;
        }
        if (!ok) {
            set_error("region buffers kept overflowing");
            return VJ_ERR_LIMIT;
        }
        float ms_i = 0;
        HIP_TRY(hipEventElapsedTime(&ms_i, L->ev[0], L->ev[1]));
        out->timing.integral_ms += ms_i;
        out->timing.cascade_ms += ms_roi;
        out->timing.total_ms += ms_i + ms_roi;
        out->timing.n_cascade_launches = 1;
        std::vector<RoiDet> raw2(n_det2);
        if (n_det2) HIP_TRY(hipMemcpy(raw2.data(), e->d_roi_det.p, (size_t)n_det2 * sizeof(RoiDet), hipMemcpyDeviceToHost));
        for (const RoiDet& d : raw2) {
            const uint32_t f = (uint32_t)(d.off / fbytes);
            const uint32_t el = (uint32_t)((d.off - (uint64_t)f * fbytes) / 4u);
            const RoiDev& r1 = regions[d.roi];
            dets2.push_back(Det2{index[d.roi], d.slot, el % stride - (uint32_t)r1.x, el / stride - (uint32_t)r1.y});
        }
        if (count2) {
            const unsigned long long* se = (const unsigned long long*)((const uint32_t*)L->h_pinned + CountsLayout::roi_off_u32 + 8);
            const unsigned long long* tc = (const unsigned long long*)((const uint32_t*)L->h_pinned + CountsLayout::stage_off_u32) + 2;
            child2_nodes += tc[0];
            child2_rects += tc[1];
            for (size_t s2 = 0; s2 < pl2->stages.size(); ++s2) out->counters.stage_entered[s2] += se[s2];
        }
    }
    std::sort(dets2.begin(), dets2.end(), [](const Det2& a, const Det2& b) {
        return std::make_tuple(a.roi, a.slot, a.y, a.x) < std::make_tuple(b.roi, b.slot, b.y, b.x);
    });
    out->count = (uint32_t)dets2.size();
    if (!dets2.empty()) {
        out->rects = (vj_rect*)malloc(dets2.size() * sizeof(vj_rect));
        if (!out->rects) return VJ_ERR_NOMEM;
        for (size_t i = 0; i < dets2.size(); ++i) {
            const vj_scale_info& si = pl2->scales_info[dets2[i].slot];
            out->rects[i] = vj_rect{(int32_t)dets2[i].x, (int32_t)dets2[i].y, si.win_w, si.win_h, 0.0f, dets2[i].roi, si.scale_idx};
        }
    }
    if (p->min_neighbors != 0 && out->count) {
        rc = vj_group_rectangles(out->rects, &out->count, (int)std::max<uint32_t>(p->min_neighbors, 1u), 0.2);
        if (rc) return rc;
    }
    if (count2) {
        vj_counters& k = out->counters;
        k.windows = k.stage_entered[0];
        uint64_t rect_evals = 0;
        for (size_t s2 = 0; s2 < pl2->stages.size(); ++s2) {   // (as fill_counters: roots on the host, visited child nodes from the device)
            k.stump_evals += k.stage_entered[s2] * (pl2->trees ? pl2->prog.n_roots[s2] : pl2->prog.n_nodes[s2]);
            rect_evals += k.stage_entered[s2] * (pl2->trees ? pl2->prog.n_root_rects[s2] : pl2->prog.n_rects[s2]);
        }
        if (pl2->trees) {
            k.stump_evals += child2_nodes;
            rect_evals += child2_rects;
        }
        k.gather_bytes = 48ull * k.windows + 16ull * rect_evals;
    }
    return VJ_OK;
}


extern "C" {

int vj_env_create(int device_index, vj_env** out) {
    if (!out) return VJ_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess || n <= 0) {
        set_error("no HIP device (hipGetDeviceCount: %s); this library has no CPU fallback",
                  he == hipSuccess ? "0 devices" : hipGetErrorString(he));
        return VJ_ERR_NO_DEVICE;
    }
    if (device_index < 0 || device_index >= n) {
        set_error("device_index %d out of range [0,%d)", device_index, n);
        return VJ_ERR_ARG;
    }
    auto e = std::make_unique<vj_env>();
    e->device = device_index;
    HIP_TRY(hipSetDevice(device_index));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_index));
    snprintf(e->name, sizeof(e->name), "%s (%s)", prop.name, prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; the kernels are built for gfx950 only", device_index, prop.gcnArchName);
        return VJ_ERR_NO_DEVICE;
    }
    e->n_cu = prop.multiProcessorCount;
    if (const int hrc = prepare_tile_kernels()) {
        set_error("raising the dynamic LDS limit on device %d failed: %s", device_index, hipGetErrorString((hipError_t)hrc));
        return VJ_ERR_HIP;
    }
    if (const int hrc = prepare_cv_tile_kernels()) {
        set_error("raising the dynamic LDS limit on device %d failed: %s", device_index, hipGetErrorString((hipError_t)hrc));
        return VJ_ERR_HIP;
    }
    if (const int hrc = prepare_group_kernels()) {
        set_error("raising the dynamic LDS limit on device %d failed: %s", device_index, hipGetErrorString((hipError_t)hrc));
        return VJ_ERR_HIP;
    }
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    { const int lrc = e->lane0.create(); if (lrc) return lrc; }
    HIP_TRY(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->join_ev, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    if (const char* s = getenv("VJ_BLOCKS_PER_CU")) e->blocks_per_cu = std::max(1, atoi(s));
    if (const char* s = getenv("VJ_PASS_SPLIT")) {  // e.g. "4,9,15"
        for (const char* q = s; *q;) {
            char* endp;
            long v = strtol(q, &endp, 10);
            if (endp == q) break;
            e->split_override.push_back((int)v);
            q = *endp ? endp + 1 : endp;
        }
    }
    *out = e.release();
    return VJ_OK;
}

void vj_env_destroy(vj_env* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    drop_plans(e);
    for (DevBuf* b : {&e->d_sum, &e->d_sqsum, &e->d_band_sum, &e->d_band_sq, &e->d_band_sqp, &e->d_tilted, &e->d_tilt_diag, &e->d_tilt_col, &e->d_out,
                      &e->d_skip_bits, &e->d_rois, &e->d_roi_units, &e->d_roi_det, &e->d_roi_tiles, &e->d_group, &e->d_cv_det, &e->d_cv_counts,
                      &e->d_cv_accept, &e->d_cv_tq, &e->d_cv_fail_rows, &e->d_cv_fail_walk, &e->d_run_table})
        b->release();
    e->lane0.destroy();
    for (DevBuf& b : e->d_q) b.release();
    for (DevBuf& b : e->d_q2) b.release();
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    if (e->join_ev) (void)hipEventDestroy(e->join_ev);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int vj_env_device_name(const vj_env* e, char* buf, size_t cap) {
    if (!e || !buf || cap == 0) return VJ_ERR_ARG;
    snprintf(buf, cap, "%s, %d CUs", e->name, e->n_cu);
    return VJ_OK;
}

static void drop_plans(vj_env* e) {
    for (auto& kv : e->plans) kv.second->release_device();
    e->plans.clear();
    for (auto& kv : e->cv_plans) kv.second->release_device();
    e->cv_plans.clear();
}

int vj_env_configure(vj_env* e, const char* key, const char* value) {
    if (!e || !key || !value) return VJ_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    if (strcmp(key, "pass_cut_nodes") == 0) {   // default pass cuts: after these numbers of cumulative nodes
        std::vector<int> v;
        for (const char* q = value; *q;) {
            char* end = nullptr;
            const long x = strtol(q, &end, 10);
            if (end == q) {
                set_error("pass_cut_nodes: expected comma-separated integers, got '%s'", value);
                return VJ_ERR_ARG;
            }
            v.push_back((int)x);
            q = *end == ',' ? end + 1 : end;
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->pass_cut_nodes = v;
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "pass_split") == 0) {
        std::vector<int> v;
        for (const char* q = value; *q;) {
            char* endp;
            long x = strtol(q, &endp, 10);
            if (endp == q) {
                set_error("pass_split: expected comma-separated integers, got '%s'", value);
                return VJ_ERR_ARG;
            }
            v.push_back((int)x);
            q = *endp ? endp + 1 : endp;
        }
        e->split_override = v;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);  // pass bounds are part of a plan
        return VJ_OK;
    }
    if (strcmp(key, "tile_classes_kb") == 0) {  // "36,64,140"; "0,0,0" disables the LDS-tile path
        int v[TILE_CLASSES] = {0, 0, 0};
        const char* q = value;
        for (int i = 0; i < TILE_CLASSES && *q; ++i) {
            char* endp;
            v[i] = (int)strtol(q, &endp, 10);
            if (endp == q || v[i] < -4 || v[i] > 140) {
                set_error("tile_classes_kb: expected up to %d values in [0,140]", TILE_CLASSES);
                return VJ_ERR_ARG;
            }
            q = *endp ? endp + 1 : endp;
        }
        for (int i = 0; i < TILE_CLASSES; ++i) e->tile_class_kb[i] = v[i];
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "tile_finish") == 0) {
        e->tile_finish = atoi(value) != 0 ? 1 : 0;
        return VJ_OK;
    }
    if (strcmp(key, "tile_ws_min") == 0) {
        e->tile_ws_min = std::max(0, std::min(atoi(value), (int)TILE_SP_MAX_WINDOWS));
        return VJ_OK;
    }
    if (strcmp(key, "tile_ws_max") == 0) {
        e->tile_ws_max = std::max(0, std::min(atoi(value), (int)TILE_WS_MAX_WINDOWS));
        return VJ_OK;
    }
    if (strcmp(key, "tile_sp_max") == 0) {
        e->tile_sp_max = std::max(0, std::min(atoi(value), (int)TILE_SP_MAX_WINDOWS));
        return VJ_OK;
    }
    if (strcmp(key, "tile_lds_nest") == 0) {
        e->tile_lds_nest = atoi(value) != 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "tree_split_queues") == 0) {
        e->tree_split_queues = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "q_band_px") == 0 || strcmp(key, "q_group_units") == 0) {   // band-major first-pass units and queue pass (0: off)
        (strcmp(key, "q_band_px") == 0 ? e->q_band_px : e->q_group_units) = std::max(0, atoi(value));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "q_band_min_frames") == 0) {
        e->q_band_min_frames = std::max(1, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "cv_pairs") == 0) {
        e->cv_pairs = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "cv_tail_max") == 0) {   // (part of the plan: cached plans are dropped)
        e->cv_tail_max = std::max(0, std::min(atoi(value), (int)CV_TAIL_MAX));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "cv_tree_chunk") == 0 || strcmp(key, "cv_tree_chain_blocks") == 0) {
        (strcmp(key, "cv_tree_chunk") == 0 ? e->cv_tree_chunk : e->cv_tree_chain_blocks) = std::max(1, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "one_pass_max_frames") == 0) {   // (part of the plan key: nothing to drop)
        e->one_pass_max_frames = std::max(0, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "tilted_bands") == 0) {
        e->tilted_bands = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "cv_tiles_tilted") == 0) {   // (part of the plan: cached plans are dropped)
        e->cv_tiles_tilted = atoi(value) != 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "cv_tree2") == 0) {   // (the default balance depends on it: cached plans are dropped)
        e->cv_tree2 = atoi(value) != 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "cv_row_band_px") == 0) {   // OpenCV profile: row order of cv_profile_pass (part of the plan: cached plans are dropped)
        e->cv_row_band_px = std::max(0, atoi(value));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "cv_tree_chains") == 0) {   // OpenCV profile, stage trees made of chains: compacting chain sweeps (1) or the per-lane walk (0)
        e->cv_tree_chains = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "cv_tree_queue_cap") == 0) {
        e->cv_tree_queue_cap = std::max(0, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "roi_tiles") == 0) {   // region pass: smallest (region, scale) grid that runs on LDS tiles (0: none)
        e->roi_tile_min_windows = std::max(0, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "rois_on_device") == 0) {
        e->rois_on_device = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "cv_tiles") == 0 || strcmp(key, "cv_tile_ws_max") == 0 || strcmp(key, "cv_tile_min_windows") == 0 ||
        strcmp(key, "cv_row_blocks") == 0 || strcmp(key, "cv_tile_min_windows0") == 0 || strcmp(key, "cv_row_blocks_tree") == 0 ||
        strcmp(key, "cv_tile_min_windows_tree") == 0) {
        // OpenCV profile: 0 = every scale on cv_profile_pass; the finish threshold; the smallest tile worth staging
        const int v = atoi(value);
        if (strcmp(key, "cv_tiles") == 0) e->cv_tiles = v != 0;
        else if (strcmp(key, "cv_tile_ws_max") == 0) e->cv_tile_ws_max = std::max(0, std::min(v, (int)CVT_WS_MAX));
        else if (strcmp(key, "cv_row_blocks") == 0) e->cv_row_blocks = v < 0 ? -1 : std::max(1, std::min(v, 4));
        else if (strcmp(key, "cv_tile_min_windows0") == 0) e->cv_tile_min_windows0 = std::max(64, v);
        else if (strcmp(key, "cv_row_blocks_tree") == 0) e->cv_row_blocks_tree = std::max(1, std::min(v, 4));
        else if (strcmp(key, "cv_tile_min_windows_tree") == 0) e->cv_tile_min_windows_tree = std::max(64, v);
        else e->cv_tile_min_windows = v < 0 ? -1 : std::max(64, v);
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (auto& kv : e->cv_plans) kv.second->release_device();
        e->cv_plans.clear();
        return VJ_OK;
    }
    if (strcmp(key, "wide_tail") == 0) {
        e->wide_tail = std::max(-1, std::min(atoi(value), 1));   // -1: by batch size
        return VJ_OK;
    }
    if (strcmp(key, "min_chunk") == 0) {
        e->min_chunk = std::max(1, std::min(atoi(value), 64));
        return VJ_OK;
    }
    if (strcmp(key, "q_slices") == 0) {
        e->q_slices = std::max(-1, std::min(atoi(value), 64));
        return VJ_OK;
    }
    if (strcmp(key, "thin_pass_spread") == 0) {
        e->thin_pass_spread = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "group_max") == 0) {
        e->group_max = std::max(1, std::min(atoi(value), (int)GROUP_MAX));
        return VJ_OK;
    }
    if (strcmp(key, "sp_tail_max") == 0) {
        e->sp_tail_max = std::max(0, std::min(atoi(value), 48));
        return VJ_OK;
    }
    if (strcmp(key, "gather_pairs") == 0) {
        e->gather_pairs = std::max(-1, std::min(atoi(value), 2));   // -1: by batch size
        return VJ_OK;
    }
    if (strcmp(key, "tile_class_order") == 0) {
        e->tile_class_order = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "tile_stage_x4") == 0) {
        e->tile_stage_x4 = atoi(value) != 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "tile_deinterleave") == 0) {
        e->tile_deinterleave = atoi(value) != 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "tile_sp_begin") == 0) {
        e->tile_sp_begin = std::max(0, atoi(value));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);   // the LDS layout of the tile launches depends on it
        return VJ_OK;
    }
    if (strcmp(key, "tile_min_windows") == 0 || strcmp(key, "tile_end") == 0 || strcmp(key, "tile_min_lanes") == 0 ||
        strcmp(key, "tile_accept_windows") == 0 || strcmp(key, "tile_max_dwords_per_window") == 0) {
        const int v = atoi(value);
        if (v < 0 || v > 65536) {
            set_error("%s out of range", key);
            return VJ_ERR_ARG;
        }
        (strcmp(key, "tile_max_dwords_per_window") == 0 ? e->tile_max_dwords_per_window
         : strcmp(key, "tile_min_windows") == 0      ? e->tile_min_windows
         : strcmp(key, "tile_end") == 0           ? e->tile_end
         : strcmp(key, "tile_accept_windows") == 0 ? e->tile_accept_windows
                                                  : e->tile_min_lanes) = v;
        if (strcmp(key, "tile_end") != 0 && strcmp(key, "tile_min_lanes") != 0) e->tile_thresholds_set = true;   // the caller's choice holds for every frame size
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "plan_cache_max") == 0) {
        e->plan_cache_max = std::max(2, atoi(value));   // vj_detect_chain holds two plans at once
        return VJ_OK;
    }
    if (strcmp(key, "max_subbatch") == 0) {
        e->max_subbatch = std::max(0, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "det_cap") == 0) {  // (re)sets the detection buffer capacity; it still grows on overflow
        const int v = atoi(value);
        if (v < 1) {
            set_error("det_cap must be >= 1");
            return VJ_ERR_ARG;
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->det_cap_init = (uint32_t)v;
        e->lane0.det_cap = 0;
        return VJ_OK;
    }
    if (strcmp(key, "gather_waves") == 0) {   // waves per workgroup of the global-gather kernels: 3, 4, or -1 = 4 for calls of >= 8 frames
        e->gather_waves = atoi(value);
        return VJ_OK;
    }
    if (strcmp(key, "concurrent_blocks_per_cu") == 0) {
        e->concurrent_blocks_per_cu = std::max(1, atoi(value));
        return VJ_OK;
    }
    if (strcmp(key, "tile_split") == 0) {   // one value for every batch size, or "small,mid,large" (<= 4, < 32, >= 32 frames)
        float a = 0, b = 0, c3 = 0;
        const int n = sscanf(value, "%f,%f,%f", &a, &b, &c3);
        if (n == 3) {
            e->tile_split_small = std::max(0.0f, a);
            e->tile_split_mid = std::max(0.0f, b);
            e->tile_split = std::max(0.0f, c3);
        } else {
            e->tile_split_small = e->tile_split_mid = e->tile_split = std::max(0.0f, (float)atof(value));
        }
        e->tile_split_set = true;      // the caller's values hold: no feedback
        e->balance.clear();
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "balance_exact") == 0) {
        e->balance_exact = atoi(value) != 0;
        e->balance.clear();
        return VJ_OK;
    }
    if (strcmp(key, "balance_export") == 0) return balance_export(e, value);   // value: a file path
    if (strcmp(key, "balance_import") == 0) return balance_import(e, value);
    if (strcmp(key, "auto_balance") == 0) {   // 1: find the chain balance of a batch workload from its first calls' times; "reset": start over
        if (strcmp(value, "reset") == 0) e->tile_split_set = false;
        else e->auto_balance = atoi(value) != 0;
        e->balance.clear();
        return VJ_OK;
    }
    if (strcmp(key, "xcd_affinity") == 0) {
        e->xcd_affinity = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "seg_cut2") == 0) {   // stage trees: second cut inside a long chain, after this many of its stages (0: none)
        e->seg_cut2 = std::max(0, atoi(value));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "tile_segments") == 0) {
        e->tile_segments = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "general_prefix") == 0) {
        e->general_prefix = atoi(value) != 0;
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "grid_block_w") == 0) {
        e->grid_block_w = std::max(0, std::min(atoi(value), (int)UNIT_WINDOWS));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "global_blocks") == 0) {
        e->global_blocks = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "tile_lds_reserve_kb") == 0) {
        e->tile_lds_reserve_kb = std::max(0, std::min(atoi(value), 96));
        HIP_TRY(hipStreamSynchronize(e->stream));
        drop_plans(e);
        return VJ_OK;
    }
    if (strcmp(key, "concurrent") == 0) {  // 1: tile launches and the global first pass overlap on two streams
        e->concurrent = atoi(value) != 0;
        return VJ_OK;
    }
    if (strcmp(key, "tile_repack") == 0) {  // "3,5": stages before which tiles re-pack ("" = never)
        unsigned long long m = 0;
        for (const char* q = value; *q;) {
            char* endp;
            long x = strtol(q, &endp, 10);
            if (endp == q || x < 1 || x > 63) {
                set_error("tile_repack: expected comma-separated stage indices in [1,63]");
                return VJ_ERR_ARG;
            }
            m |= 1ull << x;
            q = *endp ? endp + 1 : endp;
        }
        e->tile_repack_mask = m;
        return VJ_OK;
    }
    if (strcmp(key, "integral_rows") == 0) {   // integral: see IntegralArgs::rows_mode
        e->integral_rows_mode = std::max(0, std::min(atoi(value), 2));
        return VJ_OK;
    }
    if (strcmp(key, "blocks_per_cu") == 0) {
        const int v = atoi(value);
        if (v < 1 || v > 16) {
            set_error("blocks_per_cu must be in [1,16]");
            return VJ_ERR_ARG;
        }
        e->blocks_per_cu = v;
        return VJ_OK;
    }
    set_error("unknown option '%s'", key);
    return VJ_ERR_ARG;
}

int vj_env_reserve(vj_env* e, int max_w, int max_h, int max_batch) {
    if (!e || max_w <= 0 || max_h <= 0 || max_batch <= 0) return VJ_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    return ensure_image_buffers(e, max_w, max_h, max_batch, true);
}

int vj_integral_image(vj_env* e, const vj_image* image, uint32_t* sum, uint64_t* sqsum) {
    if (!e || !image || !image->data || !sum || !sqsum) return VJ_ERR_ARG;
    const int w = image->width, h = image->height, ch = image_channels(*image);
    if (w <= 0 || h <= 0 || (ch != 1 && ch != 3 && ch != 4) || image->stride < w * ch) return VJ_ERR_ARG;
    if ((uint64_t)(w + 1) * (uint64_t)(h + 3) >= (1ull << 30)) {
        set_error("image too large");
        return VJ_ERR_LIMIT;
    }
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if ((rc = ensure_image_buffers(e, w, h, 1, true, ch))) return rc;
    const uint8_t* d_gray;
    size_t fb;
    int gs;
    if ((rc = stage_frames(e, image, 1, w, h, &d_gray, &fb, &gs))) return rc;
    if ((rc = enqueue_integral(e, d_gray, fb, gs, w, h, 1, ch))) return rc;
    const size_t n = (size_t)(w + 1) * (size_t)(h + 1);
    HIP_TRY(hipMemcpyAsync(sum, e->d_sum.p, n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(sqsum, e->d_sqsum.p, n * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return VJ_OK;
}

static int check_single_image(const vj_image* image, int* ch_out) {
    const int w = image->width, h = image->height, ch = image_channels(*image);
    if (w <= 0 || h <= 0 || (ch != 1 && ch != 3 && ch != 4) || image->stride < w * ch) return VJ_ERR_ARG;
    if ((uint64_t)(w + 1) * (uint64_t)(h + 3) >= (1ull << 30)) {
        set_error("image too large");
        return VJ_ERR_LIMIT;
    }
    *ch_out = ch;
    return VJ_OK;
}

int vj_integral_tilted(vj_env* e, const vj_image* image, uint32_t* tilted) {
    if (!e || !image || !image->data || !tilted) return VJ_ERR_ARG;
    int ch, rc;
    if ((rc = check_single_image(image, &ch))) return rc;
    const int w = image->width, h = image->height;
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = ensure_image_buffers(e, w, h, 1, true, ch))) return rc;
    const uint8_t* d_gray;
    size_t fb;
    int gs;
    if ((rc = stage_frames(e, image, 1, w, h, &d_gray, &fb, &gs))) return rc;
    if ((rc = enqueue_tilted(e, d_gray, fb, gs, w, h, 1, ch))) return rc;
    HIP_TRY(hipMemcpyAsync(tilted, e->d_tilted.p, (size_t)(w + 1) * (size_t)(h + 1) * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return VJ_OK;
}

int vj_grayscale(vj_env* e, const vj_image* image, uint8_t* gray, int gray_stride) {
    if (!e || !image || !image->data || !gray || gray_stride < image->width) return VJ_ERR_ARG;
    int ch, rc;
    if ((rc = check_single_image(image, &ch))) return rc;
    const int w = image->width, h = image->height;
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = ensure_image_buffers(e, w, h, 1, true, ch))) return rc;
    const uint8_t* d_src;
    size_t fb;
    int gs;
    if ((rc = stage_frames(e, image, 1, w, h, &d_src, &fb, &gs))) return rc;
    const size_t pitch = ((size_t)w + 3) & ~(size_t)3;
    if ((rc = e->d_out.ensure(pitch * (size_t)h))) return rc;
    TiltedArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.gray = d_src;
    ta.gray_frame_bytes = fb;
    ta.gray_stride = (uint32_t)gs;
    ta.channels = (uint32_t)ch;
    ta.width = (uint32_t)w;
    ta.height = (uint32_t)h;
    ta.n_frames = 1;
    const int hrc = launch_grayscale(ta, (uint8_t*)e->d_out.p, (uint32_t)pitch, e->stream);
    if (hrc) {
        set_error("grayscale launch failed: %s", hipGetErrorString((hipError_t)hrc));
        return VJ_ERR_HIP;
    }
    if ((size_t)gray_stride == pitch && pitch == (size_t)w) {
        // one block only when the device rows carry no padding: with w % 4 != 0 a block copy would write the pad bytes
        // over the caller's row padding (caller data if `gray` is a view into a larger image) and past a tight last row
        HIP_TRY(hipMemcpyAsync(gray, e->d_out.p, pitch * (size_t)h, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        return VJ_OK;
    }
    // another row stride on the host: one block into page-locked memory, the rows from there (a 2-D copy into pageable
    // memory goes row by row in the runtime)
    Lane* L = &e->lane0;
    const size_t need = pitch * (size_t)h;
    if (L->h_stage_bytes < need) {
        if (L->h_stage) (void)hipHostFree(L->h_stage);
        L->h_stage = nullptr;
        L->h_stage_bytes = 0;
        HIP_TRY(hipHostMalloc(&L->h_stage, need, hipHostMallocDefault));
        L->h_stage_bytes = need;
    }
    HIP_TRY(hipMemcpyAsync(L->h_stage, e->d_out.p, need, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int y = 0; y < h; ++y) memcpy(gray + (size_t)y * (size_t)gray_stride, (const uint8_t*)L->h_stage + (size_t)y * pitch, (size_t)w);
    return VJ_OK;
}

int vj_host_alloc(vj_env* e, size_t bytes, void** out) {
    if (!e || !out || bytes == 0) return VJ_ERR_ARG;
    *out = nullptr;
    HIP_TRY(hipSetDevice(e->device));
    const hipError_t he = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (he != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(he));
        *out = nullptr;
        return he == hipErrorOutOfMemory ? VJ_ERR_NOMEM : VJ_ERR_HIP;
    }
    return VJ_OK;
}

void vj_host_free(vj_env* e, void* p) {
    if (!e || !p) return;
    (void)hipSetDevice(e->device);
    (void)hipHostFree(p);
}

int vj_integral(vj_env* e, const uint8_t* gray, int w, int h, int stride, uint32_t* sum, uint64_t* sqsum) {
    if (!e || !gray || !sum || !sqsum || w <= 0 || h <= 0 || stride < w) return VJ_ERR_ARG;
    vj_image im{gray, w, h, stride, 0, 1};
    return vj_integral_image(e, &im, sum, sqsum);
}

int vj_detect(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames, const vj_params* p,
              vj_result* out) {
    if (!e || !c || !p || !out || n_frames < 0 || (n_frames > 0 && !frames)) return VJ_ERR_ARG;
    memset(out, 0, sizeof(*out));
    if (n_frames == 0) return VJ_OK;
    int rc = check_params(*p);
    if (rc) return rc;
    int W, H, CH;
    rc = check_frames(frames, n_frames, &W, &H, &CH);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    Plan* pl;
    vj_env::Balance* bal = balance_of(e, c, W, H, *p, n_frames, true);   // (created before the plan is looked up: it names the split)
    if (bal) balance_touch(bal, n_frames);
    rc = get_plan(e, c, W, H, *p, &pl, n_frames);
    if (rc) return rc;
    const uint64_t max_frames = max_frames_per_subbatch(e, pl);
    if (max_frames == 0) {
        set_error("a single frame exceeds the 32-bit offset range");
        return VJ_ERR_LIMIT;
    }
    std::vector<RawDet> dets;
    for (int f0 = 0; f0 < n_frames; f0 += (int)max_frames) {
        const int nf = (int)std::min<uint64_t>(max_frames, (uint64_t)(n_frames - f0));
        rc = detect_subbatch(e, pl, frames, f0, nf, W, H, *p, &dets, &out->counters, &out->timing);
        if (rc) return rc;
    }
    // feedback for the chain balance: only the timed (uncounted) kernel variants.  A plan none of whose scales can run on tiles
    // has nothing to balance; a CANDIDATE split that leaves one chain only (every tile moved to the gathers, or a share of
    // the scales without large ones) is measured like any other — the search must be able to step back from it
    if (bal && bal->phase != 3) {
        bool any_tile_scale = false;
        for (const ScaleDev& sd : pl->scales) any_tile_scale |= sd.tile_rw != 0;
        if (!any_tile_scale) {
            bal->cur = bal->best;
            bal->phase = 3;
        } else if (!(p->flags & VJ_FLAG_COUNTERS) && (uint64_t)n_frames <= max_frames && n_frames == bal->n_ref) {
            balance_report(bal, out->timing.cascade_ms / (float)n_frames, !e->tile_thresholds_set);
        }
    }
    out->timing.tile_split = pl->tile_split;
    out->timing.balance_state = !bal ? 0 : bal->phase == 3 ? 2 : 1;
    out->timing.balance_calls = bal ? bal->calls : 0;
    return build_result(pl, dets, n_frames, *p, out);
}

// Second cascade on regions of interest (config 5: eyes inside faces).  A ROI is a view —
// pointer + stride — into its frame, as the reference's callers would pass a sub-image header; detection windows
// come in few distinct sizes, so the ROIs are grouped by size and every group is ONE batched vj_detect pass.
int vj_detect_rois(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames, const vj_roi* rois, int n_rois,
                   const vj_params* p, vj_result* out) {
    if (!e || !c || !p || !out || n_frames < 0 || n_rois < 0 || (n_rois > 0 && (!frames || !rois))) return VJ_ERR_ARG;
    memset(out, 0, sizeof(*out));
    {
        const int prc = check_params(*p);
        if (prc) return prc;
    }
    std::map<std::pair<int, int>, std::vector<int>> by_size;   // (w, h) -> ROI indices
    for (int i = 0; i < n_rois; ++i) {
        const vj_roi& r = rois[i];
        if (r.frame < 0 || r.frame >= n_frames || !frames[r.frame].data || r.w <= 0 || r.h <= 0 || r.x < 0 || r.y < 0 ||
            r.x + r.w > frames[r.frame].width || r.y + r.h > frames[r.frame].height) {
            set_error("roi %d lies outside its frame", i);
            return VJ_ERR_ARG;
        }
        by_size[{r.w, r.h}].push_back(i);
    }
    // frames of one size, a linear cascade, the exhaustive grid: every region in one pass on the frames' own integral images
    // (a scale mask selects among the plan's scales; every region still enumerates its own prefix of them)
    if (n_rois > 0 && n_frames > 0 && !(p->flags & (VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW)) && p->scale_factor > 1.0f && e->rois_on_device) {
        bool same = true;
        for (int i = 0; i < n_frames && same; ++i)
            same = frames[i].data && frames[i].width == frames[0].width && frames[i].height == frames[0].height &&
                   image_channels(frames[i]) == image_channels(frames[0]) && frames[i].stride >= frames[i].width * image_channels(frames[i]);
        int W = 0, H = 0, CH = 0;
        if (same && check_frames(frames, n_frames, &W, &H, &CH) == VJ_OK) {
            HIP_TRY(hipSetDevice(e->device));
            return detect_rois_on_device(e, c, frames, n_frames, rois, n_rois, p, W, H, out);
        }
    }
    std::vector<vj_rect> all;
    for (const auto& kv : by_size) {
        std::vector<vj_image> views;
        int ch0 = -1;
        for (int i : kv.second) {
            const vj_roi& r = rois[i];
            const vj_image& f = frames[r.frame];
            const int ch = image_channels(f);
            if (ch0 < 0) ch0 = ch;
            if (ch != ch0) {
                set_error("ROIs of one size must come from frames with the same channel count");
                return VJ_ERR_ARG;
            }
            views.push_back(vj_image{f.data + (size_t)r.y * (size_t)f.stride + (size_t)r.x * (size_t)ch, r.w, r.h, f.stride,
                                     f.on_device, f.channels});
        }
        vj_result part;
        int rc = vj_detect(e, c, views.data(), (int)views.size(), p, &part);
        if (rc) {
            vj_result_free(&part);
            return rc;
        }
        for (uint32_t k = 0; k < part.count; ++k) {
            vj_rect rr = part.rects[k];
            rr.frame = kv.second[(size_t)rr.frame];   // index in the batch -> ROI index
            all.push_back(rr);
        }
        // counters and times add up over the groups
        out->counters.windows += part.counters.windows;
        out->counters.stump_evals += part.counters.stump_evals;
        out->counters.gather_bytes += part.counters.gather_bytes;
        for (int s2 = 0; s2 < VJ_MAX_STAGES; ++s2) out->counters.stage_entered[s2] += part.counters.stage_entered[s2];
        out->timing.integral_ms += part.timing.integral_ms;
        out->timing.cascade_ms += part.timing.cascade_ms;
        out->timing.total_ms += part.timing.total_ms;
        vj_result_free(&part);
    }
    std::stable_sort(all.begin(), all.end(), [](const vj_rect& a, const vj_rect& b) { return a.frame < b.frame; });
    out->count = (uint32_t)all.size();
    if (!all.empty()) {
        out->rects = (vj_rect*)malloc(all.size() * sizeof(vj_rect));
        if (!out->rects) return VJ_ERR_NOMEM;
        memcpy(out->rects, all.data(), all.size() * sizeof(vj_rect));
    }
    return VJ_OK;
}


// ------------------------------------------------------------------ two cascades, hand-off on the device
int vj_detect_chain(vj_env* e, const vj_cascade* first, const vj_cascade* second, const vj_image* frames, int n_frames,
                    const vj_params* p_first, const vj_params* p_second, vj_result* out_first, vj_result* out_second) {
    if (!e || !first || !second || !p_first || !p_second || !out_first || !out_second || n_frames < 0 || (n_frames > 0 && !frames))
        return VJ_ERR_ARG;
    memset(out_first, 0, sizeof(*out_first));
    memset(out_second, 0, sizeof(*out_second));
    if (n_frames == 0) return VJ_OK;
    for (const vj_params* pp : {p_first, p_second}) {
        const int prc = check_params(*pp);
        if (prc) return prc;
    }
    if ((p_first->flags | p_second->flags) & (VJ_FLAG_SKIP_LIST | VJ_FLAG_SKIP_ROW)) {
        // the CPU variants' skip rules make a window's fate depend on its row's history inside ITS image: the first cascade runs
        // as vj_detect does, the second one per region size on the sub-images (vj_detect_rois' path for these modes) — the
        // hand-off goes through the host, the results are the ones the definition above gives
        int rc = vj_detect(e, first, frames, n_frames, p_first, out_first);
        if (rc) return rc;
        std::vector<vj_roi> regions(out_first->count);
        for (uint32_t i = 0; i < out_first->count; ++i) {
            const vj_rect& q = out_first->rects[i];
            regions[i] = vj_roi{(int32_t)q.frame, q.x, q.y, q.w, q.h};
        }
        if (regions.empty()) return VJ_OK;
        return vj_detect_rois(e, second, frames, n_frames, regions.data(), (int)regions.size(), p_second, out_second);
    }
    const bool grouped = p_first->min_neighbors != 0;   // the regions are the GROUPED candidates (grouped on the device)
    int W, H, CH;
    int rc = check_frames(frames, n_frames, &W, &H, &CH);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    Plan *pl1, *pl2;
    // (the first cascade's chain balance is found by feedback as in vj_detect: its kernels run before the region pass starts)
    vj_env::Balance* bal = balance_of(e, first, W, H, *p_first, n_frames, true);
    if (bal) balance_touch(bal, n_frames);
    if ((rc = get_plan(e, first, W, H, *p_first, &pl1, n_frames))) return rc;
    // the second cascade is planned for the frame's stride and every scale a region as large as the frame could use;
    // each region picks its own scales and grid on the device
    if ((rc = get_plan(e, second, W, H, *p_second, &pl2))) return rc;
    if ((rc = get_plan(e, first, W, H, *p_first, &pl1, n_frames))) return rc;   // (the cache may have evicted it for pl2: look it up again)
    const uint64_t max_frames = std::min(max_frames_per_subbatch(e, pl1), max_frames_per_subbatch(e, pl2));
    if (max_frames == 0) {
        set_error("a single frame exceeds the 32-bit offset range");
        return VJ_ERR_LIMIT;
    }
    {   // the second plan's scale records go to the device with a queue layout (none is used by the region pass)
        uint64_t dummy = 0;
        if (pl2->frames_q == 0 && (rc = layout_queues(pl2, 1, &dummy))) return rc;
    }
    Lane* L = &e->lane0;
    std::vector<RawDet> dets1;
    std::vector<vj_rect> grouped1;       // grouped mode: the first result, as the device grouped it
    struct Det2 { int roi; uint32_t slot, x, y; };
    std::vector<Det2> dets2;
    uint64_t child2_nodes = 0, child2_rects = 0;   // counted calls, multi-node trees: visited nodes below the roots (second cascade)
    const bool count2 = (p_second->flags & VJ_FLAG_COUNTERS) != 0;
    const uint32_t stride = (uint32_t)W + 1u;
    const uint64_t fbytes = (uint64_t)pl1->frame_elems * 4u;
    for (int f0 = 0; f0 < n_frames; f0 += (int)max_frames) {
        const int nf = (int)std::min<uint64_t>(max_frames, (uint64_t)(n_frames - f0));
        if ((rc = enqueue_prepare(e, L, pl1, frames + f0, nf, W, H, false, nullptr))) return rc;
        if (e->roi_unit_cap == 0) e->roi_unit_cap = 1u << 18;
        if (e->roi_det_cap == 0) e->roi_det_cap = 1u << 16;
        uint32_t n_det1 = 0, n_units = 0, n_det2 = 0;
        size_t roi_weight_off = 0;   // grouped mode: where the groups' member counts sit in d_group
        float ms_roi = 0;
        bool ok = false;
        for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
            if ((rc = enqueue_cascade(e, L, pl1, W, H, *p_first))) return rc;
            if ((rc = e->d_rois.ensure((size_t)L->det_cap * sizeof(RoiDev)))) return rc;
            GroupArgs ga;
            memset(&ga, 0, sizeof(ga));
            if (grouped) {
                // one buffer: keys (u64) | grouped | the zeroed counter block | frame_first | weights
                const size_t cap = L->det_cap, nfz = (size_t)nf;
                const size_t o_keys = 0, o_grouped = o_keys + cap * 8u, o_zero = o_grouped + cap * sizeof(RoiDev),
                             o_first = o_zero + (3u * nfz + 1u) * 4u, o_gw = o_first + (nfz + 1u) * 4u, o_rw = o_gw + cap * 4u,
                             total = o_rw + cap * 4u;
                if ((rc = e->d_group.ensure(total))) return rc;
                char* gb = (char*)e->d_group.p;
                ga.det = (const DetEntry*)L->d_det.p;
                ga.det_count = (const uint32_t*)L->d_counts.p + MAX_PASSES * CountsLayout::q_counts;
                ga.det_cap = L->det_cap;
                ga.scales = (const ScaleDev*)pl1->d_scales.p;
                ga.frame_bytes = pl1->frame_elems * 4u;
                ga.stride = stride;
                ga.n_frames = (uint32_t)nf;
                ga.threshold = (int32_t)std::max<uint32_t>(p_first->min_neighbors, 1u);   // clod.cpp:1326
                ga.eps = 0.2;                                                               // clod.cpp:11
                ga.keys = (uint64_t*)(gb + o_keys);
                ga.grouped = (RoiDev*)(gb + o_grouped);
                ga.frame_count = (uint32_t*)(gb + o_zero);
                ga.frame_cursor = ga.frame_count + nfz;
                ga.grouped_count = ga.frame_cursor + nfz;
                ga.overflow = ga.grouped_count + nfz;
                ga.frame_first = (uint32_t*)(gb + o_first);
                ga.grouped_weight = (uint32_t*)(gb + o_gw);
                ga.roi_weight = (uint32_t*)(gb + o_rw);
                roi_weight_off = o_rw;
                ga.rois = (RoiDev*)e->d_rois.p;
                ga.max_rois = L->det_cap;
                ga.group_max = (uint32_t)e->group_max;
            }
            if ((rc = e->d_roi_units.ensure((size_t)e->roi_unit_cap * sizeof(RoiUnit)))) return rc;
            if ((rc = e->d_roi_det.ensure((size_t)e->roi_det_cap * sizeof(RoiDet)))) return rc;
            if (e->roi_tile_cap == 0) e->roi_tile_cap = 1u << 16;
            if ((rc = e->d_roi_tiles.ensure((size_t)e->roi_tile_cap * sizeof(RoiTile)))) return rc;
            uint32_t* roi_counts = (uint32_t*)L->d_counts.p + CountsLayout::roi_off_u32;   // zeroed with the block by enqueue_cascade (the tile tickets too)
            RoiArgs ra;
            CascadeArgs ca;
            fill_region_args(e, L, pl2, second, *p_second, W, H, nf, L->det_cap, &ra, &ca);
            ra.det_in = (const DetEntry*)L->d_det.p;
            ra.det_in_count = (const uint32_t*)L->d_counts.p + MAX_PASSES * CountsLayout::q_counts;
            ra.det_in_cap = L->det_cap;
            ra.scales_in = (const ScaleDev*)pl1->d_scales.p;
            HIP_TRY(hipEventRecord(L->launch_ev[2 * VJ_MAX_LAUNCHES - 2], e->stream));
            if (grouped) {
                ga.n_rois = ra.n_rois;
                const int grc = launch_group_rois(ga, e->stream);
                if (grc) {
                    set_error("grouping launch failed: %s", hipGetErrorString((hipError_t)grc));
                    return VJ_ERR_HIP;
                }
                HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::roi_off_u32 + 5, ga.overflow, 4, hipMemcpyDeviceToHost, e->stream));
            }
            const int hrc = launch_roi_chain(ra, ca, !grouped, pl2->trees, count2, pl2->general, std::max(1, e->n_cu * e->blocks_per_cu), e->stream,
                                             e->concurrent ? e->stream2 : nullptr, e->fork_ev, e->join_ev);
            if (hrc) {
                set_error("region pass launch failed: %s", hipGetErrorString((hipError_t)hrc));
                return VJ_ERR_HIP;
            }
            HIP_TRY(hipEventRecord(L->launch_ev[2 * VJ_MAX_LAUNCHES - 1], e->stream));
            // the region counters join the block enqueue_cascade already copies; copy them again now that they are final
            HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::roi_off_u32, roi_counts, 5 * 4, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::roi_off_u32 + 6, roi_counts + 6, 4, hipMemcpyDeviceToHost, e->stream));   // region tiles
            HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::roi_off_u32 + 8, roi_counts + 8, (size_t)VJ_MAX_STAGES * 2 * 4,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipMemcpyAsync((uint32_t*)L->h_pinned + CountsLayout::stage_off_u32 + 4, (uint32_t*)L->d_counts.p + CountsLayout::stage_off_u32 + 4, 16,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipEventRecord(L->done, e->stream));
            HIP_TRY(hipEventSynchronize(L->done));
            const uint32_t* hc = (const uint32_t*)L->h_pinned;
            n_det1 = hc[MAX_PASSES * CountsLayout::q_counts];
            n_units = hc[CountsLayout::roi_off_u32 + 1];
            n_det2 = hc[CountsLayout::roi_off_u32 + 4];
            if (hc[CountsLayout::roi_off_u32 + 2] != 0) {
                set_error("internal error: %u regions outside their frames", hc[CountsLayout::roi_off_u32 + 2]);
                return VJ_ERR_HIP;
            }
            if (n_det1 > L->det_cap) {   // as finish_batch would: grow, and run both cascades again
                uint32_t want = L->det_cap;
                while (want < n_det1) want *= 2;
                if ((rc = L->d_det.ensure((size_t)want * sizeof(DetEntry)))) return rc;
                L->det_cap = want;
                continue;
            }
            if (grouped && hc[CountsLayout::roi_off_u32 + 5] != 0) {
                // a frame with more raw candidates than the device kernel groups: the same result the long way round
                // (group on the host, hand the regions back as a list)
                vj_result_free(out_first);
                vj_result_free(out_second);
                if ((rc = vj_detect(e, first, frames, n_frames, p_first, out_first))) return rc;
                std::vector<vj_roi> rois(out_first->count);
                for (uint32_t i = 0; i < out_first->count; ++i) {
                    const vj_rect& r = out_first->rects[i];
                    rois[i] = vj_roi{r.frame, r.x, r.y, r.w, r.h};
                }
                return vj_detect_rois(e, second, frames, n_frames, rois.data(), (int)rois.size(), p_second, out_second);
            }
            const uint32_t n_tiles2 = hc[CountsLayout::roi_off_u32 + 6];
            if (n_units > e->roi_unit_cap || n_det2 > e->roi_det_cap || n_tiles2 > e->roi_tile_cap) {
                while (e->roi_unit_cap < n_units) e->roi_unit_cap *= 2;
                while (e->roi_det_cap < n_det2) e->roi_det_cap *= 2;
                while (e->roi_tile_cap < n_tiles2) e->roi_tile_cap *= 2;
                continue;
            }
            HIP_TRY(hipEventElapsedTime(&ms_roi, L->launch_ev[2 * VJ_MAX_LAUNCHES - 2], L->launch_ev[2 * VJ_MAX_LAUNCHES - 1]));
            ok = true;
        }
        if (!ok) {
            set_error("region buffers kept overflowing");
            return VJ_ERR_LIMIT;
        }
        // first cascade: decode (device order kept in raw1 for the region numbers)
        const size_t base1 = dets1.size();
        std::vector<DetEntry> raw1;
        if ((rc = finish_batch(e, L, pl1, f0, W, H, *p_first, &dets1, &out_first->counters, &out_first->timing, &raw1))) return rc;
        // second cascade
        std::vector<RoiDet> raw2(n_det2);
        if (n_det2) HIP_TRY(hipMemcpy(raw2.data(), e->d_roi_det.p, (size_t)n_det2 * sizeof(RoiDet), hipMemcpyDeviceToHost));
        std::vector<RoiDev> regions;
        size_t base_g = grouped1.size();
        if (grouped) {   // the first result is the region list itself
            const uint32_t n_rois = ((const uint32_t*)L->h_pinned)[CountsLayout::roi_off_u32 + 0];
            regions.resize(n_rois);
            std::vector<uint32_t> weights(n_rois);
            if (n_rois) {
                HIP_TRY(hipMemcpy(regions.data(), e->d_rois.p, (size_t)n_rois * sizeof(RoiDev), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(weights.data(), (const char*)e->d_group.p + roi_weight_off, (size_t)n_rois * 4u, hipMemcpyDeviceToHost));
            }
            for (uint32_t i = 0; i < n_rois; ++i)
                grouped1.push_back(vj_rect{regions[i].x, regions[i].y, regions[i].w, regions[i].h, (float)weights[i], regions[i].frame + f0, -1});
        }
        for (const RoiDet& d : raw2) {
            const uint32_t f = (uint32_t)(d.off / fbytes);
            const uint32_t el = (uint32_t)((d.off - (uint64_t)f * fbytes) / 4u);
            if (grouped) {
                const RoiDev& r1 = regions[d.roi];
                dets2.push_back(Det2{(int)(base_g + d.roi), d.slot, el % stride - (uint32_t)r1.x, el / stride - (uint32_t)r1.y});
            } else {
                const RawDet& r1 = dets1[base1 + d.roi];
                dets2.push_back(Det2{(int)(base1 + d.roi), d.slot, el % stride - r1.x, el / stride - r1.y});
            }
        }
        out_second->timing.cascade_ms += ms_roi;
        out_second->timing.total_ms += ms_roi;
        out_second->timing.n_cascade_launches = 1;
        if (count2) {
            const unsigned long long* se = (const unsigned long long*)((const uint32_t*)L->h_pinned + CountsLayout::roi_off_u32 + 8);
            const unsigned long long* tc = (const unsigned long long*)((const uint32_t*)L->h_pinned + CountsLayout::stage_off_u32) + 2;
            child2_nodes += tc[0];
            child2_rects += tc[1];
            for (size_t s2 = 0; s2 < pl2->stages.size(); ++s2) out_second->counters.stage_entered[s2] += se[s2];
        }
    }
    if (bal && bal->phase != 3) {   // (as vj_detect: one sub-batch, timed kernel variants only)
        bool any_tile_scale = false;
        for (const ScaleDev& sd : pl1->scales) any_tile_scale |= sd.tile_rw != 0;
        if (!any_tile_scale) {
            bal->cur = bal->best;
            bal->phase = 3;
        } else if (!(p_first->flags & VJ_FLAG_COUNTERS) && (uint64_t)n_frames <= max_frames && n_frames == bal->n_ref) {
            balance_report(bal, out_first->timing.cascade_ms / (float)n_frames, !e->tile_thresholds_set);
        }
    }
    out_first->timing.tile_split = pl1->tile_split;
    out_first->timing.balance_state = !bal ? 0 : bal->phase == 3 ? 2 : 1;
    out_first->timing.balance_calls = bal ? bal->calls : 0;
    // the first result in its sorted order; regions are numbered by their position in it
    std::vector<uint32_t> rank;
    if (grouped) {   // frames in order, groups in cv::partition's class order: already the result's order
        rank.resize(grouped1.size());
        for (size_t i = 0; i < rank.size(); ++i) rank[i] = (uint32_t)i;
        out_first->count = (uint32_t)grouped1.size();
        if (!grouped1.empty()) {
            out_first->rects = (vj_rect*)malloc(grouped1.size() * sizeof(vj_rect));
            if (!out_first->rects) return VJ_ERR_NOMEM;
            memcpy(out_first->rects, grouped1.data(), grouped1.size() * sizeof(vj_rect));
        }
        fill_counters(pl1, n_frames, *p_first, out_first);
    } else {
        std::vector<uint32_t> order(dets1.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return std::tie(dets1[a].frame, dets1[a].slot, dets1[a].y, dets1[a].x) < std::tie(dets1[b].frame, dets1[b].slot, dets1[b].y, dets1[b].x);
        });
        rank.resize(dets1.size());
        for (size_t i = 0; i < order.size(); ++i) rank[order[i]] = (uint32_t)i;
        if ((rc = build_result(pl1, dets1, n_frames, *p_first, out_first))) return rc;
    }
    std::sort(dets2.begin(), dets2.end(), [&](const Det2& a, const Det2& b) {
        return std::make_tuple(rank[a.roi], a.slot, a.y, a.x) < std::make_tuple(rank[b.roi], b.slot, b.y, b.x);
    });
    out_second->count = (uint32_t)dets2.size();
    if (!dets2.empty()) {
        out_second->rects = (vj_rect*)malloc(dets2.size() * sizeof(vj_rect));
        if (!out_second->rects) return VJ_ERR_NOMEM;
        for (size_t i = 0; i < dets2.size(); ++i) {
            const vj_scale_info& si = pl2->scales_info[dets2[i].slot];
            out_second->rects[i] = vj_rect{(int32_t)dets2[i].x, (int32_t)dets2[i].y, si.win_w, si.win_h, 0.0f, (int32_t)rank[dets2[i].roi],
                                           si.scale_idx};
        }
    }
    if (p_second->min_neighbors != 0 && out_second->count) {
        rc = vj_group_rectangles(out_second->rects, &out_second->count, (int)std::max<uint32_t>(p_second->min_neighbors, 1u), 0.2);
        if (rc) return rc;
    }
    if (count2) {
        vj_counters& k = out_second->counters;
        k.windows = k.stage_entered[0];
        uint64_t rect_evals = 0;
        for (size_t s2 = 0; s2 < pl2->stages.size(); ++s2) {   // (as fill_counters: roots on the host, visited child nodes from the device)
            k.stump_evals += k.stage_entered[s2] * (pl2->trees ? pl2->prog.n_roots[s2] : pl2->prog.n_nodes[s2]);
            rect_evals += k.stage_entered[s2] * (pl2->trees ? pl2->prog.n_root_rects[s2] : pl2->prog.n_rects[s2]);
        }
        if (pl2->trees) {
            k.stump_evals += child2_nodes;
            rect_evals += child2_rects;
        }
        k.gather_bytes = 48ull * k.windows + 16ull * rect_evals;
    }
    return VJ_OK;
}

// ------------------------------------------------------------------ frame streams
struct vj_stream {
    vj_env* e = nullptr;
    vj_params p;
    int W = 0, H = 0, CH = 1, max_batch = 0;
    std::unique_ptr<Plan> plan;     // private: its queue layout must not change while a batch is in flight
    Lane lanes[2];
    hipStream_t copy = nullptr;
    int fifo[2] = {0, 0};           // lanes holding submitted batches, oldest first
    int n_pending = 0;
    int n_frames_of[2] = {0, 0};
};

int vj_stream_create(vj_env* e, const vj_cascade* c, int width, int height, int channels, int max_batch, const vj_params* p,
                     vj_stream** out) {
    if (!out) return VJ_ERR_ARG;
    *out = nullptr;
    if (!e || !c || !p || width <= 0 || height <= 0 || max_batch <= 0) return VJ_ERR_ARG;
    const int CH = channels <= 1 ? 1 : channels;
    if (CH != 1 && CH != 3 && CH != 4) return VJ_ERR_ARG;
    if (check_params(*p)) return VJ_ERR_ARG;
    if ((uint64_t)(width + 1) * (uint64_t)(height + 3) >= (1ull << 30)) {
        set_error("image too large");
        return VJ_ERR_LIMIT;
    }
    HIP_TRY(hipSetDevice(e->device));
    auto s = std::make_unique<vj_stream>();
    s->e = e;
    s->p = *p;
    s->W = width;
    s->H = height;
    s->CH = CH;
    s->max_batch = max_batch;
    s->plan = std::make_unique<Plan>();
    struct Guard {   // releases what was created when a later step fails
        vj_stream* s;
        ~Guard() { if (s) vj_stream_destroy(s); }
    } guard{nullptr};
    // (a chain balance already found for this workload by vj_detect's feedback is taken over; a stream does not search itself)
    const vj_env::Balance* bal = balance_of(e, c, width, height, *p, max_batch, false);
    TileThresholds sth{e->tile_min_windows, e->tile_accept_windows, e->tile_max_dwords_per_window};
    if (bal && balance_choice(bal, -1).thr == 1 && !e->tile_thresholds_set) sth = TileThresholds{std::min(384, sth.min_windows), std::min(384, sth.accept_windows), sth.max_dwords_per_window};
    int rc = build_plan(e, *c, width, height, *p, s->plan.get(), bal ? bal->best : e->split_for(max_batch, *p), sth);
    if (rc) {
        s->plan->release_device();
        return rc;
    }
    vj_stream* raw = s.release();
    guard.s = raw;
    if ((uint64_t)max_batch > max_frames_per_subbatch(e, raw->plan.get())) {
        set_error("max_batch %d exceeds what one launch sequence can address at this frame size (%llu frames)", max_batch,
                  (unsigned long long)max_frames_per_subbatch(e, raw->plan.get()));
        return VJ_ERR_LIMIT;
    }
    for (Lane& L : raw->lanes) {
        if ((rc = L.create())) return rc;
        L.shared_integrals = true;
    }
    HIP_TRY(hipStreamCreateWithFlags(&raw->copy, hipStreamNonBlocking));
    // every buffer at its final size now: nothing is (re)allocated while batches are in flight
    for (Lane& L : raw->lanes)
        if ((rc = ensure_image_buffers(e, width, height, max_batch, true, CH, &L))) return rc;
    uint64_t q_entries = 0;
    if ((rc = layout_queues(raw->plan.get(), max_batch, &q_entries))) return rc;
    const size_t n_pass = raw->plan->pass_bounds.size() - 1;
    for (size_t ps = 1; ps < n_pass; ++ps)
        if ((rc = e->d_q[ps].ensure(std::max<uint64_t>(q_entries, 1) * sizeof(QEntry)))) return rc;
    guard.s = nullptr;
    *out = raw;
    return VJ_OK;
}

int vj_stream_submit(vj_stream* s, const vj_image* frames, int n_frames) {
    if (!s || !frames || n_frames <= 0 || n_frames > s->max_batch) return VJ_ERR_ARG;
    if (s->n_pending >= 2) {
        set_error("two batches are pending: collect one before submitting the next");
        return VJ_ERR_LIMIT;
    }
    int W, H, CH;
    int rc = check_frames(frames, n_frames, &W, &H, &CH);
    if (rc) return rc;
    if (W != s->W || H != s->H || CH != s->CH) {
        set_error("frames do not have the stream's size / channel count");
        return VJ_ERR_ARG;
    }
    vj_env* e = s->e;
    HIP_TRY(hipSetDevice(e->device));
    const int lane = s->n_pending == 0 ? 0 : 1 - s->fifo[0];
    Lane* L = &s->lanes[lane];
    // the lane's frame buffer is free once the integral kernels of its previous batch have read it
    HIP_TRY(hipEventSynchronize(L->integral_done));
    if ((rc = enqueue_prepare(e, L, s->plan.get(), frames, n_frames, W, H, true, s->copy))) return rc;
    if ((rc = enqueue_cascade(e, L, s->plan.get(), W, H, s->p))) return rc;
    s->fifo[s->n_pending] = lane;
    s->n_frames_of[lane] = n_frames;
    s->n_pending++;
    return VJ_OK;
}

int vj_stream_collect(vj_stream* s, vj_result* out) {
    if (!s || !out) return VJ_ERR_ARG;
    memset(out, 0, sizeof(*out));
    if (s->n_pending == 0) {
        set_error("no batch is pending");
        return VJ_ERR_ARG;
    }
    vj_env* e = s->e;
    HIP_TRY(hipSetDevice(e->device));
    const int lane = s->fifo[0];
    s->fifo[0] = s->fifo[1];
    s->n_pending--;
    Lane* L = &s->lanes[lane];
    std::vector<RawDet> dets;
    int rc = finish_batch(e, L, s->plan.get(), 0, s->W, s->H, s->p, &dets, &out->counters, &out->timing);
    if (rc) return rc;
    return build_result(s->plan.get(), dets, s->n_frames_of[lane], s->p, out);
}

void vj_stream_destroy(vj_stream* s) {
    if (!s) return;
    (void)hipSetDevice(s->e->device);
    if (s->e->stream) (void)hipStreamSynchronize(s->e->stream);
    if (s->e->stream2) (void)hipStreamSynchronize(s->e->stream2);
    if (s->copy) {
        (void)hipStreamSynchronize(s->copy);
        (void)hipStreamDestroy(s->copy);
    }
    for (Lane& L : s->lanes) L.destroy();
    if (s->plan) s->plan->release_device();
    delete s;
}

void vj_result_free(vj_result* r) {
    if (!r) return;
    free(r->rects);
    r->rects = nullptr;
    r->count = 0;
}

}  // extern "C"
