// Host driver of the OpenCV arithmetic profile: cvHaarDetectObjects' scale-cascade path as the reference
// keeps it in tempcv.cpp (its private copy of OpenCV 2.4.2 haar.cpp; call site main.cpp:145), on the GPU.
//   scale loop, window grid            tempcv.cpp:1344-1417
//   cvSetImagesForHaarClassifierCascade  tempcv.cpp:549-768 (equRect, cvRound-ed rectangles, f32 weights,
//                                        CV_ADJUST_WEIGHTS = 0; the "align blocks" flags can never be set:
//                                        kx = r0.width / base_w >= 1)
//   stage threshold bias               tempcv.cpp:262, 419
//   hidden-cascade flags               tempcv.cpp:410-470 (isStumpBased, is_tree, per-stage two_rects) — they select
//                                        the arithmetic of a node sum (vj_cv_profile.hip: cv_node_sum)
// Stumps or multi-node trees, linear cascades or stage trees, upright or tilted features (tilted integral).
// Second arithmetic profile.  OpenCV itself is not available here or on the GPU box, so
// parity is against the oracle's restatement of the same lines (oc_detect_opencvlike): unpinned.
#include "vj_env_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace vj;

namespace {

inline int cv_round(double v) { return (int)std::lrint(v); }   // cvRound: round half to even

struct CvScaleHost {
    double factor;
    int idx;
    int win_w, win_h, end_x, end_y;
};

// Everything that depends on (cascade, frame size, parameters) only: scales, feature tables, stage records, row list.
static int build_cv_plan(vj_env* e, const vj_cascade* c, int W, int H, const vj_cv_params* p, bool small_batch, CvPlan* pl) {
    if ((int)c->stages.size() > VJ_MAX_STAGES || c->stages.empty()) {
        set_error("cascade has %zu stages; 1..%d are supported", c->stages.size(), VJ_MAX_STAGES);
        return VJ_ERR_LIMIT;
    }
    pl->prog = build_stage_program(*c);
    const StageProgram& prog = pl->prog;
    std::vector<uint32_t> order;
    if (!stage_sweep_order(prog, &order)) {
        set_error("stage links form a cycle");
        return VJ_ERR_UNSUPPORTED;
    }
    // icvCreateHidHaarClassifierCascade's flags (tempcv.cpp:410-470)
    bool trees = false, is_tree = false, has_tilted = false;
    for (const auto& st : c->stages) is_tree |= st.next != -1;
    for (const auto& t : c->trees)
        if (t.n_nodes != 1) trees = true;
    for (const auto& nd : c->nodes) has_tilted |= nd.tilted != 0;
    // two-node trees (frontalface_alt2): root + one node child — the shape the tile kernel's tree path knows
    bool tree2 = trees;
    for (const auto& t : c->trees) {
        if (!tree2) break;
        if (t.n_nodes != 2) { tree2 = false; break; }
        const vj_node_desc& n0 = c->nodes[t.first_node];
        const vj_node_desc& n1 = c->nodes[t.first_node + 1];
        const int kids = (n0.left > 0) + (n0.right > 0);
        if (kids != 1 || (n0.left > 0 ? n0.left : n0.right) != 1 || n1.left > 0 || n1.right > 0) tree2 = false;
    }
    pl->tree2 = tree2;
    pl->trees = trees;
    pl->is_tree = is_tree;
    pl->has_tilted = has_tilted;
    pl->n_order = (uint32_t)order.size();
    pl->n_stages = (uint32_t)c->stages.size();
    std::vector<uint8_t> two_rects(c->stages.size(), 1);
    for (size_t s2 = 0; s2 < c->stages.size(); ++s2)
        for (int t = 0; t < c->stages[s2].n_trees; ++t) {
            const vj_tree_desc& td = c->trees[c->stages[s2].first_tree + t];
            for (int l = 0; l < td.n_nodes; ++l) {
                const vj_rect_desc& r2 = c->nodes[td.first_node + l].rect[2];
                // :452-457: the third rectangle counts unless |weight| < DBL_EPSILON or it is empty
                if (!(std::fabs((double)r2.weight) < 2.220446049250313e-16 || r2.w == 0 || r2.h == 0)) two_rects[s2] = 0;
            }
        }
    // ---- the scale loop (tempcv.cpp:1344-1377)
    const uint32_t stride = (uint32_t)W + 1u;
    std::vector<CvScaleHost> hs;
    {
        int n_factors = 0;
        double factor = 1;
        for (; factor * c->win_w < W - 10 && factor * c->win_h < H - 10; n_factors++, factor *= p->scale_factor) {}
        factor = 1;
        for (int k = 0; k < n_factors; ++k, factor *= p->scale_factor) {
            const double ystep = std::max(2., factor);
            CvScaleHost s;
            s.factor = factor;
            s.idx = k;
            s.win_w = cv_round(c->win_w * factor);
            s.win_h = cv_round(c->win_h * factor);
            s.end_x = cv_round((W - s.win_w) / ystep);
            s.end_y = cv_round((H - s.win_h) / ystep);
            if (s.win_w < p->min_w || s.win_h < p->min_h) continue;
            if (s.end_x <= 0 || s.end_y <= 0) continue;
            hs.push_back(s);
        }
    }
    // stage trees: the tiles run the tree's linear prefix — leading stages of the sweep order that reject outright and pass
    // on to the next one (cv_profile_pass finds the same prefix) — when the prefix is stages 0, 1, 2, ... themselves
    uint32_t tree_prefix = 0;
    if (is_tree) {
        while (tree_prefix + 1u < order.size() && order[tree_prefix] == tree_prefix && prog.on_fail[tree_prefix] == STAGE_REJECT &&
               prog.on_pass[tree_prefix] == (int)order[tree_prefix + 1u])
            ++tree_prefix;
    }
    pl->tree_prefix = tree_prefix;
    // Is the rest of the tree a sequence of chains (CvChainDev)?  A chain is a run of sweep positions whose pass edges follow
    // the order and end in an accept, whose rejects all go to ONE place: nowhere (final) or the first stage of the NEXT chain.
    if (is_tree && tree_prefix != 0u && prog.on_pass[order[tree_prefix - 1u]] == (int)order[tree_prefix]) {
        CvChainDev ch;
        memset(&ch, 0, sizeof(ch));
        bool ok = true;
        uint32_t b = tree_prefix;
        while (b < order.size() && ok) {
            if (ch.n == 4u) { ok = false; break; }
            const int f = prog.on_fail[order[b]];
            uint32_t e2 = b;
            while (true) {
                const uint32_t sid = order[e2];
                if (prog.on_fail[sid] != f) { ok = false; break; }
                const int np = prog.on_pass[sid];
                ++e2;
                if (np == STAGE_ACCEPT) break;
                if (e2 >= order.size() || np != (int)order[e2]) { ok = false; break; }
            }
            if (!ok) break;
            ch.begin[ch.n] = b;
            ch.end[ch.n] = e2;
            if (f != STAGE_REJECT) {
                if (e2 < order.size() && f == (int)order[e2]) ch.chained |= 1u << ch.n;   // rejects start the next chain
                else ok = false;
            }
            ++ch.n;
            b = e2;
        }
        ch.tail_max = (uint32_t)std::max(0, std::min(e->cv_tail_max, (int)CV_TAIL_MAX));
        if (ok && ch.n != 0u && !((ch.chained >> (ch.n - 1u)) & 1u)) pl->chains = ch;
    }
    // (a stage tree's row kernel evaluates the whole tree at every grid position and is far slower per window than the tiles'
    // prefix: trees send every scale they can to tiles and leave the row kernel two workgroups per CU)
    // (linear cascades, profiles/r04_notes.md #5b: stumps 2 workgroups x tiles of >= 2048 windows — 64 x 1080p frontalface_alt 66.6 ms against 80.5
    // at 3 x 1536, frontalface_default 51.4 / 57.1, 256 x 720p 112.7 / 136.9 —, multi-node trees the other way round: frontalface_alt2 99.3 / 77.5;
    // two-node trees with both nodes fetched at once (CvArgs::tree2, #5d): 2 x 1536 69.9 ms against 80.9 at 3 x 1536)
    const bool rows_tree2 = pl->tree2 && !is_tree && !has_tilted && e->cv_tree2;
    // (cascades with tilted features, #12: a tile holds two images, so the shapes that fit are smaller — and the row kernel they relieve is 2-3x
    // the tiles' cost per window: two workgroups x tiles of >= 512 windows, `fullbody` 16 x 1080p 28-29 ms against 40.8 with every tile of
    // >= 2048 windows refused, `upperbody` 50.7 / 96.6, `mcs_righteye`'s kernels 30.9 / 72.4)
    const bool tilt_tiles = has_tilted && e->cv_tiles_tilted && !is_tree;
    const int lin_blocks = e->cv_row_blocks > 0 ? e->cv_row_blocks : tilt_tiles ? 2 : trees && !rows_tree2 ? 3 : 2;
    const int lin_min_windows = e->cv_tile_min_windows > 0 ? e->cv_tile_min_windows : tilt_tiles ? 512 : trees ? 1536 : 2048;
    pl->row_blocks = is_tree ? e->cv_row_blocks_tree : lin_blocks;
    // a call of <= 4 frames is bound by latency, not by the balance of two saturated chains: one row-kernel workgroup per CU
    // and every scale that has a tile of 512 windows on tiles (one 1080p frame: 2.4 -> 2.0 ms)
    int min_windows = is_tree ? e->cv_tile_min_windows_tree : lin_min_windows, min_windows0 = tilt_tiles ? std::min(e->cv_tile_min_windows0, 512) : e->cv_tile_min_windows0;
    if (small_batch && !is_tree) {
        pl->row_blocks = 1;
        min_windows = std::min(min_windows, 512);
        min_windows0 = std::min(min_windows0, 1024);
    }
    const size_t n_nodes = c->nodes.size();
    std::vector<CvScaleDev>& scales = pl->scales;
    scales.assign(hs.size(), CvScaleDev{});
    std::vector<CvNodeRec> table(hs.size() * n_nodes);
    table.reserve(2 * hs.size() * n_nodes);   // tile copies of the small scales' records are appended: no reallocation, `recs` stays valid
    std::vector<UnitDev> rows;
    std::vector<int> tile_class(hs.size(), -1);
    bool reach_ok = true;
    const uint32_t frame_elems = frame_elems_for(W, H);
    for (size_t k = 0; k < hs.size(); ++k) {
        uint64_t max_reach = 0;   // furthest element a feature of this scale touches, from the window origin
        const double scale = hs[k].factor;
        CvScaleDev& sd = scales[k];
        memset(&sd, 0, sizeof(sd));
        sd.ystep = std::max(2., scale);
        // equRect (tempcv.cpp:607-611): x = y = cvRound(scale), (orig - 2) * scale rounded
        const int ex = cv_round(scale), ew = cv_round((c->win_w - 2) * scale), eh = cv_round((c->win_h - 2) * scale);
        const double weight_scale = 1. / (ew * eh);
        sd.inv_area = weight_scale;
        sd.win_w = (uint32_t)hs[k].win_w;
        sd.win_h = (uint32_t)hs[k].win_h;
        sd.end_x = (uint32_t)hs[k].end_x;
        sd.end_y = (uint32_t)hs[k].end_y;
        sd.q0 = (uint32_t)ex * stride + (uint32_t)ex;
        sd.q1 = sd.q0 + (uint32_t)ew;
        sd.q2 = (uint32_t)(ex + eh) * stride + (uint32_t)ex;
        sd.q3 = sd.q2 + (uint32_t)ew;
        sd.table_first = (uint32_t)(k * n_nodes);
        sd.scale_idx = (uint32_t)hs[k].idx;
        CvNodeRec* recs = table.data() + k * n_nodes;
        for (size_t t = 0; t < c->trees.size(); ++t) {
            const vj_tree_desc& td = c->trees[t];
            for (int j = 0; j < td.n_nodes; ++j) {
                const vj_node_desc& nd = c->nodes[td.first_node + j];
                CvNodeRec& r = recs[td.first_node + j];
                memset(&r, 0, sizeof(r));
                if (nd.rect[0].weight == 0.0f || nd.rect[1].weight == 0.0f) {
                    set_error("node %d: rect 0 and rect 1 must both be weighted", td.first_node + j);
                    return VJ_ERR_UNSUPPORTED;
                }
                double sum0 = 0, area0 = 0;
                const double correction_ratio = weight_scale * (!nd.tilted ? 1 : 0.5);   // :731
                for (int q = 0; q < 3; ++q) {
                    // hidfeature->rect[k].p0 == 0 ends the list (tempcv.cpp:663): only a third rectangle can be absent (:452-455)
                    if (q == 2 && (std::fabs((double)nd.rect[2].weight) < 2.220446049250313e-16 || nd.rect[2].w == 0 || nd.rect[2].h == 0))
                        break;
                    const int tx = cv_round(nd.rect[q].x * scale), ty = cv_round(nd.rect[q].y * scale);
                    const int tw = cv_round(nd.rect[q].w * scale), th = cv_round(nd.rect[q].h * scale);
                    // corners p0, p1 = p0 + da, p2 = p0 + db, p3 = p0 + da + db (element offsets)
                    const int64_t p0 = (int64_t)ty * stride + tx;
                    int64_t da, db;
                    if (!nd.tilted) {         // :735-741
                        da = tw;
                        db = (int64_t)th * stride;
                    } else {                  // :743-750: p1 = (y + h, x - h), p2 = (y + w, x + w), p3 = (y + w + h, x + w - h)
                        da = (int64_t)th * stride - th;
                        db = (int64_t)tw * stride + tw;
                    }
                    if (p0 < 0 || da < 0 || db < 0 || (p0 + da + db) * 4 > 0x7fffffffll) {
                        set_error("feature offsets exceed the device record range");
                        return VJ_ERR_LIMIT;
                    }
                    r.lt[q] = (uint32_t)(p0 * 4);
                    r.da[q] = (uint32_t)(da * 4);
                    r.db[q] = (uint32_t)(db * 4);
                    r.w[q] = (float)(nd.rect[q].weight * correction_ratio);
                    if (q == 0)
                        area0 = tw * th;
                    else
                        sum0 += r.w[q] * tw * th;                          // float * int * int, added to a double (:756)
                    max_reach = std::max<uint64_t>(max_reach, (uint64_t)(p0 + da + db));
                    max_reach = std::max<uint64_t>(max_reach, (uint64_t)(p0 + db));
                }
                r.w[0] = (float)(-sum0 / area0);
                r.thr = nd.threshold;
                uint32_t flags = nd.tilted ? CV_NODE_TILTED : 0u;
                auto leaf_or_node = [&](int v, uint32_t flag, uint32_t* dst) {
                    if (v > 0) {
                        flags |= flag;
                        *dst = (uint32_t)v;
                    } else {
                        const float a = c->alpha[td.first_alpha - v];
                        memcpy(dst, &a, 4);
                    }
                };
                leaf_or_node(nd.left, NODE_LEFT_IS_NODE, &r.left);
                leaf_or_node(nd.right, NODE_RIGHT_IS_NODE, &r.right);
                if (j == td.n_nodes - 1) flags |= NODE_TREE_LAST;
                r.flags = flags;
            }
        }
        // evaluated windows satisfy x + win_w <= W and y + win_h <= H (border rule); a feature may overshoot its
        // window by one column / row (separate rounding): the frame allocation has two zeroed slack rows for that
        const uint64_t origin_max = (uint64_t)(H - hs[k].win_h) * stride + (uint64_t)(W - hs[k].win_w);
        if (origin_max + max_reach >= (uint64_t)frame_elems) reach_ok = false;
        for (uint32_t iy = 0; iy < sd.end_y; ++iy) rows.push_back(UnitDev{(uint32_t)k, iy, 0, 0});

        // ---- LDS-tile path (vj_cv_tile.hip): stump cascades with linear stages and upright features.  A tile is tw x th
        // windows (tw divides 64, so a tile row never straddles a word of the reject / visited bitmap); its footprint is
        // the span of its window origins plus the furthest corner any feature or the equRect reaches.  Two LDS classes
        // like the clod profile's tiles: two workgroups per CU or one, next to one workgroup of cv_profile_pass.
        // Tilted features (round 4): the tile's footprint of the TILTED integral is staged right behind the sum's (same origin, pitch and
        // rows: a tilted rectangle's corners (y, x), (y + h, x - h), (y + w, x + w), (y + w + h, x + w - h) lie inside the window's box), a
        // tilted node's record carries that distance in its corner offsets, and the kernel's node code does not change.
        const bool tiles_tilted = has_tilted && e->cv_tiles_tilted && !is_tree;
        if (e->cv_tiles && (!trees || (tree2 && !is_tree)) && (!is_tree || tree_prefix != 0u) && (!has_tilted || tiles_tilted) && sd.end_x < 65536u &&
            sd.end_y < 65536u) {
            uint32_t reach_x = (uint32_t)(ex + ew), reach_y = (uint32_t)(ex + eh);
            for (size_t n = 0; n < n_nodes; ++n) {
                const CvNodeRec& r = recs[n];
                for (int q = 0; q < 3; ++q)
                    if (q < 2 || r.w[2] != 0.0f) {
                        const uint32_t p0 = r.lt[q] / 4u;
                        if (r.flags & CV_NODE_TILTED) {   // da = h * (stride - 1), db = w * (stride + 1): rightmost corner x + w, lowest y + w + h
                            const uint32_t hh = (r.da[q] / 4u) / (stride - 1u), ww = (r.db[q] / 4u) / (stride + 1u);
                            reach_x = std::max(reach_x, p0 % stride + ww);
                            reach_y = std::max(reach_y, p0 / stride + ww + hh);
                        } else {
                            reach_x = std::max(reach_x, p0 % stride + r.da[q] / 4u);
                            reach_y = std::max(reach_y, p0 / stride + (r.db[q] / 4u) / stride);
                        }
                    }
            }
            const uint32_t images = has_tilted ? 2u : 1u;   // LDS images per tile
            static const uint32_t kTw[] = {64, 32, 16}, kTh[] = {32, 24, 16, 12, 8, 4};
            // LDS budget of a CU: the row kernel's workgroups (20 KiB each) stay resident next to two tile workgroups
            // of class 0 or one of class 1; a tile workgroup also owns CVT_LDS_HEADER bytes of queues
            const uint32_t avail = 160u * 1024u - (uint32_t)pl->row_blocks * 20u * 1024u - 1024u;
            const uint32_t class_bytes[2] = {avail / 2u - (uint32_t)CVT_LDS_HEADER, avail - (uint32_t)CVT_LDS_HEADER};
            uint32_t best_n = 0, b_tw = 0, b_th = 0, b_pitch = 0, b_rows = 0;
            int b_cls = -1;
            for (int cls = 0; cls < 2 && b_cls < 0; ++cls) {
                for (uint32_t tw : kTw)
                    for (uint32_t th : kTh) {
                        const uint32_t pitch = (((uint32_t)std::ceil((double)(tw - 1) * sd.ystep) + 3u + reach_x) + 3u) & ~3u;
                        const uint32_t trows = (uint32_t)std::ceil((double)(th - 1) * sd.ystep) + 3u + reach_y;
                        if ((uint64_t)pitch * trows * 4u * images > class_bytes[cls]) continue;
                        const uint32_t nwin = std::min(tw, sd.end_x) * std::min(th, sd.end_y);
                        if (nwin > best_n) { best_n = nwin; b_tw = tw; b_th = th; b_pitch = pitch; b_rows = trows; }
                    }
                // a class-0 tile must be worth two workgroups per CU; else try the larger class
                if (best_n >= (uint32_t)(cls == 0 ? min_windows0 : min_windows))
                    b_cls = cls;
                else
                    best_n = 0;
            }
            if (b_cls >= 0) {
                sd.tile_tw = b_tw;
                sd.tile_th = b_th;
                sd.tile_pitch = b_pitch;
                sd.tile_rows = b_rows;
                sd.tile_table_first = (uint32_t)table.size();
                tile_class[k] = b_cls;
                table.resize(table.size() + n_nodes);   // (within the reserved capacity)
                CvNodeRec* trec = table.data() + sd.tile_table_first;
                for (size_t n = 0; n < n_nodes; ++n) {
                    trec[n] = recs[n];
                    for (int q = 0; q < 3; ++q)
                        if (q < 2 || recs[n].w[2] != 0.0f) {
                            const uint32_t p0 = recs[n].lt[q] / 4u;
                            if (recs[n].flags & CV_NODE_TILTED) {   // in the tilted image behind the sum image, corner steps in the tile's pitch
                                const uint32_t hh = (recs[n].da[q] / 4u) / (stride - 1u), ww = (recs[n].db[q] / 4u) / (stride + 1u);
                                trec[n].lt[q] = (b_pitch * b_rows + (p0 / stride) * b_pitch + p0 % stride) * 4u;
                                trec[n].da[q] = (hh * b_pitch - hh) * 4u;
                                trec[n].db[q] = (ww * b_pitch + ww) * 4u;
                            } else {
                                const uint32_t hh = (recs[n].db[q] / 4u) / stride;
                                trec[n].lt[q] = ((p0 / stride) * b_pitch + p0 % stride) * 4u;
                                trec[n].db[q] = hh * b_pitch * 4u;
                            }
                        }
                }
            }
        }
    }
    // tiles of one frame by LDS class; rows of the scales that stay on cv_profile_pass; one recurrence domain per window
    // row of a tile scale in the per-frame bitmap
    std::vector<UnitDev> tiles, rows_rest, bit_segs;
    pl->n_tile_scales = 0;
    {
        uint32_t word = 0;
        for (size_t k = 0; k < hs.size(); ++k) {
            CvScaleDev& sd = scales[k];
            if (sd.tile_th == 0u) continue;
            sd.tq_slot = pl->n_tile_scales;
            sd.tq_win_first = (uint32_t)std::min<uint64_t>(pl->tile_windows, 0xffffffffull);
            ++pl->n_tile_scales;
            pl->tile_windows += (uint64_t)sd.end_x * sd.end_y;
            sd.bits_base = word;
            const uint32_t wpr = (sd.end_x + 63u) / 64u;
            for (uint32_t iy = 0; iy < sd.end_y; ++iy) bit_segs.push_back(UnitDev{(uint32_t)k, word + iy * wpr, wpr, 0});
            word += wpr * sd.end_y;
        }
        pl->bits_frame_words = word;
        for (int cls = 0; cls < 2; ++cls) {
            pl->class_first[cls] = (uint32_t)tiles.size();
            uint32_t lds = 0;
            for (size_t k = 0; k < hs.size(); ++k) {
                const CvScaleDev& sd = scales[k];
                if (sd.tile_th == 0u || tile_class[k] != cls) continue;
                lds = std::max(lds, (uint32_t)CVT_LDS_HEADER + sd.tile_pitch * sd.tile_rows * 4u * (has_tilted ? 2u : 1u));
                for (uint32_t iy0 = 0; iy0 < sd.end_y; iy0 += sd.tile_th)
                    for (uint32_t ix0 = 0; ix0 < sd.end_x; ix0 += sd.tile_tw) tiles.push_back(UnitDev{(uint32_t)k, ix0 | (iy0 << 16), 0, 0});
            }
            pl->class_lds[cls] = lds;
        }
        pl->class_first[2] = (uint32_t)tiles.size();
        // Band-major row order (cv_row_band_px != 0): the rows of ALL scales whose top lies in one band of the image, band after
        // band.  The waves of an XCD walk one contiguous piece of the (frame, row) list together (cv_profile_pass), so what they
        // gather from at one time is a band of one frame's integral image — within the XCD's 4 MB of L2 — instead of a whole
        // scale's rows, i.e. the whole 8.3 MB image (rows are independent: the skip rule runs along a row).
        if (e->cv_row_band_px > 0) {
            const double band = (double)e->cv_row_band_px;
            std::stable_sort(rows.begin(), rows.end(), [&](const UnitDev& x, const UnitDev& y) {
                const uint32_t bx = (uint32_t)((double)x.first * scales[x.scale].ystep / band), by = (uint32_t)((double)y.first * scales[y.scale].ystep / band);
                return bx != by ? bx < by : x.scale != y.scale ? x.scale < y.scale : x.first < y.first;
            });
        }
        for (const UnitDev& r : rows)
            if (scales[r.scale].tile_th == 0u) rows_rest.push_back(r);
    }
    std::vector<StageDev> stages(c->stages.size());
    for (size_t s = 0; s < c->stages.size(); ++s) {
        memset(&stages[s], 0, sizeof(StageDev));
        stages[s].first_node = prog.first_node[s];
        stages[s].n_nodes = prog.n_nodes[s];
        stages[s].threshold = c->stages[s].threshold - 0.0001f;   // icv_stage_threshold_bias, in f32
        stages[s].n_trees = (uint32_t)c->stages[s].n_trees;
        stages[s].on_pass = prog.on_pass[s];
        stages[s].on_fail = prog.on_fail[s];
        stages[s].order = s < order.size() ? order[s] : 0u;
        // an f64 product per rectangle only on cvRunHaarClassifierCascadeSum's stump path (:863-888)
        stages[s].cv_f64 = (two_rects[s] && !trees && !is_tree) ? 1u : 0u;
        // wave-split finish of the tile kernel: bound on the difference between any two summation orders of the stage's
        // leaf values (the f32 form of the clod profile's bound, build_plan; the kernel scales it to f64's unit roundoff)
        double amax = 0.0;
        for (int t = 0; t < c->stages[s].n_trees; ++t) {
            const vj_tree_desc& td = c->trees[c->stages[s].first_tree + t];
            double m = 0.0;
            for (int k = 0; k <= td.n_nodes; ++k) m = std::max(m, (double)std::fabs(c->alpha[td.first_alpha + k]));
            amax += m;
        }
        stages[s].sp_delta = (float)(4.0 * (double)prog.n_nodes[s] * std::ldexp(1.0, -24) * amax * 1.001 + 1e-30);
    }
    if (!reach_ok) {
        set_error("feature reach exceeds the frame allocation");
        return VJ_ERR_LIMIT;
    }

    int rc;
    if ((rc = pl->d_table.ensure(std::max<size_t>(table.size(), 1) * sizeof(CvNodeRec)))) return rc;
    if ((rc = pl->d_scales.ensure(std::max<size_t>(scales.size(), 1) * sizeof(CvScaleDev)))) return rc;
    if ((rc = pl->d_stages.ensure(stages.size() * sizeof(StageDev)))) return rc;
    if ((rc = pl->d_rows.ensure(std::max<size_t>(rows.size(), 1) * sizeof(UnitDev)))) return rc;
    if ((rc = pl->d_tiles.ensure(std::max<size_t>(tiles.size(), 1) * sizeof(UnitDev)))) return rc;
    if ((rc = pl->d_rows_rest.ensure(std::max<size_t>(rows_rest.size(), 1) * sizeof(UnitDev)))) return rc;
    if ((rc = pl->d_bit_segs.ensure(std::max<size_t>(bit_segs.size(), 1) * sizeof(UnitDev)))) return rc;
    if (!tiles.empty()) HIP_TRY(hipMemcpy(pl->d_tiles.p, tiles.data(), tiles.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    if (!rows_rest.empty()) HIP_TRY(hipMemcpy(pl->d_rows_rest.p, rows_rest.data(), rows_rest.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    if (!bit_segs.empty()) HIP_TRY(hipMemcpy(pl->d_bit_segs.p, bit_segs.data(), bit_segs.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    pl->n_rows_rest = (uint32_t)rows_rest.size();
    pl->n_bit_segs = (uint32_t)bit_segs.size();
    if (!table.empty()) HIP_TRY(hipMemcpy(pl->d_table.p, table.data(), table.size() * sizeof(CvNodeRec), hipMemcpyHostToDevice));
    if (!scales.empty()) HIP_TRY(hipMemcpy(pl->d_scales.p, scales.data(), scales.size() * sizeof(CvScaleDev), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl->d_stages.p, stages.data(), stages.size() * sizeof(StageDev), hipMemcpyHostToDevice));
    if (!rows.empty()) HIP_TRY(hipMemcpy(pl->d_rows.p, rows.data(), rows.size() * sizeof(UnitDev), hipMemcpyHostToDevice));
    pl->n_rows = (uint32_t)rows.size();
    return VJ_OK;
}

static int get_cv_plan(vj_env* e, const vj_cascade* c, int W, int H, const vj_cv_params* p, int n_frames, CvPlan** out) {
    uint64_t sf_bits;
    memcpy(&sf_bits, &p->scale_factor, 8);
    const bool small_batch = n_frames <= 4;
    const vj_env::CvPlanKey key(c->uid, W, H, p->min_w, p->min_h, sf_bits, small_batch ? 1 : 0);
    auto it = e->cv_plans.find(key);
    if (it != e->cv_plans.end()) {
        it->second->last_used = ++e->plan_tick;
        *out = it->second.get();
        return VJ_OK;
    }
    if ((int)e->cv_plans.size() >= std::max(1, e->plan_cache_max)) {   // bounded: the least recently used plans go first
        HIP_TRY(hipStreamSynchronize(e->stream));
        while ((int)e->cv_plans.size() >= std::max(1, e->plan_cache_max)) {
            auto lru = e->cv_plans.begin();
            for (auto i = e->cv_plans.begin(); i != e->cv_plans.end(); ++i)
                if (i->second->last_used < lru->second->last_used) lru = i;
            lru->second->release_device();
            e->cv_plans.erase(lru);
        }
    }
    auto pl = std::make_unique<CvPlan>();
    const int rc = build_cv_plan(e, c, W, H, p, small_batch, pl.get());
    if (rc) {
        pl->release_device();
        return rc;
    }
    pl->last_used = ++e->plan_tick;
    *out = pl.get();
    e->cv_plans[key] = std::move(pl);
    return VJ_OK;
}

}  // namespace

extern "C" {

void vj_cv_params_default(vj_cv_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->scale_factor = 1.1;
}

int vj_detect_opencv(vj_env* e, const vj_cascade* c, const vj_image* frames, int n_frames, const vj_cv_params* p,
                     vj_result* out) {
    if (!e || !c || !p || !out || n_frames < 0 || (n_frames > 0 && !frames)) return VJ_ERR_ARG;
    memset(out, 0, sizeof(*out));
    if (n_frames == 0) return VJ_OK;
    if (!(p->scale_factor > 1.0)) {
        set_error("scale_factor must be > 1");
        return VJ_ERR_ARG;
    }
    const int W = frames[0].width, H = frames[0].height;
    if (W <= 0 || H <= 0 || W >= 65535 || H >= 65535) return VJ_ERR_ARG;
    if ((uint64_t)(W + 1) * (uint64_t)(H + 3) >= (1ull << 30)) {   // the limit of vj_detect (check_frames): 32-bit byte offsets
        set_error("image too large");                              // into one frame's sum image
        return VJ_ERR_LIMIT;
    }
    const int CH = image_channels(frames[0]);
    for (int i = 0; i < n_frames; ++i)
        if (!frames[i].data || frames[i].width != W || frames[i].height != H || image_channels(frames[i]) != CH ||
            (CH != 1 && CH != 3 && CH != 4) || frames[i].stride < W * CH) {
            set_error("frame %d: all frames of a batch must be non-null and of equal size and channel count", i);
            return VJ_ERR_ARG;
        }
    HIP_TRY(hipSetDevice(e->device));
    CvPlan* pl;
    int rc = get_cv_plan(e, c, W, H, p, n_frames, &pl);
    if (rc) return rc;
    const std::vector<CvScaleDev>& scales = pl->scales;
    const StageProgram& prog = pl->prog;
    const bool trees = pl->trees, is_tree = pl->is_tree, has_tilted = pl->has_tilted;
    const uint32_t stride = (uint32_t)W + 1u;
    const uint32_t frame_elems = frame_elems_for(W, H);
    DevBuf& d_det = e->d_cv_det;
    DevBuf& d_counts = e->d_cv_counts;
    // counters: stage_entered[VJ_MAX_STAGES] | visited | ... | detection count | pad | 4 x 8 tile ticket counters
    // ... | 64 tree-queue counters (one per tile scale) | chain-pass ticket
    const size_t counts_bytes = 2 * VJ_MAX_STAGES * sizeof(uint64_t) + 16 + 4 * 8 * sizeof(uint32_t) + 64 * sizeof(uint32_t) + 16;
    if ((rc = d_counts.ensure(counts_bytes))) return rc;

    const bool count = (p->flags & VJ_FLAG_COUNTERS) != 0;
    const uint64_t frame_bytes = (uint64_t)frame_elems * 4u;
    int max_frames = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_frames, 0xfffffff0ull / frame_bytes));
    if (e->max_subbatch > 0) max_frames = std::min(max_frames, e->max_subbatch);
    uint32_t det_cap = 1u << 16;
    std::vector<vj_rect> all;
    for (int f0 = 0; f0 < n_frames && pl->n_rows != 0; f0 += max_frames) {
        const int nf = std::min(max_frames, n_frames - f0);
        if ((rc = ensure_image_buffers(e, W, H, nf, true, CH))) return rc;
        const uint8_t* d_gray;
        size_t gray_frame_bytes;
        int gray_stride;
        if ((rc = stage_frames(e, frames + f0, nf, W, H, &d_gray, &gray_frame_bytes, &gray_stride))) return rc;
        HIP_TRY(hipEventRecord(e->lane0.ev[0], e->stream));
        if ((rc = enqueue_integral(e, d_gray, gray_frame_bytes, gray_stride, W, H, nf, CH))) return rc;
        if (has_tilted && (rc = enqueue_tilted(e, d_gray, gray_frame_bytes, gray_stride, W, H, nf, CH))) return rc;
        HIP_TRY(hipEventRecord(e->lane0.ev[1], e->stream));
        bool rows_only = false, done = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if ((rc = d_det.ensure((size_t)det_cap * sizeof(CvDet)))) return rc;
            HIP_TRY(hipMemsetAsync(d_counts.p, 0, counts_bytes, e->stream));
            CvArgs a;
            memset(&a, 0, sizeof(a));
            a.sum = (const uint32_t*)e->d_sum.p;
            a.sqsum = (const uint64_t*)e->d_sqsum.p;
            a.tilted = has_tilted ? (const uint32_t*)e->d_tilted.p : nullptr;
            a.n_order = pl->n_order;
            a.table = (const uint32_t*)pl->d_table.p;
            a.scales = (const CvScaleDev*)pl->d_scales.p;
            a.stages = (const StageDev*)pl->d_stages.p;
            a.rows = (const UnitDev*)pl->d_rows.p;
            a.n_rows = pl->n_rows;
            a.n_frames = (uint32_t)nf;
            a.n_stages = pl->n_stages;
            a.frame_elems = frame_elems;
            a.stride = stride;
            a.sum_h = (uint32_t)H + 1u;
            a.det = (CvDet*)d_det.p;
            a.det_count = (uint32_t*)((unsigned long long*)d_counts.p + 2 * VJ_MAX_STAGES);
            a.det_cap = det_cap;
            a.stage_entered = (unsigned long long*)d_counts.p;
            a.tail_max = (uint32_t)std::max(0, std::min(e->cv_tail_max, (int)CV_TAIL_MAX));
            a.tree2 = pl->tree2 && !is_tree && !has_tilted && e->cv_tree2 ? 1u : 0u;
            a.pairs = e->cv_pairs ? 1u : 0u;
            if (is_tree && pl->chains.n != 0u && !count && e->cv_tree_chains) {   // the rows kernel sweeps the chains too (cv_chain_sweep): a fail list per wave
                a.chains = pl->chains;
                const size_t waves = (size_t)std::max(1, e->n_cu * 4) * CV_WAVES_PER_BLOCK;
                if ((rc = e->d_cv_fail_rows.ensure(waves * CV_QCAP * 16u))) return rc;
                a.fail_scratch = e->d_cv_fail_rows.p;
            }
            HIP_TRY(hipEventRecord(e->lane0.ev[2], e->stream));
            int hrc = 0;
            // (a stage tree's tile path leaves no per-stage counts of the visited windows: counted calls walk the rows)
            const bool tiles = pl->n_tile_scales != 0 && pl->class_first[2] != 0 && !(is_tree && (count || rows_only));
            uint32_t tq_cap = 0;
            if (tiles && is_tree) {
                // Stage tree: the tiles run the linear prefix on every grid window (cv_tile_pass<2>), cv_tree_walk the rest of the
                // tree for the survivors, skip_resolve + cv_tree_emit the sequential walk.  cv_profile_pass keeps the large scales.
                const size_t bits_bytes = (size_t)pl->bits_frame_words * 8u * (size_t)nf;
                if ((rc = e->d_skip_bits.ensure(bits_bytes))) return rc;
                if ((rc = e->d_cv_accept.ensure(bits_bytes))) return rc;
                // room for 1 / 2^shift of the tile windows (a few percent survive the prefix); an overflow halves the shift and
                // runs the call again; a forced capacity (tests) falls back to the rows instead
                // (the shift belongs to the PLAN: one survivor-heavy workload does not make every later call allocate more)
                if (pl->tq_shift < 0) pl->tq_shift = e->cv_tq_shift;
                const bool chain_pass = pl->chains.n != 0u && pl->n_tile_scales <= 64u && e->cv_tree_queue_cap <= 0 && e->cv_tree_chains;
                const uint64_t want = chain_pass ? cv_tq_first((uint32_t)std::min<uint64_t>(pl->tile_windows, 0xffffffffull), pl->n_tile_scales, (uint32_t)nf, (uint32_t)pl->tq_shift) + 4096u
                                                 : ((pl->tile_windows * (uint64_t)nf) >> pl->tq_shift) + 4096u;
                tq_cap = (uint32_t)std::min<uint64_t>(want, 1ull << 28);
                if (e->cv_tree_queue_cap > 0) tq_cap = (uint32_t)e->cv_tree_queue_cap;
                {   // never more than a quarter of what the device has free: beyond that the rows take the call
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)tq_cap * sizeof(CvTreeEntry) > e->d_cv_tq.cap &&
                        (size_t)tq_cap * sizeof(CvTreeEntry) - e->d_cv_tq.cap > free_b / 4u) {
                        rows_only = true;
                        --attempt;
                        continue;
                    }
                }
                if ((rc = e->d_cv_tq.ensure((size_t)tq_cap * sizeof(CvTreeEntry)))) return rc;
                HIP_TRY(hipMemsetAsync(e->d_skip_bits.p, 0, bits_bytes, e->stream));
                HIP_TRY(hipMemsetAsync(e->d_cv_accept.p, 0, bits_bytes, e->stream));
                const bool two = e->concurrent && pl->n_rows_rest != 0;
                hipStream_t sB = two ? e->stream2 : e->stream;
                if (two) {
                    HIP_TRY(hipEventRecord(e->fork_ev, e->stream));
                    HIP_TRY(hipStreamWaitEvent(e->stream2, e->fork_ev, 0));
                }
                if (pl->n_rows_rest != 0) {
                    CvArgs b = a;
                    b.rows = (const UnitDev*)pl->d_rows_rest.p;
                    b.n_rows = pl->n_rows_rest;
                    const int nb = std::max(1, e->n_cu * (two ? pl->row_blocks : 4));
                    b.total_waves = (uint32_t)nb * CV_WAVES_PER_BLOCK;
                    hrc = launch_cv_profile_pass(b, trees, false, true, nb, sB);
                }
                uint32_t* tickets = a.det_count + 4;
                uint32_t* tq_count = tickets + 32;
                CvTileArgs t;
                memset(&t, 0, sizeof(t));
                t.tq_shift = chain_pass ? (uint32_t)pl->tq_shift : 0xffffffffu;   // one sub-queue per scale, or one flat queue (cv_tree_walk)
                t.sum = a.sum;
                t.tilted = a.tilted;
                t.sqsum = a.sqsum;
                t.table = a.table;
                t.scales = a.scales;
                t.stages = a.stages;
                t.n_frames = (uint32_t)nf;
                t.n_stages = pl->tree_prefix;        // the tiles stop after the prefix
                t.frame_elems = frame_elems;
                t.stride = stride;
                t.sum_h = a.sum_h;
                t.bits = (unsigned long long*)e->d_skip_bits.p;
                t.bits_frame_words = pl->bits_frame_words;
                t.repack_mask = ~3ull;
                t.ws_begin = 0xffffffffu;            // no finish: whoever survives the prefix leaves the tile
                t.ws_max = 0;
                t.det = a.det;
                t.det_count = a.det_count;
                t.det_cap = det_cap;
                t.tq = (CvTreeEntry*)e->d_cv_tq.p;
                t.tq_count = tq_count;
                t.tq_cap = tq_cap;
                for (int cls = 0; cls < 2 && !hrc; ++cls) {
                    const uint32_t n_cls = pl->class_first[cls + 1] - pl->class_first[cls];
                    if (!n_cls) continue;
                    CvTileArgs ta = t;
                    ta.tiles = (const UnitDev*)pl->d_tiles.p + pl->class_first[cls];
                    ta.n_tiles = n_cls;
                    ta.lds_bytes = pl->class_lds[cls];
                    ta.ticket = tickets + 8 * cls;
                    const int per_cu = std::max(1, std::min(2, (int)((160u * 1024u - (uint32_t)pl->row_blocks * 20u * 1024u) / ta.lds_bytes)));
                    const int tb = (int)std::min<uint64_t>((uint64_t)n_cls * (uint64_t)nf, (uint64_t)e->n_cu * (uint64_t)per_cu);
                    hrc = launch_cv_tile_pass(ta, 2, false, false, std::max(1, tb), e->stream);
                }
                CvTreeArgs w;
                memset(&w, 0, sizeof(w));
                w.sum = a.sum;
                w.table = a.table;
                w.scales = a.scales;
                w.stages = a.stages;
                w.n_order = pl->n_order;
                w.prefix = pl->tree_prefix;
                w.sum_bytes = (uint32_t)((uint64_t)frame_elems * 4u * (uint64_t)nf);
                w.tq = (const CvTreeEntry*)e->d_cv_tq.p;
                w.tq_count = tq_count;
                w.tq_cap = tq_cap;
                w.reject = (unsigned long long*)e->d_skip_bits.p;
                w.accept = (unsigned long long*)e->d_cv_accept.p;
                w.bits_frame_words = pl->bits_frame_words;
                w.n_frames = (uint32_t)nf;
                w.segs = (const UnitDev*)pl->d_bit_segs.p;
                w.n_segs = pl->n_bit_segs;
                w.det = a.det;
                w.det_count = a.det_count;
                w.det_cap = det_cap;
                if (chain_pass) {
                    // the tree is made of chains: chunks of one scale's survivors, swept like a linear cascade (cv_chain_sweep)
                    const int wb = std::max(1, e->n_cu * std::max(1, std::min(e->cv_tree_chain_blocks, 4)));
                    w.tq_shift = (uint32_t)pl->tq_shift;
                    w.n_scales = (uint32_t)scales.size();
                    w.ticket = tq_count + 64;
                    w.chains = pl->chains;
                    w.total_waves = (uint32_t)wb * 4u;
                    w.chunk = (uint32_t)std::max(64, std::min(e->cv_tree_chunk, (int)CV_TQ_CHUNK));
                    if ((rc = e->d_cv_fail_walk.ensure((size_t)w.total_waves * CV_TQ_CHUNK * sizeof(CvTreeEntry)))) return rc;
                    w.fail_scratch = e->d_cv_fail_walk.p;
                    if (!hrc) hrc = launch_cv_tree_chain_pass(w, wb, e->stream);
                } else if (!hrc) {
                    hrc = launch_cv_tree_walk(w, std::max(1, e->n_cu * 4), e->stream);
                }
                if (!hrc) {
                    CascadeArgs ra;
                    memset(&ra, 0, sizeof(ra));
                    ra.skip_bits = (unsigned long long*)e->d_skip_bits.p;
                    ra.skip_frame_words = pl->bits_frame_words;
                    ra.skip_segs = (const UnitDev*)pl->d_bit_segs.p;
                    ra.n_skip_segs = pl->n_bit_segs;
                    ra.n_frames = (uint32_t)nf;
                    hrc = launch_skip_resolve(ra, std::max(1, e->n_cu * 2), e->stream);
                }
                if (!hrc) hrc = launch_cv_tree_emit(w, std::max(1, e->n_cu * 2), e->stream);
                if (two) {
                    HIP_TRY(hipEventRecord(e->join_ev, e->stream2));
                    HIP_TRY(hipStreamWaitEvent(e->stream, e->join_ev, 0));
                }
            } else if (tiles) {
                // The small scales on LDS tiles (vj_cv_tile.hip), the rest on cv_profile_pass, concurrently on two streams:
                // the row kernel is bound by the texture-address unit, the tile kernel by LDS and VALU.  The row kernel is
                // launched first with one workgroup per CU so that the tile workgroups find their LDS share next to it.
                if ((rc = e->d_skip_bits.ensure((size_t)pl->bits_frame_words * 8u * (size_t)nf))) return rc;
                HIP_TRY(hipMemsetAsync(e->d_skip_bits.p, 0, (size_t)pl->bits_frame_words * 8u * (size_t)nf, e->stream));
                const bool two = e->concurrent && pl->n_rows_rest != 0;
                hipStream_t sB = two ? e->stream2 : e->stream;
                if (two) {
                    HIP_TRY(hipEventRecord(e->fork_ev, e->stream));
                    HIP_TRY(hipStreamWaitEvent(e->stream2, e->fork_ev, 0));
                }
                if (pl->n_rows_rest != 0) {
                    CvArgs b = a;
                    b.rows = (const UnitDev*)pl->d_rows_rest.p;
                    b.n_rows = pl->n_rows_rest;
                    const int nb = std::max(1, e->n_cu * (two ? pl->row_blocks : 4));
                    b.total_waves = (uint32_t)nb * CV_WAVES_PER_BLOCK;
                    hrc = launch_cv_profile_pass(b, trees, count, is_tree, nb, sB);
                }
                CvTileArgs t;
                memset(&t, 0, sizeof(t));
                t.sum = a.sum;
                t.tilted = a.tilted;
                t.sqsum = a.sqsum;
                t.table = a.table;
                t.scales = a.scales;
                t.stages = a.stages;
                t.n_frames = (uint32_t)nf;
                t.n_stages = pl->n_stages;
                t.frame_elems = frame_elems;
                t.stride = stride;
                t.sum_h = a.sum_h;
                t.bits = (unsigned long long*)e->d_skip_bits.p;
                t.bits_frame_words = pl->bits_frame_words;
                t.repack_mask = ~3ull;        // before every stage from 2 on (as the clod profile's tiles)
                t.ws_begin = 3;
                t.ws_max = (uint32_t)e->cv_tile_ws_max;
                t.det = a.det;
                t.det_count = a.det_count;
                t.det_cap = det_cap;
                t.stage_entered = a.stage_entered;
                uint32_t* tickets = a.det_count + 4;
                for (int mode = 0; mode < 2 && !hrc; ++mode) {
                    for (int cls = 0; cls < 2 && !hrc; ++cls) {
                        const uint32_t n_cls = pl->class_first[cls + 1] - pl->class_first[cls];
                        if (!n_cls) continue;
                        CvTileArgs ta = t;
                        ta.tiles = (const UnitDev*)pl->d_tiles.p + pl->class_first[cls];
                        ta.n_tiles = n_cls;
                        ta.lds_bytes = pl->class_lds[cls];
                        ta.ticket = tickets + 8 * (mode * 2 + cls);
                        const int per_cu = std::max(1, std::min(2, (int)((160u * 1024u - (uint32_t)pl->row_blocks * 20u * 1024u) / ta.lds_bytes)));
                        const int tb = (int)std::min<uint64_t>((uint64_t)n_cls * (uint64_t)nf, (uint64_t)e->n_cu * (uint64_t)per_cu);
                        hrc = launch_cv_tile_pass(ta, mode, count, pl->tree2, std::max(1, tb), e->stream);
                    }
                    if (mode == 0 && !hrc) {   // reject bits -> visited bits, one recurrence domain per window row (skip_resolve)
                        CascadeArgs ra;
                        memset(&ra, 0, sizeof(ra));
                        ra.skip_bits = (unsigned long long*)e->d_skip_bits.p;
                        ra.skip_frame_words = pl->bits_frame_words;
                        ra.skip_segs = (const UnitDev*)pl->d_bit_segs.p;
                        ra.n_skip_segs = pl->n_bit_segs;
                        ra.n_frames = (uint32_t)nf;
                        hrc = launch_skip_resolve(ra, std::max(1, e->n_cu * 2), e->stream);
                    }
                }
                if (two) {
                    HIP_TRY(hipEventRecord(e->join_ev, e->stream2));
                    HIP_TRY(hipStreamWaitEvent(e->stream, e->join_ev, 0));
                }
            } else {
                // four workgroups (16 waves) per CU: every wave walks its own window row, and the rows in flight on an XCD
                // should stay inside its 4 MiB L2 (64 x 1080p: 376 / 235 / 179 / 153 / 173 / 194 / 193 ms for 1 / 2 / 3 / 4 / 5 / 6 / 8)
                const int n_blocks = std::max(1, e->n_cu * 4);
                a.total_waves = (uint32_t)n_blocks * CV_WAVES_PER_BLOCK;
                hrc = launch_cv_profile_pass(a, trees, count, is_tree, n_blocks, e->stream);
            }
            if (hrc) {
                set_error("cascade launch failed: %s", hipGetErrorString((hipError_t)hrc));
                return VJ_ERR_HIP;
            }
            HIP_TRY(hipEventRecord(e->lane0.ev[3], e->stream));
            std::vector<unsigned long long> h((counts_bytes + 7) / 8);
            HIP_TRY(hipMemcpyAsync(h.data(), d_counts.p, counts_bytes, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            const uint32_t n_det = (uint32_t)(h[2 * VJ_MAX_STAGES] & 0xffffffffull);
            bool tq_overflow = false;
            if (tq_cap != 0u) {   // more prefix survivors than the tree queue (or one scale's sub-queue) holds?
                const uint32_t* tqc = (const uint32_t*)(h.data() + 2 * VJ_MAX_STAGES) + 4 + 32;
                const bool per_scale = pl->chains.n != 0u && pl->n_tile_scales <= 64u && e->cv_tree_queue_cap <= 0 && e->cv_tree_chains;
                if (!per_scale) {
                    tq_overflow = tqc[0] > tq_cap;
                } else {
                    for (const CvScaleDev& sd : scales)
                        if (sd.tile_th != 0u && (uint64_t)tqc[sd.tq_slot] > cv_tq_cap(sd.end_x, sd.end_y, (uint32_t)nf, (uint32_t)pl->tq_shift)) tq_overflow = true;
                }
            }
            if (tq_overflow) {
                if (e->cv_tree_queue_cap > 0 || pl->tq_shift <= 0) rows_only = true;
                else pl->tq_shift = std::max(0, pl->tq_shift - 2);      // 1/16 -> 1/4 -> every window, for THIS plan
                --attempt;
                continue;
            }
            if (n_det > det_cap) {   // overflow: grow and redo this sub-batch's cascade
                while (det_cap < n_det) det_cap *= 2;
                continue;
            }
            float ms_i = 0, ms_c = 0, ms_t = 0;
            HIP_TRY(hipEventElapsedTime(&ms_i, e->lane0.ev[0], e->lane0.ev[1]));
            HIP_TRY(hipEventElapsedTime(&ms_c, e->lane0.ev[2], e->lane0.ev[3]));
            HIP_TRY(hipEventElapsedTime(&ms_t, e->lane0.ev[0], e->lane0.ev[3]));
            out->timing.integral_ms += ms_i;
            out->timing.cascade_ms += ms_c;
            out->timing.total_ms += ms_t;
            out->timing.n_cascade_launches = 1;
            if (count) {
                for (size_t s = 0; s < pl->n_stages; ++s) out->counters.stage_entered[s] += h[s];
                out->counters.windows += h[VJ_MAX_STAGES];
            }
            std::vector<CvDet> raw(n_det);
            if (n_det) HIP_TRY(hipMemcpy(raw.data(), d_det.p, (size_t)n_det * sizeof(CvDet), hipMemcpyDeviceToHost));
            for (const CvDet& d : raw)
                all.push_back(vj_rect{(int32_t)d.x, (int32_t)d.y, (int32_t)scales[d.slot].win_w, (int32_t)scales[d.slot].win_h,
                                      0.0f, f0 + (int32_t)d.frame, (int32_t)scales[d.slot].scale_idx});
            done = true;
            break;
        }
        if (!done) {   // (cannot happen: the counts of a repeated pass are the counts that sized its buffers)
            set_error("vj_detect_opencv: the detection buffer overflowed twice");
            return VJ_ERR_LIMIT;
        }
    }
    std::sort(all.begin(), all.end(), [](const vj_rect& a, const vj_rect& b) {
        return std::tie(a.frame, a.scale_idx, a.y, a.x) < std::tie(b.frame, b.scale_idx, b.y, b.x);
    });
    out->count = (uint32_t)all.size();
    if (!all.empty()) {
        out->rects = (vj_rect*)malloc(all.size() * sizeof(vj_rect));
        if (!out->rects) return VJ_ERR_NOMEM;
        memcpy(out->rects, all.data(), all.size() * sizeof(vj_rect));
    }
    if (p->min_neighbors != 0 && out->count) {   // groupRectangles(rectList, max(minNeighbors, 1), GROUP_EPS)
        rc = vj_group_rectangles(out->rects, &out->count, (int)std::max<uint32_t>(p->min_neighbors, 1u), 0.2);
        if (rc) return rc;
    }
    if (count) {
        vj_counters& k = out->counters;
        uint64_t rect_evals = 0;
        for (size_t s = 0; s < pl->n_stages; ++s) {
            k.stump_evals += k.stage_entered[s] * prog.n_nodes[s];
            rect_evals += k.stage_entered[s] * prog.n_rects[s];
        }
        k.gather_bytes = 48ull * k.stage_entered[0] + 16ull * rect_evals;
    }
    return VJ_OK;
}

}  // extern "C"
