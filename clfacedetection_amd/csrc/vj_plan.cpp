// Host-side planning: scale enumeration, per-scale window grid and the per-scale
// feature table.  This is the reference's own host logic (it runs on the CPU in the
// reference's OpenCL path too, clod.cpp:1198-1246) restated literally, because its
// mixed round()/lrint()/float/double choices define which windows exist and which
// weights the kernel multiplies by.  Compiled with -ffp-contract=off.
#include "vj_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace vj {

// round() on the f32 product, as `(cl_uint)round(int * float)` does (clod.cpp:387-388,
// 404-407, 554-557): half away from zero, then conversion to unsigned.
static inline uint32_t round_u32(float v) { return (uint32_t)std::round((double)v); }

std::vector<vj_scale_info> plan_scales(const vj_cascade& c, int W, int H, const vj_params& p) {
    std::vector<vj_scale_info> out;
    const float sf = p.scale_factor;
    const int w0 = c.win_w, h0 = c.win_h;
    // Scale count loop (clod.cpp:1198-1204): float * int compared against int.
    int count = 0;
    for (float cs = 1; cs * (float)w0 < (float)(W - 10) && cs * (float)h0 < (float)(H - 10); cs *= sf) {
        if (++count > 4096) break;  // scale_factor <= 1 would never terminate in the reference
    }
    float s = 1;
    for (int k = 0; k < count; ++k, s *= sf) {
        vj_scale_info si;
        memset(&si, 0, sizeof(si));
        si.scale_idx = k;
        si.scale = s;
        // setupScale (clod.cpp:371-415)
        si.step = (float)std::max(2.0, (double)s);
        si.win_w = (int32_t)round_u32((float)w0 * s);
        si.win_h = (int32_t)round_u32((float)h0 * s);
        bool ok = true;
        if (si.win_w < p.min_w || si.win_h < p.min_h) ok = false;
        if (p.max_w != 0 && si.win_w > p.max_w) ok = false;
        if (p.max_h != 0 && si.win_h > p.max_h) ok = false;
        if (si.win_w > W || si.win_h > H) ok = false;
        if (ok) {
            si.equ_x = (int32_t)round_u32(s);
            si.equ_y = si.equ_x;
            si.equ_w = (int32_t)round_u32((float)(w0 - 2) * s);
            si.equ_h = (int32_t)round_u32((float)(h0 - 2) * s);
            si.area = (uint32_t)(si.equ_w * si.equ_h);
            // lrint of an int / float quotient: round-half-even in the default mode
            si.nx = (int32_t)std::lrint((double)((float)(W - si.win_w) / si.step));
            si.ny = (int32_t)std::lrint((double)((float)(H - si.win_h) / si.step));
            if (p.flags & VJ_FLAG_GRID_F64) {
                // the block variant keeps `step` as a double (clod.cpp:862) and divides in f64 (:890-891)
                const double stepd = std::max(2.0, (double)s);
                si.nx = (int32_t)std::lrint((double)(uint32_t)(W - si.win_w) / stepd);
                si.ny = (int32_t)std::lrint((double)(uint32_t)(H - si.win_h) / stepd);
            }
            if (si.nx < 0) si.nx = 0;
            if (si.ny < 0) si.ny = 0;
            si.accepted = 1;
        }
        out.push_back(si);
    }
    return out;
}

// precomputeKernelCascade (clod.cpp:529-578) for every node of the cascade, in flat
// node order, into the 64-byte device record.
int build_node_table(const vj_cascade& c, int width, const vj_scale_info& s, NodeRec* recs) {
    return build_node_table_stride(c, (uint32_t)width + 1u, s, recs, 0u);
}

// Same records with the row stride of some other image layout (the LDS tiles of the
// tile kernel have their own pitch); weights and thresholds do not depend on it.
// deint_half != 0: the rows of that layout are de-interleaved — even columns first, odd
// columns from element deint_half on — so column c sits at (c & 1) * deint_half + (c >> 1);
// the left->right distance of a rectangle then depends on the parities and may be negative
// (dw is a signed 16-bit field).
int build_node_table_stride(const vj_cascade& c, uint32_t stride, const vj_scale_info& s, NodeRec* recs,
                            uint32_t deint_half) {
    auto col = [deint_half](uint32_t cx) { return deint_half ? (cx & 1u) * deint_half + (cx >> 1) : cx; };
    const float cs = s.scale;
    const float area = (float)s.area;
    for (size_t t = 0; t < c.trees.size(); ++t) {
        const vj_tree_desc& td = c.trees[t];
        for (int k = 0; k < td.n_nodes; ++k) {
            const vj_node_desc& nd = c.nodes[td.first_node + k];
            NodeRec& r = recs[td.first_node + k];
            memset(&r, 0, sizeof(r));
            // (nd.tilted is not read: precomputeFeatures takes the rectangles as they are, clod.cpp:448-492; build_plan refuses such
            // a cascade unless the caller asks for the reference's reading with VJ_FLAG_TILTED_AS_UPRIGHT)
            if (nd.rect[0].weight == 0.0f || nd.rect[1].weight == 0.0f) {
                set_error("node %d: rect 0 and rect 1 must both be weighted (clod.cl:60-68 reads both)",
                          td.first_node + k);
                return VJ_ERR_UNSUPPORTED;
            }
            float first_rect_area = 0.0f;
            float sum_rect_area = 0.0f;
            uint32_t dw[3] = {0, 0, 0};
            for (int q = 0; q < 3; ++q) {
                const float ow = nd.rect[q].weight;
                if (ow != 0.0f) {
                    const uint32_t rx = round_u32((float)nd.rect[q].x * cs);
                    const uint32_t ry = round_u32((float)nd.rect[q].y * cs);
                    const uint32_t rw = round_u32((float)nd.rect[q].w * cs);
                    const uint32_t rh = round_u32((float)nd.rect[q].h * cs);
                    const float wgt = ow / area;
                    const uint64_t lt = ((uint64_t)ry * stride + col(rx)) * 4u;
                    const uint64_t dh = (uint64_t)rh * stride * 4u;
                    const int64_t dwb = ((int64_t)col(rx + rw) - (int64_t)col(rx)) * 4;
                    if (lt > 0xffffffffull || dh > 0xffffffffull || dwb > 32767 || dwb < -32768) {
                        set_error("feature offsets exceed the device record range");
                        return VJ_ERR_LIMIT;
                    }
                    r.lt[q] = (uint32_t)lt;
                    r.dh[q] = (uint32_t)dh;
                    dw[q] = (uint32_t)dwb & 0xffffu;   // signed 16-bit
                    r.w[q] = wgt;
                    if (q > 0)
                        sum_rect_area += wgt * (float)rw * (float)rh;
                    else
                        first_rect_area = (float)(rw * rh);
                } else {
                    r.w[q] = 0.0f;
                }
            }
            r.w[0] = -sum_rect_area / first_rect_area;
            r.thr = nd.threshold;
            uint32_t flags = 0;
            auto leaf_or_node = [&](int v, uint32_t flag, uint32_t* dst) {
                if (v > 0) {
                    flags |= flag;
                    *dst = (uint32_t)v;
                } else {
                    const float a = c.alpha[td.first_alpha - v];
                    memcpy(dst, &a, 4);
                }
            };
            leaf_or_node(nd.left, NODE_LEFT_IS_NODE, &r.left);
            leaf_or_node(nd.right, NODE_RIGHT_IS_NODE, &r.right);
            if (k == td.n_nodes - 1) flags |= NODE_TREE_LAST;
            r.dw01 = dw[0] | (dw[1] << 16);
            r.dw2_flags = dw[2] | (flags << 16);
        }
    }
    return VJ_OK;
}

}  // namespace vj

using namespace vj;

extern "C" {

int vj_plan_scales(const vj_cascade* c, int width, int height, const vj_params* p, vj_scale_info* out, int cap,
                   int* n) {
    if (!c || !p || !n || width <= 0 || height <= 0 || (cap > 0 && !out)) return VJ_ERR_ARG;
    if (!(p->scale_factor > 1.0f)) {
        set_error("scale_factor must be > 1");
        return VJ_ERR_ARG;
    }
    std::vector<vj_scale_info> v = plan_scales(*c, width, height, *p);
    *n = (int)v.size();
    for (int i = 0; i < *n && i < cap; ++i) out[i] = v[i];
    return VJ_OK;
}

int vj_plan_feature_table(const vj_cascade* c, int width, const vj_scale_info* s, uint32_t* offsets, float* weights) {
    if (!c || !s || !offsets || !weights || width <= 0 || !s->accepted) return VJ_ERR_ARG;
    std::vector<NodeRec> recs(c->nodes.size());
    int rc = build_node_table(*c, width, *s, recs.data());
    if (rc) return rc;
    // Expand the compact record back into the reference's KernelOptimizedRect fields
    // (clod.cpp:45-51): four element offsets per rect.
    for (size_t i = 0; i < recs.size(); ++i) {
        const NodeRec& r = recs[i];
        const uint32_t dw[3] = {r.dw01 & 0xffffu, r.dw01 >> 16, r.dw2_flags & 0xffffu};
        for (int q = 0; q < 3; ++q) {
            uint32_t* o = offsets + i * 12 + q * 4;
            const bool present = q < 2 || r.w[2] != 0.0f;
            o[0] = present ? r.lt[q] / 4u : 0u;
            o[1] = present ? (r.lt[q] + dw[q]) / 4u : 0u;
            o[2] = present ? (r.lt[q] + r.dh[q]) / 4u : 0u;
            o[3] = present ? (r.lt[q] + r.dh[q] + dw[q]) / 4u : 0u;
            weights[i * 3 + q] = r.w[q];
        }
    }
    return VJ_OK;
}

int vj_shard_frames(int n_frames, int n_ranks, int rank, int* first, int* count) {
    if (n_frames < 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !first || !count) return VJ_ERR_ARG;
    const int base = n_frames / n_ranks, extra = n_frames % n_ranks;
    *first = rank * base + std::min(rank, extra);
    *count = base + (rank < extra ? 1 : 0);
    return VJ_OK;
}

int vj_shard_scales(const vj_cascade* c, int width, int height, const vj_params* p, int n_ranks, int rank, uint64_t scale_mask[2]) {
    if (!c || !p || !scale_mask || width <= 0 || height <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return VJ_ERR_ARG;
    if (!(p->scale_factor > 1.0f)) return VJ_ERR_ARG;
    const std::vector<vj_scale_info> sc = plan_scales(*c, width, height, *p);
    if (sc.size() > 127) {
        set_error("more than 127 scales cannot be expressed as a scale mask");
        return VJ_ERR_LIMIT;
    }
    // Estimated cost of a scale = windows x a per-window weight (1/16 units, integers only so that every host computes the same
    // shares).  Scales whose window is at most 72 px run on LDS tiles (weight 16 at s = 1, rising with the staged footprint per
    // window); larger ones are global gathers, about six times as expensive per window and rising with the window (measured per
    // scale: profiles/r03_notes.md #7).  The two kinds run as two overlapping chains on a device, so they are dealt separately:
    // every rank gets its part of BOTH — tiles to the rank with the least tile cost, gathers to the rank with the least
    // (gather cost + half its tile cost): equal gather shares where the tile shares are equal, more gathers where a rank got fewer tiles.
    const uint64_t base = (uint64_t)std::max(c->win_w, c->win_h);
    std::vector<uint64_t> w(sc.size());
    std::vector<int> gather(sc.size());
    std::vector<size_t> order(sc.size());
    for (size_t k = 0; k < sc.size(); ++k) {
        const uint64_t side = (uint64_t)std::max(sc[k].win_w, sc[k].win_h);
        const uint64_t q = std::max<uint64_t>(16u, side * 16u / std::max<uint64_t>(base, 1u));   // scale factor in 1/16
        gather[k] = side > 72u;
        const uint64_t weight = gather[k] ? 96u + 13u * (std::max<uint64_t>(q, 61u) - 61u) / 16u : 16u + 10u * (q - 16u) / 16u;
        w[k] = sc[k].accepted ? (uint64_t)sc[k].nx * (uint64_t)sc[k].ny * weight : 0;
        order[k] = k;
    }
    // longest processing time first; equal costs keep their scale order
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return w[a] > w[b]; });
    std::vector<uint64_t> tile((size_t)n_ranks, 0), gath((size_t)n_ranks, 0);
    auto est = [](uint64_t t, uint64_t g) { return g + t / 2u; };
    scale_mask[0] = scale_mask[1] = 0;
    for (int cls = 0; cls < 2; ++cls)
        for (size_t k : order) {
            if (gather[k] != cls) continue;
            size_t r = 0;
            for (size_t i = 1; i < (size_t)n_ranks; ++i) {
                const bool less = cls == 0 ? tile[i] < tile[r] : est(tile[i], gath[i] + w[k]) < est(tile[r], gath[r] + w[k]);
                if (less) r = i;
            }
            (cls == 0 ? tile[r] : gath[r]) += w[k];
            if ((int)r == rank) scale_mask[k >> 6] |= 1ull << (k & 63);
        }
    // an all-zero mask means "every scale" to vj_detect: a rank without a share says so explicitly
    if ((scale_mask[0] | scale_mask[1]) == 0) scale_mask[1] = VJ_SCALE_MASK_NONE;
    return VJ_OK;
}

int vj_count_windows(const vj_cascade* c, int width, int height, const vj_params* p, uint64_t* out) {
    if (!c || !p || !out || width <= 0 || height <= 0) return VJ_ERR_ARG;
    if (!(p->scale_factor > 1.0f)) return VJ_ERR_ARG;
    uint64_t total = 0;
    for (const vj_scale_info& s : plan_scales(*c, width, height, *p))
        if (s.accepted) total += (uint64_t)s.nx * (uint64_t)s.ny;
    *out = total;
    return VJ_OK;
}

}  // extern "C"
