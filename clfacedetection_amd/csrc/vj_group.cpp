// Grouping of raw candidates (min_neighbors != 0) — SURVEY.md §8(f) row 1.
//
// Host code in the reference too (filterResult, clod.cpp:182-357, called at clod.cpp:1325-1326
// with MAX(min_neighbors, 1) and EPS 0.2).  The clod port carries porting bugs (uninitialised
// accumulators, MAX for MIN, width+width typos — SURVEY.md §2.2-5) and is never executed by the
// demo, so this follows the original it was ported from, cv::groupRectangles as kept in
// tempcv.cpp:130-243: partition by SimilarRects(eps), average every class, drop classes with
// <= groupThreshold members and small rectangles inside larger, better supported ones.
// cv::partition (OpenCV 2.4.2 core, not in the reference tree) labels connected components in
// order of first appearance; any components algorithm with that labelling is equivalent.
#include "vj_internal.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>

namespace vj {

struct IRect { int x, y, w, h; };

// ASimilarRects (tempcv.cpp:130-143)
static inline bool similar(const IRect& r1, const IRect& r2, double eps) {
    const double delta = eps * (std::min(r1.w, r2.w) + std::min(r1.h, r2.h)) * 0.5;
    return std::abs(r1.x - r2.x) <= delta && std::abs(r1.y - r2.y) <= delta &&
           std::abs(r1.x + r1.w - r2.x - r2.w) <= delta && std::abs(r1.y + r1.h - r2.y - r2.h) <= delta;
}

// Connected components of the "similar" graph, labelled in order of first appearance.
static int label_components(const std::vector<IRect>& v, double eps, std::vector<int>* labels) {
    const int n = (int)v.size();
    labels->assign(n, -1);
    int ncls = 0;
    std::vector<int> stack;
    for (int i = 0; i < n; ++i) {
        if ((*labels)[i] != -1) continue;
        (*labels)[i] = ncls;
        stack.assign(1, i);
        while (!stack.empty()) {
            const int a = stack.back();
            stack.pop_back();
            for (int b = 0; b < n; ++b)
                // the predicate is symmetric in value but evaluate it as partition() does: (vec[i], vec[j])
                if ((*labels)[b] == -1 && (similar(v[a], v[b], eps) || similar(v[b], v[a], eps))) {
                    (*labels)[b] = ncls;
                    stack.push_back(b);
                }
        }
        ++ncls;
    }
    return ncls;
}

// AgroupRectangles(rectList, groupThreshold, eps, weights, 0) (tempcv.cpp:145-243).
void group_rectangles(std::vector<IRect>* rects, int group_threshold, double eps, std::vector<int>* weights) {
    weights->clear();
    if (group_threshold <= 0 || rects->empty()) {
        weights->assign(rects->size(), 1);
        return;
    }
    std::vector<int> labels;
    const int ncls = label_components(*rects, eps, &labels);
    std::vector<IRect> rr(ncls, IRect{0, 0, 0, 0});
    std::vector<int> rw(ncls, 0);
    for (size_t i = 0; i < labels.size(); ++i) {
        const int c = labels[i];
        // int accumulators as in the original (tempcv.cpp:167-172); written as unsigned adds so that a (purely
        // theoretical: > 2^31 of summed coordinates) overflow wraps instead of being undefined
        auto acc = [](int a, int b) { return (int)((unsigned)a + (unsigned)b); };
        rr[c].x = acc(rr[c].x, (*rects)[i].x);
        rr[c].y = acc(rr[c].y, (*rects)[i].y);
        rr[c].w = acc(rr[c].w, (*rects)[i].w);
        rr[c].h = acc(rr[c].h, (*rects)[i].h);
        rw[c]++;
    }
    auto sat = [](float v) { return v > (float)INT_MAX ? INT_MAX : (int)v; };
    for (int i = 0; i < ncls; ++i) {
        const float s = 1.f / rw[i];
        rr[i] = IRect{sat(rr[i].x * s), sat(rr[i].y * s), sat(rr[i].w * s), sat(rr[i].h * s)};
    }
    std::vector<IRect> out;
    for (int i = 0; i < ncls; ++i) {
        const IRect r1 = rr[i];
        const int n1 = rw[i];
        if (n1 <= group_threshold) continue;
        int j;
        for (j = 0; j < ncls; ++j) {  // filter out small rectangles inside large rectangles
            const int n2 = rw[j];
            if (j == i || n2 <= group_threshold) continue;
            const IRect r2 = rr[j];
            const int dx = r2.w * eps > INT_MAX ? INT_MAX : (int)(r2.w * eps);
            const int dy = r2.h * eps > INT_MAX ? INT_MAX : (int)(r2.h * eps);
            typedef long long ll;
            if (r1.x >= (ll)r2.x - dx && r1.y >= (ll)r2.y - dy && (ll)r1.x + r1.w <= (ll)r2.x + r2.w + dx &&
                (ll)r1.y + r1.h <= (ll)r2.y + r2.h + dy && (n2 > std::max(3, n1) || n1 < 3))
                break;
        }
        if (j == ncls) {
            out.push_back(r1);
            weights->push_back(n1);
        }
    }
    rects->swap(out);
}

}  // namespace vj

extern "C" int vj_group_rectangles(vj_rect* rects, uint32_t* count, int group_threshold, double eps) {
    if (!count || (*count && !rects) || !(eps >= 0.0)) return VJ_ERR_ARG;
    for (uint32_t k = 0; k < *count; ++k) {   // image coordinates: keeps every sum and difference below far inside int
        const vj_rect& r = rects[k];
        const int lim = 1 << 20;
        if (r.x < -lim || r.x > lim || r.y < -lim || r.y > lim || r.w < 0 || r.w > lim || r.h < 0 || r.h > lim) {
            vj::set_error("rectangle %u is not an image rectangle", k);
            return VJ_ERR_ARG;
        }
    }
    // groups are formed per frame; input order inside a frame is kept (labels depend on it)
    std::vector<vj_rect> out;
    uint32_t i = 0;
    const uint32_t n = *count;
    while (i < n) {
        uint32_t j = i;
        while (j < n && rects[j].frame == rects[i].frame) ++j;
        std::vector<vj::IRect> v;
        for (uint32_t k = i; k < j; ++k) v.push_back(vj::IRect{rects[k].x, rects[k].y, rects[k].w, rects[k].h});
        std::vector<int> w;
        vj::group_rectangles(&v, group_threshold, eps, &w);
        for (size_t k = 0; k < v.size(); ++k)
            out.push_back(vj_rect{v[k].x, v[k].y, v[k].w, v[k].h, (float)w[k], rects[i].frame, -1});
        i = j;
    }
    for (size_t k = 0; k < out.size(); ++k) rects[k] = out[k];
    *count = (uint32_t)out.size();
    return VJ_OK;
}
