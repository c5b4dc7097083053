/*
 * vj_oracle.c — CPU restatement of the reference's clif/clod detect path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under clfacedetection_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker / reported CPU baseline.
 *
 * PARITY PINNING: the reference (GabrieleCocco/CLFaceDetection) ships no tests, no
 * golden vectors and no fixtures, and cannot be built in this image (it needs the
 * OpenCV 2.4.2 and CLUtil headers, which are absent; writing stand-ins for them is
 * not allowed).  This restatement is therefore pinned only by (1) the known-answer
 * constants in the reference's cascade XMLs and (2) the reference-run figures that
 * the survey recorded in SURVEY.md §6/§8 (scale counts, candidate-window counts,
 * per-stage survivor counts, stump-evaluation totals, raw detection counts) — see
 * tests/test_oracle_pins.py.  For tree cascades (frontalface_alt2, _alt_tree) the
 * reference's clod path itself is wrong (SURVEY.md §2.2-3,4): parity there is
 * UNPINNED beyond this file and its independent numpy twin (oracle/np_oracle.py,
 * tests/test_np_twin.py).
 *
 * Every function cites the reference lines it follows (paths relative to
 * CLFaceDetection/).  All float arithmetic is IEEE binary32 in the written order;
 * build with -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ types */
typedef struct oc_cascade {
    int32_t win_w, win_h;
    int32_t n_stages, n_trees, n_nodes, n_alpha;
    /* per stage — CvHaarStageClassifier (tempcv.hpp:95-105) */
    const int32_t* stage_first_tree;
    const int32_t* stage_n_trees;
    const float*   stage_threshold;
    const int32_t* stage_parent;
    const int32_t* stage_next;
    const int32_t* stage_child;
    /* per tree — CvHaarClassifier (tempcv.hpp:81-93) */
    const int32_t* tree_first_node;
    const int32_t* tree_n_nodes;
    const int32_t* tree_first_alpha;
    /* per node */
    const int32_t* node_rect;      /* n_nodes * 3 * 4 : x y w h                 */
    const float*   node_weight;    /* n_nodes * 3 (0 = rect absent)             */
    const float*   node_threshold;
    const int32_t* node_left;      /* >0 node index, <=0 -> alpha[-v]           */
    const int32_t* node_right;
    const float*   alpha;
    const int32_t* node_tilted;    /* n_nodes, != 0: <tilted>1 (NULL = none); only the OpenCV-like path reads it */
} oc_cascade;

typedef struct oc_scale {
    int32_t scale_idx;
    float   scale, step;
    int32_t win_w, win_h;
    int32_t equ_x, equ_y, equ_w, equ_h;
    uint32_t area;
    int32_t nx, ny;
    int32_t accepted;
} oc_scale;

typedef struct oc_rect { int32_t x, y, w, h, scale_idx; } oc_rect;

typedef struct oc_stats {
    uint64_t windows;
    uint64_t stump_evals;
    uint64_t rect_evals;           /* sum of rect count over evaluated nodes    */
    uint64_t stage_entered[64];
} oc_stats;

/* ------------------------------------------------- a1: integral (cvIntegral) */
/* clifIntegral / clifGrayscaleIntegral host branch (clif.cpp:280-285, 326-335) call
 * cvIntegral(gray, sum CV_32SC1, sqsum CV_64FC1), both (h+1)x(w+1), first row and
 * column zero.  cvIntegral is exact integer arithmetic (OpenCV 2.4.2, not in the
 * reference tree); sum wraps in 32 bits, sqsum is exact in double below 2^53.      */
void oc_integral(const uint8_t* gray, int w, int h, int stride, int32_t* sum, double* sqsum) {
    const int sw = w + 1;
    for (int x = 0; x <= w; ++x) { sum[x] = 0; sqsum[x] = 0.0; }
    for (int y = 0; y < h; ++y) {
        const uint8_t* src = gray + (size_t)y * stride;
        const int32_t* up = sum + (size_t)y * sw;
        const double* upq = sqsum + (size_t)y * sw;
        int32_t* row = sum + (size_t)(y + 1) * sw;
        double* rowq = sqsum + (size_t)(y + 1) * sw;
        uint32_t s = 0;
        uint64_t q = 0;
        row[0] = 0;
        rowq[0] = 0.0;
        for (int x = 0; x < w; ++x) {
            const uint32_t v = src[x];
            s += v;
            q += (uint64_t)v * v;
            row[x + 1] = (int32_t)((uint32_t)up[x + 1] + s);
            rowq[x + 1] = upq[x + 1] + (double)q;
        }
    }
}

/* ------------------------------------------------------- a2: scale counting */
/* clod.cpp:1198-1204 (same loop at :841-847, :1366-1372). */
int oc_scale_count(int win_w, int win_h, int W, int H, float scale_factor) {
    int scale_count = 0;
    for (float current_scale = 1;
         current_scale * win_w < W - 10 && current_scale * win_h < H - 10;
         current_scale *= scale_factor) {
        scale_count++;
        if (scale_count > 4096) break;
    }
    return scale_count;
}

/* ----------------------------------------------------------- a3: setupScale */
/* clod.cpp:371-415.  Returns 0 when the scale is accepted, -1 otherwise. */
int oc_setup_scale(float current_scale, int W, int H, int win_w, int win_h,
                   int min_w, int min_h, int max_w, int max_h, oc_scale* out) {
    memset(out, 0, sizeof(*out));
    out->scale = current_scale;
    /* :384  *step = MAX(2.0, (float)current_scale);  (double max, stored to float) */
    double stepd = 2.0 > (double)current_scale ? 2.0 : (double)current_scale;
    out->step = (float)stepd;
    /* :387-388 */
    out->win_w = (int32_t)(uint32_t)round(win_w * current_scale);
    out->win_h = (int32_t)(uint32_t)round(win_h * current_scale);
    /* :391-401 */
    if (out->win_w < min_w || out->win_h < min_h) return -1;
    if (max_w != 0 && out->win_w > max_w) return -1;
    if (max_h != 0 && out->win_h > max_h) return -1;
    if (out->win_w > W || out->win_h > H) return -1;
    /* :404-408 */
    out->equ_x = (int32_t)(uint32_t)round(current_scale);
    out->equ_y = out->equ_x;
    out->equ_w = (int32_t)(uint32_t)round((win_w - 2) * current_scale);
    out->equ_h = (int32_t)(uint32_t)round((win_h - 2) * current_scale);
    out->area = (uint32_t)(out->equ_w * out->equ_h);
    /* :411-412 */
    out->nx = (int)lrint((W - out->win_w) / out->step);
    out->ny = (int)lrint((H - out->win_h) / out->step);
    out->accepted = 1;
    return 0;
}

/* ------------------------------------------------- a5: per-scale feature table */
/* KernelOptimizedRect / KernelClassifier (clod.cpp:45-57) per NODE. */
typedef struct k_rect { uint32_t lt, rt, lb, rb; float weight; } k_rect;
typedef struct k_node { k_rect rect[3]; float threshold; } k_node;

/* precomputeKernelCascade (clod.cpp:529-578), applied to every node's feature. */
static void precompute_nodes(const oc_cascade* c, float current_scale, uint32_t area,
                             uint32_t ii_width, k_node* out) {
    for (int n = 0; n < c->n_nodes; ++n) {
        float first_rect_area = 0;
        float sum_rect_area = 0;
        for (int r = 0; r < 3; ++r) {
            float original_weight = c->node_weight[n * 3 + r];
            if (original_weight != 0) {
                const int32_t* orc = c->node_rect + (n * 3 + r) * 4;
                uint32_t rect_x = (uint32_t)round(orc[0] * current_scale);
                uint32_t rect_y = (uint32_t)round(orc[1] * current_scale);
                uint32_t rect_width = (uint32_t)round(orc[2] * current_scale);
                uint32_t rect_height = (uint32_t)round(orc[3] * current_scale);
                float rect_weight = original_weight / (float)area;
                out[n].rect[r].lt = ii_width * rect_y + rect_x;
                out[n].rect[r].rt = ii_width * rect_y + rect_x + rect_width;
                out[n].rect[r].lb = ii_width * (rect_y + rect_height) + rect_x;
                out[n].rect[r].rb = ii_width * (rect_y + rect_height) + rect_x + rect_width;
                out[n].rect[r].weight = rect_weight;
                if (r > 0)
                    sum_rect_area += rect_weight * rect_width * rect_height;
                else
                    first_rect_area = rect_width * rect_height;
            } else {
                out[n].rect[r].lt = out[n].rect[r].rt = out[n].rect[r].lb = out[n].rect[r].rb = 0;
                out[n].rect[r].weight = 0;
            }
        }
        out[n].rect[0].weight = (-sum_rect_area / first_rect_area);
        out[n].threshold = c->node_threshold[n];
    }
}

/* Exported for the table tests: offsets[n*12 + r*4 + {lt,rt,lb,rb}], weights[n*3+r]. */
void oc_feature_table(const oc_cascade* c, float current_scale, uint32_t area, int W,
                      uint32_t* offsets, float* weights) {
    k_node* kn = (k_node*)malloc(sizeof(k_node) * (size_t)c->n_nodes);
    precompute_nodes(c, current_scale, area, (uint32_t)W + 1u, kn);
    for (int n = 0; n < c->n_nodes; ++n)
        for (int r = 0; r < 3; ++r) {
            offsets[n * 12 + r * 4 + 0] = kn[n].rect[r].lt;
            offsets[n * 12 + r * 4 + 1] = kn[n].rect[r].rt;
            offsets[n * 12 + r * 4 + 2] = kn[n].rect[r].lb;
            offsets[n * 12 + r * 4 + 3] = kn[n].rect[r].rb;
            weights[n * 3 + r] = kn[n].rect[r].weight;
        }
    free(kn);
}

/* --------------------------------------------------------- a4: the window list */
typedef struct k_window { uint32_t x, y, offset; float variance; } k_window; /* clod.cpp:33-38 */

/* computeVariance (clod.cpp:418-446).  signed_mean != 0 reads the pixel sum through
 * int* exactly as :426 does; 0 reads it unsigned.               */
static float compute_variance(const int32_t* sum, const double* sqsum, uint32_t ii_width,
                              const oc_scale* sc, uint32_t px, uint32_t py, int signed_mean) {
    const size_t x0 = (size_t)px + sc->equ_x, y0 = (size_t)py + sc->equ_y;
    const size_t w = sc->equ_w, h = sc->equ_h;
    const size_t a = ii_width * y0 + x0, b = ii_width * y0 + x0 + w;
    const size_t c = ii_width * (y0 + h) + x0, d = ii_width * (y0 + h) + x0 + w;
    float mean;
    if (signed_mean) {
        int32_t s = (int32_t)((uint32_t)sum[a] - (uint32_t)sum[b] - (uint32_t)sum[c] + (uint32_t)sum[d]);
        mean = (float)s / (float)sc->area;
    } else {
        uint32_t s = (uint32_t)sum[a] - (uint32_t)sum[b] - (uint32_t)sum[c] + (uint32_t)sum[d];
        mean = (float)s / (float)sc->area;
    }
    /* matss: each corner cast to unsigned long, then summed (clod.cpp:20-21, 432) */
    unsigned long q = (unsigned long)sqsum[a] - (unsigned long)sqsum[b] - (unsigned long)sqsum[c] +
                      (unsigned long)sqsum[d];
    float variance = (float)q;
    variance = (variance / (float)sc->area) - (mean * mean);
    if (variance >= 0)
        variance = sqrtf(variance);
    else
        variance = 1;
    return variance;
}

/* precomputeWindows (clod.cpp:495-527): row-major, x = lrint(ix * step). */
static uint32_t precompute_windows(const int32_t* sum, const double* sqsum, uint32_t ii_width,
                                   const oc_scale* sc, int signed_mean, k_window* out) {
    uint32_t n = 0;
    for (int y_index = 0; y_index < sc->ny; y_index++) {
        for (int x_index = 0; x_index < sc->nx; x_index++) {
            uint32_t px = (uint32_t)lrint(x_index * sc->step);
            uint32_t py = (uint32_t)lrint(y_index * sc->step);
            out[n].x = px;
            out[n].y = py;
            out[n].variance = compute_variance(sum, sqsum, ii_width, sc, px, py, signed_mean);
            out[n].offset = ii_width * py + px;
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------- a6: runStage */
static inline float rect_term(const uint32_t* ii, uint32_t o, const k_rect* r) {
    /* u32 wrap-around on the four corners, then one cast, then one multiply (clod.cl:60-63) */
    return (float)(ii[o + r->lt] - ii[o + r->rt] - ii[o + r->lb] + ii[o + r->rb]) * r->weight;
}

static inline float node_sum(const uint32_t* ii, uint32_t o, const k_node* k) {
    float rect_sum = 0;
    rect_sum += rect_term(ii, o, &k->rect[0]);
    rect_sum += rect_term(ii, o, &k->rect[1]);
    if (k->rect[2].weight != 0) rect_sum += rect_term(ii, o, &k->rect[2]);
    return rect_sum;
}

/* One stage on one window.  Stumps: clod.cl:49-82 (alpha[rect_sum >= norm_threshold]).
 * Multi-node trees: icvEvalHidHaarClassifier control flow (tempcv.cpp:771-792) on the
 * clod f32 arithmetic.                                           */
static float stage_sum_for(const oc_cascade* c, const k_node* kn, const uint32_t* ii,
                           const k_window* w, int stage, oc_stats* st) {
    float stage_sum = 0;
    const int t0 = c->stage_first_tree[stage], t1 = t0 + c->stage_n_trees[stage];
    for (int t = t0; t < t1; ++t) {
        const int n0 = c->tree_first_node[t];
        const float* alpha = c->alpha + c->tree_first_alpha[t];
        if (c->tree_n_nodes[t] == 1) {
            const k_node* k = kn + n0;
            float norm_threshold = k->threshold * w->variance;
            float rect_sum = node_sum(ii, w->offset, k);
            stage_sum += alpha[rect_sum >= norm_threshold];
            st->stump_evals++;
            st->rect_evals += (k->rect[2].weight != 0) ? 3 : 2;
        } else {
            int idx = 0;
            do {
                const k_node* k = kn + n0 + idx;
                float tt = k->threshold * w->variance;
                float sum = node_sum(ii, w->offset, k);
                st->stump_evals++;
                st->rect_evals += (k->rect[2].weight != 0) ? 3 : 2;
                idx = sum < tt ? c->node_left[n0 + idx] : c->node_right[n0 + idx];
            } while (idx > 0);
            stage_sum += alpha[-idx];
        }
    }
    return stage_sum;
}

/* Detect on one frame.
 *  mode 0: per-stage list compaction, the OpenCL driver's structure (clod.cpp:1212-1322,
 *          runStage clod.cl:32-93) with the intended ping-pong / zeroed counter
 *          (SURVEY.md §2.2-1,2).  Linear cascades only.
 *  mode 1: per-window walk of the stage tree (tempcv.cpp:834-861): pass -> child,
 *          fail -> climb parents until a next sibling exists, else reject; falling
 *          off the end accepts.  Works for linear cascades too (child = next stage).
 *  mode 2: the CPU per-stage-list variant (CLOD_PER_STAGE_ITERATIONS, clod.cpp:1434-1482 with
 *          runSubwindow :681-734): mode 0 plus `subwindow_incr = 2` after a stage-0 reject (:729-732) — the
 *          skip runs over the FLATTENED row-major window list, so it crosses row ends.  Linear cascades.
 *  mode 3: the plain CPU per-window variant (clod.cpp:1409-1432 with runCascade :736-787): positions are
 *          (cl_uint)round(index * step) — half away from zero, not precomputeWindows' lrint (:1416 vs :514)
 *          — and `x_incr = exit_stage != 0 ? 1 : 2` (:1430) skips the next window of the ROW after a stage-0
 *          reject; every row starts at its first window.  Linear cascades.
 *  mode 4: the block variant's plain loop (clodDetectObjectsBlock without CLOD_PER_STAGE_ITERATIONS,
 *          clod.cpp:936-1030): `step` stays a double (:862), so the grid ends are lrint of an f64 quotient
 *          (:890-891) and the positions lrint of an f64 product (:941-942, half to even — not mode 3's round());
 *          x_incr skip inside a row (:1009); the pixel sum is read through cl_uint* (:836, :944), so
 *          signed_mean does not apply.
 *  mode 5: the block variant's per-stage lists (clod.cpp:1031-1155): the same f64 grid with positions
 *          round(index * step) (:1034, half away from zero) and the skip over the flattened list (:1131-1134).
 *          In modes 2-5 stage_entered[0] counts the windows actually visited (P2 of SURVEY.md §8a-8).
 * Returns the number of rects written (at most cap; the true count is in *n_total). */
int oc_detect(const oc_cascade* c, const uint8_t* gray, int W, int H, int stride,
              int min_w, int min_h, int max_w, int max_h, float scale_factor,
              int signed_mean, int mode,
              oc_rect* out, int cap, int* n_total, oc_stats* st) {
    const uint32_t iw = (uint32_t)W + 1u;
    /* +2 zero rows of slack so the one-column feature overshoot at the bottom-right
     * corner reads defined zeros instead of past the allocation. */
    int32_t* sum = (int32_t*)calloc((size_t)iw * (H + 3), sizeof(int32_t));
    double* sqsum = (double*)calloc((size_t)iw * (H + 3), sizeof(double));
    k_node* kn = (k_node*)malloc(sizeof(k_node) * (size_t)c->n_nodes);
    memset(st, 0, sizeof(*st));
    oc_integral(gray, W, H, stride, sum, sqsum);
    const uint32_t* ii = (const uint32_t*)sum;

    int found = 0;
    const int scale_count = oc_scale_count(c->win_w, c->win_h, W, H, scale_factor);
    float current_scale = 1;
    for (int scale_index = 0; scale_index < scale_count; scale_index++, current_scale *= scale_factor) {
        oc_scale sc;
        if (oc_setup_scale(current_scale, W, H, c->win_w, c->win_h, min_w, min_h, max_w, max_h, &sc) != 0)
            continue;
        sc.scale_idx = scale_index;
        precompute_nodes(c, current_scale, sc.area, iw, kn);
        const double stepd = 2.0 > (double)current_scale ? 2.0 : (double)current_scale;   /* clod.cpp:862 */
        if (mode == 4 || mode == 5) {
            /* clod.cpp:890-891: (int - cl_uint) is unsigned, converted to double and divided by the double step */
            sc.nx = (int)lrint((unsigned)(W - sc.win_w) / stepd);
            sc.ny = (int)lrint((unsigned)(H - sc.win_h) / stepd);
            signed_mean = 0;
        }
        const size_t nwin = (size_t)(sc.nx > 0 ? sc.nx : 0) * (size_t)(sc.ny > 0 ? sc.ny : 0);
        if (nwin == 0) continue;
        k_window* a = (k_window*)malloc(sizeof(k_window) * nwin);
        if (mode == 3 || mode == 4) {
            /* clod.cpp:1409-1432: x_incr is declared before the row loop, but every row's for-statement starts
             * again at start_point.x, so only the increments inside a row use it */
            st->windows += nwin;
            unsigned x_incr = 1;
            for (int y_index = 0; y_index < sc.ny; y_index++) {
                for (int x_index = 0; x_index < sc.nx; x_index += (int)x_incr) {
                    k_window w;
                    if (mode == 3) {
                        w.x = (uint32_t)round(x_index * sc.step);   /* int * float -> float, round(): half away */
                        w.y = (uint32_t)round(y_index * sc.step);
                    } else {
                        w.x = (uint32_t)lrint(x_index * stepd);     /* :941-942, int * double, half to even */
                        w.y = (uint32_t)lrint(y_index * stepd);
                    }
                    w.variance = compute_variance(sum, sqsum, iw, &sc, w.x, w.y, signed_mean);
                    w.offset = iw * w.y + w.x;
                    int exit_stage = 1;   /* runCascade, clod.cpp:736-787 */
                    for (int stage = 0; stage < c->n_stages; ++stage) {
                        st->stage_entered[stage]++;
                        float stage_sum = stage_sum_for(c, kn, ii, &w, stage, st);
                        if (stage_sum < c->stage_threshold[stage]) { exit_stage = -stage; break; }
                    }
                    if (exit_stage > 0) {
                        if (found < cap) {
                            out[found].x = (int32_t)w.x; out[found].y = (int32_t)w.y;
                            out[found].w = sc.win_w; out[found].h = sc.win_h; out[found].scale_idx = scale_index;
                        }
                        found++;
                    }
                    x_incr = exit_stage != 0 ? 1 : 2;
                }
            }
            free(a);
            continue;
        }
        uint32_t n_in;
        if (mode == 5) {
            /* clod.cpp:1031-1067: the block variant fills its list itself, positions round(index * step) in f64 */
            n_in = 0;
            for (int y_index = 0; y_index < sc.ny; y_index++)
                for (int x_index = 0; x_index < sc.nx; x_index++) {
                    k_window* w = &a[n_in++];
                    w->x = (uint32_t)round(x_index * stepd);
                    w->y = (uint32_t)round(y_index * stepd);
                    w->variance = compute_variance(sum, sqsum, iw, &sc, w->x, w->y, 0);
                    w->offset = iw * w->y + w->x;
                }
            /* from here on it is the per-stage loop with the flattened skip (:1069-1145 == :681-734) */
        } else {
            n_in = precompute_windows(sum, sqsum, iw, &sc, signed_mean, a);
        }
        st->windows += n_in;
        const int skip_list = (mode == 2 || mode == 5);
        if (mode == 0 || skip_list) {
            k_window* b = (k_window*)malloc(sizeof(k_window) * nwin);
            uint32_t n_out = n_in;
            for (int stage = 0; stage < c->n_stages; ++stage) {
                if (!(skip_list && stage == 0)) st->stage_entered[stage] += n_in;
                n_out = 0;
                const float thr = c->stage_threshold[stage];
                uint32_t subwindow_incr = 1;   /* runSubwindow, clod.cpp:700-733 */
                for (uint32_t g = 0; g < n_in; g += subwindow_incr) {
                    float stage_sum = stage_sum_for(c, kn, ii, &a[g], stage, st);
                    subwindow_incr = 1;
                    if (skip_list && stage == 0) st->stage_entered[0]++;
                    if (stage_sum >= thr) b[n_out++] = a[g];
                    else if (skip_list && stage == 0) subwindow_incr = 2;
                }
                k_window* t = a; a = b; b = t;
                n_in = n_out;
                if (n_out == 0) break;
            }
            for (uint32_t i = 0; i < n_out; ++i) {
                if (found < cap) {
                    out[found].x = (int32_t)a[i].x;
                    out[found].y = (int32_t)a[i].y;
                    out[found].w = sc.win_w;
                    out[found].h = sc.win_h;
                    out[found].scale_idx = scale_index;
                }
                found++;
            }
            free(b);
        } else {
            for (uint32_t g = 0; g < n_in; ++g) {
                int ptr = 0, accept = 0;
                while (ptr != -1) {
                    st->stage_entered[ptr]++;
                    float stage_sum = stage_sum_for(c, kn, ii, &a[g], ptr, st);
                    if (stage_sum >= c->stage_threshold[ptr]) {
                        ptr = c->stage_child[ptr];
                        if (ptr == -1) accept = 1;
                    } else {
                        while (ptr != -1 && c->stage_next[ptr] == -1) ptr = c->stage_parent[ptr];
                        if (ptr == -1) break;
                        ptr = c->stage_next[ptr];
                    }
                }
                if (accept) {
                    if (found < cap) {
                        out[found].x = (int32_t)a[g].x;
                        out[found].y = (int32_t)a[g].y;
                        out[found].w = sc.win_w;
                        out[found].h = sc.win_h;
                        out[found].scale_idx = scale_index;
                    }
                    found++;
                }
            }
        }
        free(a);
    }
    free(kn);
    free(sum);
    free(sqsum);
    *n_total = found;
    return found < cap ? found : cap;
}

/* ----------------------------------------------------- synthetic test images */
/* The survey's generator: 32-bit xorshift (13,17,5); `noise` = low
 * byte of the generator state after each step, row-major.                          */
void oc_xorshift_noise(uint32_t seed, uint8_t* dst, size_t n) {
    uint32_t s = seed ? seed : 1u;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        dst[i] = (uint8_t)(s & 0xffu);
    }
}

/* The generator's 32-bit words themselves (the survey's `smooth` kind derives its noise in [-8, 8] from them). */
void oc_xorshift_words(uint32_t seed, uint32_t* dst, size_t n) {
    uint32_t s = seed ? seed : 1u;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        dst[i] = s;
    }
}

/* u64 -> f32 conversion probe, for known-answer tests of the device conversion. */
float oc_u64_to_f32(uint64_t v) { return (float)v; }

/* ------------------------------------------------------ f1: grouping (next row) */
/* cv::groupRectangles as the reference keeps it in tempcv.cpp:130-243 (the clod port
 * clod.cpp:182-357 is buggy).  cv::partition is OpenCV 2.4.2 core
 * (operations.hpp, not in the reference tree): disjoint-set forest with rank and path
 * compression over all ordered pairs, classes numbered in order of first appearance —
 * restated here as published.  UNPINNED: the reference has no grouping outputs to compare. */
typedef struct oc_grect { int32_t x, y, w, h; } oc_grect;

static int oc_similar(const oc_grect* r1, const oc_grect* r2, double eps) {
    int mw = r1->w < r2->w ? r1->w : r2->w, mh = r1->h < r2->h ? r1->h : r2->h;
    double delta = eps * (mw + mh) * 0.5;
    return abs(r1->x - r2->x) <= delta && abs(r1->y - r2->y) <= delta &&
           abs(r1->x + r1->w - r2->x - r2->w) <= delta && abs(r1->y + r1->h - r2->y - r2->h) <= delta;
}

static int oc_partition(const oc_grect* vec, int N, double eps, int* labels) {
    int (*nodes)[2] = (int (*)[2])malloc(sizeof(int[2]) * (size_t)(N > 0 ? N : 1));
    int i, j;
    for (i = 0; i < N; i++) { nodes[i][0] = -1; nodes[i][1] = 0; }
    for (i = 0; i < N; i++) {
        int root = i;
        while (nodes[root][0] >= 0) root = nodes[root][0];
        for (j = 0; j < N; j++) {
            if (i == j || !oc_similar(&vec[i], &vec[j], eps)) continue;
            int root2 = j;
            while (nodes[root2][0] >= 0) root2 = nodes[root2][0];
            if (root2 != root) {
                int rank = nodes[root][1], rank2 = nodes[root2][1];
                if (rank > rank2)
                    nodes[root2][0] = root;
                else {
                    nodes[root][0] = root2;
                    nodes[root2][1] += rank == rank2;
                    root = root2;
                }
                int k = j, parent;
                while ((parent = nodes[k][0]) >= 0) { nodes[k][0] = root; k = parent; }
                k = i;
                while ((parent = nodes[k][0]) >= 0) { nodes[k][0] = root; k = parent; }
            }
        }
    }
    int nclasses = 0;
    for (i = 0; i < N; i++) {
        int root = i;
        while (nodes[root][0] >= 0) root = nodes[root][0];
        if (nodes[root][1] >= 0) nodes[root][1] = ~nclasses++;
        labels[i] = ~nodes[root][1];
    }
    free(nodes);
    return nclasses;
}

/* In place on rects[0..n); weights_out[n] receives the member counts; returns the new count. */
int oc_group_rectangles(oc_grect* rects, int n, int groupThreshold, double eps, int32_t* weights_out) {
    if (groupThreshold <= 0 || n == 0) {
        for (int i = 0; i < n; i++) weights_out[i] = 1;
        return n;
    }
    int* labels = (int*)malloc(sizeof(int) * (size_t)n);
    int nclasses = oc_partition(rects, n, eps, labels);
    oc_grect* rrects = (oc_grect*)calloc((size_t)nclasses, sizeof(oc_grect));
    int* rweights = (int*)calloc((size_t)nclasses, sizeof(int));
    for (int i = 0; i < n; i++) {
        int cls = labels[i];
        rrects[cls].x += rects[i].x; rrects[cls].y += rects[i].y;
        rrects[cls].w += rects[i].w; rrects[cls].h += rects[i].h;
        rweights[cls]++;
    }
    for (int i = 0; i < nclasses; i++) {
        oc_grect r = rrects[i];
        float s = 1.f / rweights[i];
        rrects[i].x = r.x * s > 2147483647 ? 2147483647 : (int)(r.x * s);
        rrects[i].y = r.y * s > 2147483647 ? 2147483647 : (int)(r.y * s);
        rrects[i].w = r.w * s > 2147483647 ? 2147483647 : (int)(r.w * s);
        rrects[i].h = r.h * s > 2147483647 ? 2147483647 : (int)(r.h * s);
    }
    int out = 0;
    for (int i = 0; i < nclasses; i++) {
        oc_grect r1 = rrects[i];
        int n1 = rweights[i], j;
        if (n1 <= groupThreshold) continue;
        for (j = 0; j < nclasses; j++) {
            int n2 = rweights[j];
            if (j == i || n2 <= groupThreshold) continue;
            oc_grect r2 = rrects[j];
            int dx = r2.w * eps > 2147483647 ? 2147483647 : (int)(r2.w * eps);
            int dy = r2.h * eps > 2147483647 ? 2147483647 : (int)(r2.h * eps);
            if (i != j && r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.w <= r2.x + r2.w + dx &&
                r1.y + r1.h <= r2.y + r2.h + dy && (n2 > (3 > n1 ? 3 : n1) || n1 < 3))
                break;
        }
        if (j == nclasses) { rects[out] = r1; weights_out[out] = n1; out++; }
    }
    free(labels); free(rrects); free(rweights);
    return out;
}

/* ------------------------------------------- CPU baseline #2: the OpenCV-style path */
/* Restatement of cvHaarDetectObjects' scale-cascade path as the reference keeps it in tempcv.cpp
 * (a private copy of OpenCV 2.4.2 objdetect/haar.cpp, not part of the reference's build):
 *   driver loop                       tempcv.cpp:1330-1417  (double factor, ystep = max(2, factor), cvRound)
 *   icvCreateHidHaarClassifierCascade  :307-536             (isStumpBased, is_tree, per-stage two_rects :421, :458)
 *   cvSetImagesForHaarClassifierCascade   :549-768          (cvRound-ed rects, float weights, CV_ADJUST_WEIGHTS = 0,
 *                                                            tilted rectangles :743-750 with correction 0.5 :731)
 *   icvEvalHidHaarClassifier           :771-792             (tree walk; int * float products)
 *   cvRunHaarClassifierCascadeSum      :795-972             (border rule :817-820; three evaluation paths)
 *   stage threshold bias               :262, :419           (threshold - 0.0001f)
 *   ScaleCascade invoker               :1116-1185           (ixstep = result != 0 ? 1 : 2)
 * CV_HAAR_USE_SSE is commented out in tempcv.cpp (:28-36), so the scalar branches are the specification:
 *   - stump cascade, no stage tree, stage flagged two_rects (:872-888): `double rect0 = calc_sum(..);
 *     rect0 *= weight;` — an f64 product per rectangle, `sum = rect1 + rect0`;
 *   - every other stump stage (:907-911) and icvEvalHidHaarClassifier (:783-788): `calc_sum(..) * weight`
 *     is int * float, i.e. a BINARY32 product ((float)int rounds above 2^24), widened to double afterwards
 *     and accumulated in f64.
 * calc_sum is int arithmetic (sumtype = int; wraps like the hardware does).
 * This is the "reference CPU path (OpenCV cv::CascadeClassifier ...)" north_star asks to be TIMED as a
 * baseline, and the checker of the library's OpenCV arithmetic profile (vj_detect_opencv).  OpenCV itself
 * is not installable here, so nothing pins it (SURVEY.md §8c "parity unpinned at the OpenCV boundary").   */
typedef struct cv_rectp { int p0, p1, p2, p3; float weight; } cv_rectp;
typedef struct cv_node { cv_rectp rect[3]; int nrect; int tilted; float threshold; } cv_node;

static inline int cv_round(double v) { return (int)lrint(v); }

/* The tilted integral cvIntegral returns next to sum (OpenCV 2.4.2 imgproc, not in the reference tree; its
 * documented definition: tilted(X, Y) = sum of image(x, y) over y < Y, abs(x - X + 1) <= Y - y - 1), an
 * (h+1) x (w+1) CV_32S matrix with a zero first row.  Computed with the row recurrence
 *   T[Y][X] = T[Y-1][X-1] + T[Y-1][X+1] - T[Y-2][X] + I(X-1, Y-1) + I(X-1, Y-2)
 * where T[Y][-1] = T[Y-1][0] and T[Y][w+1] = T[Y-1][w] (the triangle's apex lies outside the image);
 * tests/test_oracle_cv.py checks it against the definition by brute force.  32-bit wrap-around.           */
void oc_integral_tilted(const uint8_t* gray, int w, int h, int stride, int32_t* tilted) {
    const int sw = w + 1;
    uint32_t* T = (uint32_t*)tilted;
    for (int x = 0; x <= w; ++x) T[x] = 0;
    for (int Y = 1; Y <= h; ++Y) {
        const uint32_t* t1 = T + (size_t)(Y - 1) * sw;                     /* row Y-1 */
        const uint32_t* t2 = Y >= 2 ? T + (size_t)(Y - 2) * sw : NULL;      /* row Y-2 */
        uint32_t* row = T + (size_t)Y * sw;
        const uint8_t* i1 = gray + (size_t)(Y - 1) * stride;
        const uint8_t* i2 = Y >= 2 ? gray + (size_t)(Y - 2) * stride : NULL;
        for (int X = 0; X <= w; ++X) {
            const uint32_t left = X >= 1 ? t1[X - 1] : (t2 ? t2[0] : 0u);
            const uint32_t right = X + 1 <= w ? t1[X + 1] : (t2 ? t2[w] : 0u);
            const uint32_t up2 = t2 ? t2[X] : 0u;
            const uint32_t px = X >= 1 ? (uint32_t)i1[X - 1] + (i2 ? (uint32_t)i2[X - 1] : 0u) : 0u;
            row[X] = left + right - up2 + px;
        }
    }
}

static inline int cv_calc_sum(const int32_t* img, int po, const cv_rectp* r) {
    /* calc_sum(rect, offset) = p0[offset] - p1[offset] - p2[offset] + p3[offset], sumtype int */
    return (int)((uint32_t)img[po + r->p0] - (uint32_t)img[po + r->p1] - (uint32_t)img[po + r->p2] + (uint32_t)img[po + r->p3]);
}

/* icvEvalHidHaarClassifier's node sum (tempcv.cpp:783-788), also the non-two_rects stump stages (:907-911) */
static int g_all_f64 = 0;   /* test hook, see detect_opencvlike_impl */
static inline double cv_node_sum_f32(const int32_t* sum, const int32_t* tilted, int po, const cv_node* k) {
    const int32_t* img = k->tilted ? tilted : sum;
    if (g_all_f64) {
        double s = (double)cv_calc_sum(img, po, &k->rect[0]) * (double)k->rect[0].weight;
        s += (double)cv_calc_sum(img, po, &k->rect[1]) * (double)k->rect[1].weight;
        if (k->nrect > 2) s += (double)cv_calc_sum(img, po, &k->rect[2]) * (double)k->rect[2].weight;
        return s;
    }
    double s = (double)((float)cv_calc_sum(img, po, &k->rect[0]) * k->rect[0].weight);
    s += (double)((float)cv_calc_sum(img, po, &k->rect[1]) * k->rect[1].weight);
    if (k->nrect > 2) s += (double)((float)cv_calc_sum(img, po, &k->rect[2]) * k->rect[2].weight);
    return s;
}

/* all_f64 != 0 is NOT the reference: it multiplies every rectangle in f64 (what round 1 of this repo did by
 * mistake).  It exists so that tests can show the literal arithmetic matters (tests/test_oracle_cv.py).   */
static int detect_opencvlike_impl(const oc_cascade* c, const uint8_t* gray, int W, int H, int stride,
                         int min_w, int min_h, double scaleFactor, int all_f64,
                         oc_rect* out, int cap, int* n_total, oc_stats* st) {
    const int sw = W + 1;
    int32_t* sum = (int32_t*)calloc((size_t)sw * (H + 3), sizeof(int32_t));
    double* sqsum = (double*)calloc((size_t)sw * (H + 3), sizeof(double));
    int32_t* tilted = NULL;
    cv_node* kn = (cv_node*)malloc(sizeof(cv_node) * (size_t)c->n_nodes);
    memset(st, 0, sizeof(*st));
    g_all_f64 = all_f64;
    oc_integral(gray, W, H, stride, sum, sqsum);
    /* icvCreateHidHaarClassifierCascade: cascade-wide and per-stage flags (tempcv.cpp:410-470) */
    int is_stump_based = 1, is_tree = 0, has_tilted = 0;
    int two_rects[64];
    for (int t = 0; t < c->n_trees; ++t) is_stump_based &= c->tree_n_nodes[t] == 1;
    for (int i = 0; i < c->n_stages && i < 64; ++i) {
        is_tree |= c->stage_next[i] != -1;
        two_rects[i] = 1;
        const int t0 = c->stage_first_tree[i], t1 = t0 + c->stage_n_trees[i];
        (void)all_f64;
        for (int t = t0; t < t1; ++t)
            for (int l = 0; l < c->tree_n_nodes[t]; ++l) {
                const int n = c->tree_first_node[t] + l;
                const int32_t* r2 = c->node_rect + (n * 3 + 2) * 4;
                /* :452-457: rect[2] counts unless |weight| < DBL_EPSILON or its width / height is 0 */
                if (!(fabs((double)c->node_weight[n * 3 + 2]) < 2.220446049250313e-16 || r2[2] == 0 || r2[3] == 0))
                    two_rects[i] = 0;
                if (c->node_tilted && c->node_tilted[n]) has_tilted = 1;
            }
    }
    if (has_tilted) {
        tilted = (int32_t*)calloc((size_t)sw * (H + 3), sizeof(int32_t));
        oc_integral_tilted(gray, W, H, stride, tilted);
    }
    int found = 0, n_factors = 0, scale_index = 0;
    double factor;
    for (n_factors = 0, factor = 1; factor * c->win_w < W - 10 && factor * c->win_h < H - 10;
         n_factors++, factor *= scaleFactor) {}
    factor = 1;
    for (; n_factors-- > 0; factor *= scaleFactor, scale_index++) {
        const double ystep = 2. > factor ? 2. : factor;
        const int win_w = cv_round(c->win_w * factor), win_h = cv_round(c->win_h * factor);
        const int endX = cv_round((W - win_w) / ystep), endY = cv_round((H - win_h) / ystep);
        if (win_w < min_w || win_h < min_h) continue;
        /* cvSetImagesForHaarClassifierCascade */
        const int ex = cv_round(factor), ew = cv_round((c->win_w - 2) * factor), eh = cv_round((c->win_h - 2) * factor);
        const double weight_scale = 1. / (ew * eh);
        const int q0 = ex * sw + ex, q1 = ex * sw + ex + ew, q2 = (ex + eh) * sw + ex, q3 = (ex + eh) * sw + ex + ew;
        for (int n = 0; n < c->n_nodes; ++n) {
            double sum0 = 0, area0 = 0;
            /* rects 0 and 1 always count (their p0 is never null); rect 2 unless icvCreateHid... zeroed it (:452-455) */
            const int32_t* r2 = c->node_rect + (n * 3 + 2) * 4;
            const int nr = (fabs((double)c->node_weight[n * 3 + 2]) < 2.220446049250313e-16 || r2[2] == 0 || r2[3] == 0) ? 2 : 3;
            kn[n].nrect = nr;
            kn[n].tilted = c->node_tilted ? c->node_tilted[n] != 0 : 0;
            kn[n].threshold = c->node_threshold[n];
            const double correction_ratio = weight_scale * (!kn[n].tilted ? 1 : 0.5);   /* :731 */
            for (int k = 0; k < nr; ++k) {
                const int32_t* r = c->node_rect + (n * 3 + k) * 4;
                const int tx = cv_round(r[0] * factor), ty = cv_round(r[1] * factor);
                const int tw = cv_round(r[2] * factor), th = cv_round(r[3] * factor);
                if (!kn[n].tilted) {
                    kn[n].rect[k].p0 = ty * sw + tx;
                    kn[n].rect[k].p1 = ty * sw + tx + tw;
                    kn[n].rect[k].p2 = (ty + th) * sw + tx;
                    kn[n].rect[k].p3 = (ty + th) * sw + tx + tw;
                } else {   /* :743-750 */
                    kn[n].rect[k].p2 = (ty + tw) * sw + tx + tw;
                    kn[n].rect[k].p3 = (ty + tw + th) * sw + tx + tw - th;
                    kn[n].rect[k].p0 = ty * sw + tx;
                    kn[n].rect[k].p1 = (ty + th) * sw + tx - th;
                }
                kn[n].rect[k].weight = (float)(c->node_weight[n * 3 + k] * correction_ratio);
                if (k == 0) area0 = tw * th;
                else sum0 += kn[n].rect[k].weight * tw * th;
            }
            kn[n].rect[0].weight = (float)(-sum0 / area0);
        }
        for (int iy = 0; iy < endY; iy++) {
            const int y = cv_round(iy * ystep);
            int ixstep = 1;
            for (int ix = 0; ix < endX; ix += ixstep) {
                const int x = cv_round(ix * ystep);
                int result;
                st->windows++;
                if (x < 0 || y < 0 || x + win_w >= sw || y + win_h >= H + 1) {
                    result = -1;
                } else {
                    const int po = y * sw + x;
                    double mean = (double)(int)((uint32_t)sum[po + q0] - (uint32_t)sum[po + q1] - (uint32_t)sum[po + q2] +
                                                (uint32_t)sum[po + q3]) * weight_scale;
                    double vnf = sqsum[po + q0] - sqsum[po + q1] - sqsum[po + q2] + sqsum[po + q3];
                    vnf = vnf * weight_scale - mean * mean;
                    vnf = vnf >= 0. ? sqrt(vnf) : 1.;
                    if (is_tree) {   /* :834-861: any reject returns 0 */
                        int ptr = 0;
                        result = 1;
                        while (ptr != -1) {
                            double stage_sum = 0.0;
                            const int t0 = c->stage_first_tree[ptr], t1 = t0 + c->stage_n_trees[ptr];
                            st->stage_entered[ptr]++;
                            for (int t = t0; t < t1; ++t) {
                                const int n0 = c->tree_first_node[t];
                                const float* alpha = c->alpha + c->tree_first_alpha[t];
                                int idx = 0;
                                do {
                                    const cv_node* k = kn + n0 + idx;
                                    const double tt = k->threshold * vnf;
                                    const double s = cv_node_sum_f32(sum, tilted, po, k);
                                    st->stump_evals++;
                                    idx = s < tt ? c->node_left[n0 + idx] : c->node_right[n0 + idx];
                                } while (idx > 0);
                                stage_sum += alpha[-idx];
                            }
                            if (stage_sum >= c->stage_threshold[ptr] - 0.0001f) {
                                ptr = c->stage_child[ptr];
                            } else {
                                while (ptr != -1 && c->stage_next[ptr] == -1) ptr = c->stage_parent[ptr];
                                if (ptr == -1) { result = 0; break; }
                                ptr = c->stage_next[ptr];
                            }
                        }
                    } else {
                        result = 1;
                        for (int i = 0; i < c->n_stages; ++i) {
                            double stage_sum = 0.0;
                            const int t0 = c->stage_first_tree[i], t1 = t0 + c->stage_n_trees[i];
                            st->stage_entered[i]++;
                            for (int t = t0; t < t1; ++t) {
                                const int n0 = c->tree_first_node[t];
                                const float* alpha = c->alpha + c->tree_first_alpha[t];
                                if (is_stump_based) {
                                    const cv_node* k = kn + n0;
                                    const double tt = k->threshold * vnf;
                                    double s;
                                    st->stump_evals++;
                                    if (two_rects[i]) {   /* :872-888 */
                                        const int32_t* img = k->tilted ? tilted : sum;
                                        double rect0 = cv_calc_sum(img, po, &k->rect[0]);
                                        rect0 *= k->rect[0].weight;
                                        double rect1 = cv_calc_sum(img, po, &k->rect[1]);
                                        rect1 *= k->rect[1].weight;
                                        s = rect1 + rect0;
                                    } else {              /* :907-911 */
                                        s = cv_node_sum_f32(sum, tilted, po, k);
                                    }
                                    stage_sum += alpha[s >= tt];
                                } else {                  /* :952-957 via icvEvalHidHaarClassifier */
                                    int idx = 0;
                                    do {
                                        const cv_node* k = kn + n0 + idx;
                                        const double tt = k->threshold * vnf;
                                        const double s = cv_node_sum_f32(sum, tilted, po, k);
                                        st->stump_evals++;
                                        idx = s < tt ? c->node_left[n0 + idx] : c->node_right[n0 + idx];
                                    } while (idx > 0);
                                    stage_sum += alpha[-idx];
                                }
                            }
                            if (stage_sum < c->stage_threshold[i] - 0.0001f) { result = -i; break; }
                        }
                    }
                }
                if (result > 0) {
                    if (found < cap) { out[found].x = x; out[found].y = y; out[found].w = win_w; out[found].h = win_h; out[found].scale_idx = scale_index; }
                    found++;
                }
                ixstep = result != 0 ? 1 : 2;
            }
        }
    }
    free(kn); free(sum); free(sqsum); free(tilted);
    g_all_f64 = 0;
    *n_total = found;
    return found < cap ? found : cap;
}

int oc_detect_opencvlike(const oc_cascade* c, const uint8_t* gray, int W, int H, int stride,
                         int min_w, int min_h, double scaleFactor,
                         oc_rect* out, int cap, int* n_total, oc_stats* st) {
    return detect_opencvlike_impl(c, gray, W, H, stride, min_w, min_h, scaleFactor, 0, out, cap, n_total, st);
}

/* NOT the reference's arithmetic (single-threaded test hook): every product in f64. */
int oc_detect_opencvlike_all_f64(const oc_cascade* c, const uint8_t* gray, int W, int H, int stride,
                                 int min_w, int min_h, double scaleFactor,
                                 oc_rect* out, int cap, int* n_total, oc_stats* st) {
    return detect_opencvlike_impl(c, gray, W, H, stride, min_w, min_h, scaleFactor, 1, out, cap, n_total, st);
}

/* --------------------------------------------------------------- image ingest */
/* BGR / BGRA -> gray as setupImage gets it from cvCvtColor(CV_BGR2GRAY) (clif.cpp:326-335, :328).  The
 * arithmetic is OpenCV 2.4.2's 8-bit path (imgproc color.cpp, RGB2Gray<uchar>: coefficients 1868 / 9617 /
 * 4899 at 14 fractional bits, rounding term 1 << 13) — third-party code that is NOT under /root/reference,
 * so this is its published formula, parity unpinned; clif.cl:12-15's float formula is a
 * different, unused kernel.  channels = 3 or 4, interleaved, B first.                                   */
void oc_bgr2gray(const uint8_t* bgr, int W, int H, int stride, int channels, uint8_t* gray, int gray_stride) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t* p = bgr + (size_t)y * stride + (size_t)x * channels;
            gray[(size_t)y * gray_stride + x] = (uint8_t)((p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + 8192u) >> 14);
        }
}
