// gfx950 kernels of the OpenCV arithmetic profile: cvHaarDetectObjects' scale-cascade path as the
// reference keeps it in tempcv.cpp (a private copy of OpenCV 2.4.2 haar.cpp; CV_HAAR_USE_SSE is commented
// out at :28-36, so the scalar branches are the specification) —
//   cvRunHaarClassifierCascadeSum   tempcv.cpp:795-972   f64 variance and stage sums, border rule, three paths:
//       stage tree (:834-861), stump cascade (:863-945, per-stage two_rects), general trees (:947-964)
//   icvEvalHidHaarClassifier        tempcv.cpp:771-792   sum < t ? left : right, int * float products
//   HaarDetectObjects_ScaleCascade_Invoker  :1116-1185    x = cvRound(ix * ystep), ixstep = result != 0 ? 1 : 2
//   tilted rectangles               tempcv.cpp:743-750   four corners in the tilted integral image
// Second arithmetic profile of the library; the clod profile (vj_kernels.hip) is the
// contract of the headline path.  MUST be compiled with -ffp-contract=off.
//
// One wave walks one window row.  The sequential "skip the next window after a reject" rule is a
// recurrence over the row, e[i] = !(e[i-1] && f[i-1]); because a window that follows a non-reject is always
// visited, e[i] only depends on the PARITY of the run of rejects that ends at i-1 (f computed for
// every grid position), which a wave gets from one __ballot per 64 positions plus one carry bit.  For linear
// cascades only a stage-0 reject skips (the later stages return -i != 0), so f is the stage-0 verdict and the
// later stages run on the visited survivors; a stage tree returns 0 on ANY reject, so there f is the verdict of
// the whole tree, evaluated for every grid position.
#include <hip/hip_runtime.h>
#include "vj_device.hpp"
#include "vj_devutil.hpp"

namespace vj {

constexpr uint32_t CV_TREE_SEG_GROUPS = 16;   // stage trees: a row is resolved in segments of 16 x 64 grid positions
struct CvQEntry {
    uint32_t off;   // byte offset of the window origin in the batch sum image
    uint32_t xy;    // x | y << 16
    double vnf;     // variance_norm_factor
};

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }   // cvRound: half to even

// calc_sum(rect, offset) = p0 - p1 - p2 + p3 in int (sumtype; tempcv.cpp:118-121): corner q of rectangle k sits
// at lt + {0, da, db, da + db}.  Upright: da = width, db = height * stride; tilted (:743-750): da = height *
// (stride - 1), db = width * (stride + 1).
__device__ __forceinline__ int32_t cv_calc_sum(rsrc_t img, uint32_t off, uint32_t lt, uint32_t da, uint32_t db) {
    return (int32_t)(ld_u32(img, off, lt) - ld_u32(img, off, lt + da) - ld_u32(img, off, lt + db) + ld_u32(img, off, lt + da + db));
}

// One node's weighted rectangle sum.  F64 = a stump stage flagged two_rects (tempcv.cpp:872-888):
// `double rect0 = calc_sum(..); rect0 *= weight; ... sum = rect1 + rect0` — f64 products.  Otherwise (:783-788,
// :907-911) `calc_sum(..) * weight` is int * float: the int is converted to binary32 (rounding above 2^24), the
// product is a binary32 product, and only then is it widened to double and accumulated.
template <bool F64>
__device__ __forceinline__ double cv_node_sum(rsrc_t sum_img, rsrc_t tilt_img, const NodeRecDev& r, uint32_t off) {
    const rsrc_t img = (r[15] & CV_NODE_TILTED) ? tilt_img : sum_img;   // uniform
    const int32_t r0 = cv_calc_sum(img, off, r[0], r[3], r[6]);
    const int32_t r1 = cv_calc_sum(img, off, r[1], r[4], r[7]);
    const float w0 = __uint_as_float(r[9]), w1 = __uint_as_float(r[10]), w2 = __uint_as_float(r[11]);
    if (F64) {
        const double rect0 = (double)r0 * (double)w0;
        const double rect1 = (double)r1 * (double)w1;
        return rect1 + rect0;   // two_rects: there is no third rectangle
    }
    double s = (double)((float)r0 * w0);
    s += (double)((float)r1 * w1);
    if (w2 != 0.0f) {   // uniform (node->feature.rect[2].p0 != 0)
        const int32_t r2 = cv_calc_sum(img, off, r[2], r[5], r[8]);
        s += (double)((float)r2 * w2);
    }
    return s;
}

// One stage on one window: stumps through the scalar cache; multi-node trees visit their records in index
// order under the lanes whose walk sits on them (a child always follows its parent), as stage_sum_trees does.
template <bool TREES, bool F64>
__device__ __forceinline__ double cv_stage_sum(rsrc_t img, rsrc_t timg, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off,
                                               double vnf) {
    double stage_sum = 0.0;
    if (!TREES) {
        NodeRecDev r = tab[0];
        for (uint32_t j = 0; j < n_nodes; ++j) {
            const NodeRecDev rn = tab[j + 1 < n_nodes ? j + 1 : j];
            const double t = (double)__uint_as_float(r[12]) * vnf;
            const double s = cv_node_sum<F64>(img, timg, r, off);
            stage_sum += (double)(s < t ? __uint_as_float(r[13]) : __uint_as_float(r[14]));   // alpha[sum >= t]
            r = rn;
        }
        return stage_sum;
    }
    uint32_t cur = 0, k = 0;
    float value = 0.0f;
    bool done = false;
    for (uint32_t j = 0; j < n_nodes; ++j) {
        const NodeRecDev r = tab[j];
        const uint32_t flags = r[15];
        if (!done && cur == k) {
            const double t = (double)__uint_as_float(r[12]) * vnf;
            const bool go_left = cv_node_sum<false>(img, timg, r, off) < t;
            const uint32_t nxt = go_left ? r[13] : r[14];
            if (go_left ? (flags & 1u) != 0u : (flags & 2u) != 0u) {
                cur = nxt;
            } else {
                value = __uint_as_float(nxt);
                done = true;
            }
        }
        ++k;
        if (flags & 4u) {   // last record of the tree (uniform)
            stage_sum += (double)value;
            cur = 0;
            k = 0;
            done = false;
        }
    }
    return stage_sum;
}

// Two-node trees (a root and its only node child — every tree of frontalface_alt2; upright features) with BOTH nodes' gathers
// in flight: the walk above pays two memory round trips per tree — the child's under the lanes that go there — and a thin
// sweep waits for each.  The tree's value is the walk's (tempcv.cpp:771-792: idx = sum < t ? left : right until idx <= 0):
// the child's leaf where the root's side is a node, else the root's leaf; node sums int * float widened to double (:783-788).
__device__ __forceinline__ double cv_stage_sum_tree2(rsrc_t img, kptr<NodeRecDev> tab, uint32_t n_trees, uint32_t off, double vnf) {
    double stage_sum = 0.0;
    NodeRecDev ra = tab[0], rb = tab[1];
    for (uint32_t t = 0; t < n_trees; ++t) {
        const uint32_t tn = t + 1u < n_trees ? t + 1u : t;
        const NodeRecDev na = tab[2u * tn], nb = tab[2u * tn + 1u];   // the next tree travels meanwhile
        const int32_t a0 = cv_calc_sum(img, off, ra[0], ra[3], ra[6]), a1 = cv_calc_sum(img, off, ra[1], ra[4], ra[7]);
        const int32_t b0 = cv_calc_sum(img, off, rb[0], rb[3], rb[6]), b1 = cv_calc_sum(img, off, rb[1], rb[4], rb[7]);
        double sa = (double)((float)a0 * __uint_as_float(ra[9]));
        sa += (double)((float)a1 * __uint_as_float(ra[10]));
        double sb = (double)((float)b0 * __uint_as_float(rb[9]));
        sb += (double)((float)b1 * __uint_as_float(rb[10]));
        const float wa2 = __uint_as_float(ra[11]), wb2 = __uint_as_float(rb[11]);
        if (wa2 != 0.0f || wb2 != 0.0f) {   // uniform (an absent third rectangle has lt = da = db = 0: four reads of the origin)
            const int32_t a2 = cv_calc_sum(img, off, ra[2], ra[5], ra[8]), b2 = cv_calc_sum(img, off, rb[2], rb[5], rb[8]);
            if (wa2 != 0.0f) sa += (double)((float)a2 * wa2);
            if (wb2 != 0.0f) sb += (double)((float)b2 * wb2);
        }
        const uint32_t flags = ra[15];
        const bool left_a = sa < (double)__uint_as_float(ra[12]) * vnf, left_b = sb < (double)__uint_as_float(rb[12]) * vnf;
        const bool to_child = left_a ? (flags & 1u) != 0u : (flags & 2u) != 0u;
        const float leaf_a = left_a ? __uint_as_float(ra[13]) : __uint_as_float(ra[14]);
        const float leaf_b = left_b ? __uint_as_float(rb[13]) : __uint_as_float(rb[14]);
        stage_sum += (double)(to_child ? leaf_b : leaf_a);
        ra = na;
        rb = nb;
    }
    return stage_sum;
}

// Stage sum with the stage's arithmetic mode (StageDev::cv_f64, host-computed: two_rects && stump cascade && no
// stage tree); `tree2`: CvArgs::tree2 (uniform).
template <bool TREES>
__device__ __forceinline__ double cv_stage_sum_mode(rsrc_t img, rsrc_t timg, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off,
                                                    double vnf, uint32_t f64, uint32_t tree2 = 0u) {
    if (!TREES && f64 != 0u) return cv_stage_sum<false, true>(img, timg, tab, n_nodes, off, vnf);
    if (TREES && tree2 != 0u) return cv_stage_sum_tree2(img, tab, n_nodes >> 1, off, vnf);
    return cv_stage_sum<TREES, false>(img, timg, tab, n_nodes, off, vnf);
}

// ------------------------------------------------------------------------ stage trees made of chains (CvChainDev)
// The windows that survive a stage tree's linear prefix used to carry a target stage each and ride through ONE sweep of all
// remaining stages in chunks of 64 — ever fewer lanes evaluating, 40 stages long.  A tree made of chains (frontalface_alt_tree)
// is swept like a linear cascade instead: the population of a chain is compacted after every stage (full lanes while more than
// 64 windows are left), its rejects are set aside and become the population of the next chain, and a population of at most
// CV_TAIL_MAX windows evaluates a stage stump-parallel (lane = stump, the verdict bits replayed in stump order).  Arithmetic
// and order of the additions per window are those of cv_stage_sum (tempcv.cpp:771-792, :834-861).

// One stump stage on the lane's window, two stumps per step with all of their gathers in flight (a thin sweep pays a memory
// round trip per step); the leaf values are added in stump order.  Stage trees never take the two_rects f64 branch
// (StageDev::cv_f64 is 0 for them): int * float products widened to double (:783-788).
template <bool F64 = false>
__device__ __forceinline__ double cv_stage_sum_pairs(rsrc_t img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off, double vnf) {
    double stage_sum = 0.0;
    uint32_t j = 0;
    if (n_nodes >= 2u) {
        NodeRecDev ra = tab[0], rb = tab[1];
        for (; j + 1u < n_nodes; j += 2u) {
            const uint32_t ja = j + 2u < n_nodes ? j + 2u : j, jb = j + 3u < n_nodes ? j + 3u : j + 1u;
            const NodeRecDev na = tab[ja], nb = tab[jb];   // the next pair travels meanwhile
            const int32_t a0 = cv_calc_sum(img, off, ra[0], ra[3], ra[6]), a1 = cv_calc_sum(img, off, ra[1], ra[4], ra[7]);
            const int32_t b0 = cv_calc_sum(img, off, rb[0], rb[3], rb[6]), b1 = cv_calc_sum(img, off, rb[1], rb[4], rb[7]);
            double sa, sb;
            if (F64) {   // a two_rects stump stage (tempcv.cpp:872-888): f64 products, rect1 + rect0
                sa = (double)a1 * (double)__uint_as_float(ra[10]) + (double)a0 * (double)__uint_as_float(ra[9]);
                sb = (double)b1 * (double)__uint_as_float(rb[10]) + (double)b0 * (double)__uint_as_float(rb[9]);
            } else {
                sa = (double)((float)a0 * __uint_as_float(ra[9]));
                sa += (double)((float)a1 * __uint_as_float(ra[10]));
                sb = (double)((float)b0 * __uint_as_float(rb[9]));
                sb += (double)((float)b1 * __uint_as_float(rb[10]));
            }
            const float wa2 = __uint_as_float(ra[11]), wb2 = __uint_as_float(rb[11]);
            if (!F64 && (wa2 != 0.0f || wb2 != 0.0f)) {   // uniform (an absent third rectangle has lt = da = db = 0: four reads of the origin)
                const int32_t a2 = cv_calc_sum(img, off, ra[2], ra[5], ra[8]), b2 = cv_calc_sum(img, off, rb[2], rb[5], rb[8]);
                if (wa2 != 0.0f) sa += (double)((float)a2 * wa2);
                if (wb2 != 0.0f) sb += (double)((float)b2 * wb2);
            }
            stage_sum += (double)(sa < (double)__uint_as_float(ra[12]) * vnf ? __uint_as_float(ra[13]) : __uint_as_float(ra[14]));
            stage_sum += (double)(sb < (double)__uint_as_float(rb[12]) * vnf ? __uint_as_float(rb[13]) : __uint_as_float(rb[14]));
            ra = na;
            rb = nb;
        }
    }
    if (j < n_nodes) {
        const NodeRecDev r = tab[j];
        const double s = cv_node_sum<F64>(img, img, r, off);
        stage_sum += (double)(s < (double)__uint_as_float(r[12]) * vnf ? __uint_as_float(r[13]) : __uint_as_float(r[14]));
    }
    return stage_sum;
}

// One stump stage for a THIN population q[0, n), n <= CV_TAIL_MAX: lane j takes stump j of a block of 64 (its record arrives
// with four coalesced 16-byte loads), every window is evaluated by all lanes at once (window offset uniform, corner offsets
// per lane), a __ballot gives the block's verdict bits; then lane w adds window w's leaf values IN STUMP ORDER (the leaf values
// come through the scalar cache).  Returns the pass mask (bit w: window w passes).  Upright features only (the caller checks).
// `masks`: n x CV_TAIL_BLOCKS words of LDS scratch.
template <bool F64 = false, typename E>
__device__ __forceinline__ unsigned long long cv_tail_stage(rsrc_t img, const uint32_t* recs_g, kptr<NodeRecDev> tab, uint32_t n_nodes, double thr_stage,
                                                            const E* q, uint32_t n, unsigned long long* masks, uint32_t lane) {
    const uint32_t n_blocks = (n_nodes + 63u) >> 6;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t j = b * 64u + lane;
        const bool active = j < n_nodes;
        const uint4* rp = reinterpret_cast<const uint4*>(recs_g + (size_t)(active ? j : 0u) * 16u);
        const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
        // CvNodeRec: lt[3] da[3] db[3] w[3] thr left right flags
        const uint32_t lt0 = r0.x, lt1 = r0.y, lt2 = r0.z, da0 = r0.w, da1 = r1.x, da2 = r1.y, db0 = r1.z, db1 = r1.w, db2 = r2.x;
        const float w0 = __uint_as_float(r2.y), w1 = __uint_as_float(r2.z), w2 = __uint_as_float(r2.w), thr_node = __uint_as_float(r3.x);
        for (uint32_t w = 0; w < n; ++w) {
            const E e = q[w];   // broadcast
            const uint32_t uo = __builtin_amdgcn_readfirstlane(e.off);
            auto rect = [&](uint32_t lt, uint32_t da, uint32_t db) {
                return (int32_t)(ld_u32(img, lt, uo) - ld_u32(img, lt + da, uo) - ld_u32(img, lt + db, uo) + ld_u32(img, lt + da + db, uo));
            };
            const int32_t c0 = rect(lt0, da0, db0), c1 = rect(lt1, da1, db1);
            double sum;
            if (F64) {   // two_rects stump stage: f64 products, rect1 + rect0 (tempcv.cpp:872-888)
                sum = (double)c1 * (double)w1 + (double)c0 * (double)w0;
            } else {
                const int32_t c2 = rect(lt2, da2, db2);
                sum = (double)((float)c0 * w0);
                sum += (double)((float)c1 * w1);
                const double with2 = sum + (double)((float)c2 * w2);
                sum = w2 != 0.0f ? with2 : sum;
            }
            const unsigned long long m = __ballot(active && !(sum < (double)thr_node * e.vnf));   // bit: alpha[1] (right)
            if (lane == 0) masks[w * CV_TAIL_BLOCKS + b] = m;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bool have = lane < n;
    double stage_sum = 0.0;
    kptr<uint32_t> leaf = reinterpret_cast<kptr<uint32_t>>(tab);   // record k: dwords 13 / 14 = left / right value
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const unsigned long long m = masks[(have ? lane : 0u) * CV_TAIL_BLOCKS + b];
        const uint32_t jn = min(64u, n_nodes - b * 64u);
#pragma unroll 4
        for (uint32_t k = 0; k < jn; ++k) {
            const uint32_t j = b * 64u + k;
            const float l = __uint_as_float(leaf[j * 16u + 13u]), r = __uint_as_float(leaf[j * 16u + 14u]);
            stage_sum += (double)(((m >> k) & 1ull) != 0ull ? r : l);
        }
    }
    __builtin_amdgcn_wave_barrier();   // every lane has read its masks
    return __ballot(have && stage_sum >= thr_stage);
}

// 16- / 24-byte entries to and from the per-wave fail list in global memory.  The list is written with plain stores and read
// back by the SAME wave: the reads go to the L2 (agent-scope relaxed loads: sc1), behind a drain of the wave's stores.
template <typename E>
__device__ __forceinline__ E cv_load_entry_l2(const E* p) {
    static_assert(sizeof(E) % 8 == 0, "entries are multiples of 8 bytes");
    E e;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&e);
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p);
#pragma unroll
    for (uint32_t k = 0; k < sizeof(E) / 8u; ++k) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return e;
}

// Sweep the chains of `ch` over the windows q[0, n) (LDS; entries carry .off and .vnf), all of one scale.  `final_(e, mine,
// accepted)` is called in wave-uniform control flow for windows whose walk ends: lanes with `mine` set hold such a window.
// `fail_g`: room for the population in global memory (this wave's own).
template <bool TREES, typename E, typename Final>
__device__ __forceinline__ void cv_chain_sweep(const CvChainDev& ch, kptr<StageDev> stages, rsrc_t img, rsrc_t timg, bool tail_ok, const uint32_t* table_g,
                                               kptr<NodeRecDev> table, E* q, uint32_t n, E* fail_g, unsigned long long* masks, uint32_t lane,
                                               Final final_) {
    for (uint32_t k = 0; k < ch.n && n != 0u; ++k) {
        const bool chained = ((ch.chained >> k) & 1u) != 0u;
        uint32_t nf = 0;
        for (uint32_t pos = ch.begin[k]; pos < ch.end[k] && n != 0u; ++pos) {
            const uint32_t s = stages[pos].order;
            const uint32_t first_node = stages[s].first_node, n_nodes = stages[s].n_nodes;
            const double thr = (double)stages[s].threshold;
            kptr<NodeRecDev> tab = table + first_node;
            if (!TREES && tail_ok && n <= ch.tail_max && n_nodes >= 16u && n_nodes <= CV_TAIL_BLOCKS * 64u) {   // uniform
                const unsigned long long pm = cv_tail_stage(img, table_g + (size_t)first_node * 16u, tab, n_nodes, thr, q, n, masks, lane);
                const bool have = lane < n;
                const E e = q[have ? lane : 0u];
                const bool pass = ((pm >> lane) & 1ull) != 0ull;
                const unsigned long long fm = __ballot(have && !pass);
                if (fm != 0ull) {
                    if (chained) {
                        if (have && !pass) fail_g[nf + mbcnt(fm)] = e;
                        nf += (uint32_t)__popcll(fm);
                    } else {
                        final_(e, have && !pass, false);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (pass) q[mbcnt(pm)] = e;
                n = (uint32_t)__popcll(pm);
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            uint32_t m = 0;
            for (uint32_t base = 0; base < n; base += 64u) {
                const uint32_t i = base + lane;
                const bool act = i < n;
                const E e = q[act ? i : 0u];
                bool pass = false;
                if (act) {
                    if (TREES) pass = cv_stage_sum<true, false>(img, timg, tab, n_nodes, e.off, e.vnf) >= thr;
                    else if (tail_ok) pass = cv_stage_sum_pairs(img, tab, n_nodes, e.off, e.vnf) >= thr;
                    else pass = cv_stage_sum<false, false>(img, timg, tab, n_nodes, e.off, e.vnf) >= thr;   // (tilted features)
                }
                const unsigned long long fm = __ballot(act && !pass);
                if (fm != 0ull) {
                    if (chained) {
                        if (act && !pass) fail_g[nf + mbcnt(fm)] = e;
                        nf += (uint32_t)__popcll(fm);
                    } else {
                        final_(e, act && !pass, false);
                    }
                }
                const unsigned long long mask = __ballot(pass);
                __builtin_amdgcn_wave_barrier();   // every lane has read its entry before any lane overwrites
                if (pass) q[m + mbcnt(mask)] = e;
                m += (uint32_t)__popcll(mask);
                __builtin_amdgcn_wave_barrier();
            }
            n = m;
        }
        for (uint32_t base = 0; base < n; base += 64u) {   // passed the chain's last stage: accepted
            const bool act = base + lane < n;
            const E e = q[act ? base + lane : 0u];
            final_(e, act, true);
        }
        if (!chained) break;
        // the chain's rejects are the next chain's population
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < nf; i += 64u) q[i] = cv_load_entry_l2(fail_g + i);
        __builtin_amdgcn_wave_barrier();
        n = nf;
    }
}

template <bool TREES, bool COUNT>
__device__ __forceinline__ void cv_flush(const CvArgs& a, rsrc_t img, rsrc_t timg, kptr<NodeRecDev> table, CvQEntry* q, uint32_t& n,
                                         uint32_t slot, uint32_t frame, uint32_t lane) {
    kptr<StageDev> stages = as_k(a.stages);
    for (uint32_t s = 1; s < a.n_stages && n != 0u; ++s) {
        if (COUNT && lane == 0) atomicAdd(a.stage_entered + s, (unsigned long long)n);
        kptr<NodeRecDev> tab = table + stages[s].first_node;
        const uint32_t n_nodes = stages[s].n_nodes, f64 = stages[s].cv_f64;
        const double thr = (double)stages[s].threshold;
        uint32_t m = 0;
        const bool upright = !TREES && a.tilted == nullptr;   // (the pair / stump-parallel forms read the upright sum image only)
        if (upright && n <= a.tail_max && n_nodes >= 16u && n_nodes <= CV_TAIL_BLOCKS * 64u) {
            // a thin population: the stage stump-parallel (lane = stump), verdict bits replayed in stump order (cv_tail_stage)
            const uint32_t* recs_g = reinterpret_cast<const uint32_t*>((uintptr_t)(table + stages[s].first_node));
            unsigned long long* masks = reinterpret_cast<unsigned long long*>(q + CV_TAIL_MAX);
            const unsigned long long pm = f64 != 0u ? cv_tail_stage<true>(img, recs_g, tab, n_nodes, thr, q, n, masks, lane)
                                                    : cv_tail_stage<false>(img, recs_g, tab, n_nodes, thr, q, n, masks, lane);
            const CvQEntry e = q[lane < n ? lane : 0u];
            __builtin_amdgcn_wave_barrier();
            if ((pm >> lane) & 1ull) q[mbcnt(pm)] = e;
            n = (uint32_t)__popcll(pm);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + lane;
            const bool act = i < n;
            const CvQEntry e = q[act ? i : 0u];
            bool pass = false;
            if (act && upright && a.pairs != 0u) pass = (f64 != 0u ? cv_stage_sum_pairs<true>(img, tab, n_nodes, e.off, e.vnf) : cv_stage_sum_pairs<false>(img, tab, n_nodes, e.off, e.vnf)) >= thr;
            else if (act) pass = cv_stage_sum_mode<TREES>(img, timg, tab, n_nodes, e.off, e.vnf, f64, a.tree2) >= thr;
            const unsigned long long mask = __ballot(pass);
            __builtin_amdgcn_wave_barrier();
            if (pass) q[m + mbcnt(mask)] = e;
            m += (uint32_t)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
        }
        n = m;
    }
    if (n != 0u) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(a.det_count, n);
        g = __builtin_amdgcn_readfirstlane(g);
        for (uint32_t i = lane; i < n; i += 64u)
            if (g + i < a.det_cap) a.det[g + i] = CvDet{q[i].xy & 0xffffu, q[i].xy >> 16, slot, frame};
    }
    n = 0;
}

// Which of 64 consecutive grid positions the sequential walk visits, given the reject bits F of all of them and
// the parity `carry` of the reject run that ends just before the first one; updates carry for the next 64.
__device__ __forceinline__ bool cv_visited(unsigned long long F, uint32_t lane, uint32_t n_valid, uint32_t& carry) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long zeros = ~F & below;
    uint32_t parity;
    if (zeros == 0ull) parity = (lane & 1u) ^ carry;
    else parity = (lane - 1u - (63u - (uint32_t)__clzll((long long)zeros))) & 1u;
    const unsigned long long vmask = n_valid == 64u ? ~0ull : (1ull << n_valid) - 1ull;
    const unsigned long long zall = ~F & vmask;
    if (zall == 0ull) carry ^= n_valid & 1u;
    else carry = (n_valid - 1u - (63u - (uint32_t)__clzll((long long)zall))) & 1u;
    return lane < n_valid && parity == 0u;
}

template <bool TREES, bool COUNT, bool STAGE_TREE>
__global__ __launch_bounds__(CV_WAVES_PER_BLOCK * 64) void cv_profile_pass(CvArgs a) {
    __shared__ CvQEntry lds_q[CV_WAVES_PER_BLOCK * CV_QCAP];
    __shared__ unsigned long long lds_words[CV_WAVES_PER_BLOCK][2][STAGE_TREE ? CV_TREE_SEG_GROUPS : 1];   // stage trees: verdict bits of a row segment
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CvQEntry* q = lds_q + wib * CV_QCAP;
    const uint32_t rank = blockIdx.x * CV_WAVES_PER_BLOCK + wib;
    (void)rank;
    kptr<CvScaleDev> scales = as_k(a.scales);
    kptr<UnitDev> rows = as_k(a.rows);
    kptr<StageDev> stages = as_k(a.stages);
    const uint32_t frame_bytes4 = a.frame_elems * 4u;
    const rsrc_t img = make_rsrc(a.sum, a.n_frames * frame_bytes4);
    const rsrc_t timg = make_rsrc(a.tilted != nullptr ? a.tilted : a.sum, a.n_frames * frame_bytes4);
    const uint32_t total = a.n_rows * a.n_frames;

    // Blocks are dealt round-robin over the 8 XCDs (observed placement; speed only): the waves that share an XCD — and its
    // 4 MiB L2 — walk one contiguous eighth of the (frame, scale, row) list, i.e. neighbouring rows of the same frame's
    // integral images at the same time, instead of rows of every frame in flight (fabric traffic of this kernel: 1.6 TB
    // per launch on 64 x 1080p with the strided assignment).
    uint32_t u_first = rank, u_end = total, u_step = a.total_waves;
    if (gridDim.x >= 8u) {
        const uint32_t xcd = blockIdx.x & 7u;
        const uint32_t u_begin = (uint32_t)((unsigned long long)total * xcd / 8u);
        u_end = (uint32_t)((unsigned long long)total * (xcd + 1u) / 8u);
        u_step = ((gridDim.x - xcd + 7u) >> 3) * CV_WAVES_PER_BLOCK;
        u_first = u_begin + (blockIdx.x >> 3) * CV_WAVES_PER_BLOCK + wib;
    }
    for (uint32_t u = u_first; u < u_end; u += u_step) {
        const uint32_t frame = u / a.n_rows;
        const uint32_t r = u - frame * a.n_rows;
        const uint32_t slot = rows[r].scale, iy = rows[r].first;
        const double ystep = scales[slot].ystep, inv_area = scales[slot].inv_area;
        const uint32_t win_w = scales[slot].win_w, win_h = scales[slot].win_h, end_x = scales[slot].end_x;
        const uint32_t q0 = scales[slot].q0, q1 = scales[slot].q1, q2 = scales[slot].q2, q3 = scales[slot].q3;
        kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + scales[slot].table_first;
        const rsrc_t sq_f = make_rsrc(a.sqsum + (size_t)frame * a.frame_elems, frame_bytes4 * 2u);
        const uint32_t frame_bytes = frame * frame_bytes4;
        const uint32_t y = (uint32_t)cv_round((double)iy * ystep);
        const bool row_border = y + win_h >= a.sum_h;          // pt.y + height >= sum.height -> -1 (tempcv.cpp:817-820)
        const double thr0 = (double)stages[0].threshold;
        uint32_t carry = 0;   // parity of the run of rejects that ends at the last position seen
        uint32_t n_q = 0;
        if (STAGE_TREE && !COUNT) {
            // Stage tree, every grid position (the skip rule needs the whole tree's verdict everywhere), but not every
            // position in lockstep to the end: the linear prefix runs on the dense groups; who is still inside the tree
            // afterwards (a few percent) waits in the wave's queue with its target stage, and a FULL chunk of those
            // sweeps the rest of the tree at a time — 64 lanes of survivors pay for the chains instead of every group
            // of 64 positions paying for its deepest lane.  The verdict bits of a row segment (reject / accept words)
            // sit in LDS until every position of the segment has one; then the walk is resolved group by group.
            unsigned long long* Fw = lds_words[wib][0];
            unsigned long long* Aw = lds_words[wib][1];
            uint32_t prefix = 0;   // leading stages of the sweep order that reject outright and pass to the next one
            while (prefix + 1u < a.n_order && stages[stages[prefix].order].on_fail == -2 &&
                   stages[stages[prefix].order].on_pass == (int32_t)stages[prefix + 1u].order)
                ++prefix;
            CvQEntry* fail_g = reinterpret_cast<CvQEntry*>(a.fail_scratch) + (size_t)rank * CV_QCAP;
            auto drain = [&](uint32_t seg0) {
                if (a.chains.n != 0u) {
                    // the tree is made of chains: compacting sweeps (every queued window sits at the first stage of chain 0)
                    cv_chain_sweep<TREES>(a.chains, stages, img, timg, a.tilted == nullptr, a.table + (size_t)scales[slot].table_first * 16u, table, q, n_q,
                                          fail_g, reinterpret_cast<unsigned long long*>(q + CV_TAIL_MAX), lane,
                                          [&](const CvQEntry& e, bool mine, bool accepted) {
                                              if (mine) {
                                                  const uint32_t rel = (e.xy & 0xffffu) - seg0;
                                                  atomicOr((accepted ? Aw : Fw) + (rel >> 6), 1ull << (rel & 63u));
                                              }
                                          });
                    n_q = 0;
                    __builtin_amdgcn_wave_barrier();
                    return;
                }
                for (uint32_t base = 0; base < n_q; base += 64u) {
                    const bool act = base + lane < n_q;
                    const CvQEntry e = q[act ? base + lane : 0u];
                    int32_t ptr = act ? (int32_t)(e.xy >> 16) : -3;
                    for (uint32_t oi = prefix; oi < a.n_order; ++oi) {
                        const uint32_t s = stages[oi].order;
                        const bool here = ptr == (int32_t)s;
                        if (__ballot(here) == 0ull) continue;
                        if (here) {
                            const bool pass = cv_stage_sum<TREES, false>(img, timg, table + stages[s].first_node, stages[s].n_nodes, e.off, e.vnf) >=
                                              (double)stages[s].threshold;
                            ptr = pass ? stages[s].on_pass : stages[s].on_fail;
                        }
                    }
                    if (act) {
                        const uint32_t rel = (e.xy & 0xffffu) - seg0;
                        if (ptr == -2) atomicOr(Fw + (rel >> 6), 1ull << (rel & 63u));
                        else if (ptr == -1) atomicOr(Aw + (rel >> 6), 1ull << (rel & 63u));
                    }
                }
                n_q = 0;
                __builtin_amdgcn_wave_barrier();
            };
            for (uint32_t seg0 = 0; seg0 < end_x; seg0 += CV_TREE_SEG_GROUPS * 64u) {
                const uint32_t seg_end = min(end_x, seg0 + CV_TREE_SEG_GROUPS * 64u);
                for (uint32_t ix0 = seg0; ix0 < seg_end; ix0 += 64u) {
                    const uint32_t ix = ix0 + lane;
                    const bool valid = ix < end_x;
                    const uint32_t x = (uint32_t)cv_round((double)(valid ? ix : 0u) * ystep);
                    const bool border = row_border || x + win_w >= a.stride;
                    const uint32_t po = y * a.stride + x;
                    const uint32_t off = frame_bytes + po * 4u;
                    double vnf = 1.0;
                    const bool eval = valid && !border;
                    if (eval) {
                        const int32_t isum = (int32_t)(ld_u32(img, off, q0 * 4u) - ld_u32(img, off, q1 * 4u) - ld_u32(img, off, q2 * 4u) +
                                                       ld_u32(img, off, q3 * 4u));
                        const uint64_t qq = ld_u64(sq_f, po * 8u, q0 * 8u) - ld_u64(sq_f, po * 8u, q1 * 8u) - ld_u64(sq_f, po * 8u, q2 * 8u) +
                                            ld_u64(sq_f, po * 8u, q3 * 8u);
                        const double mean = (double)isum * inv_area;
                        vnf = (double)qq;
                        vnf = vnf * inv_area - mean * mean;
                        vnf = vnf >= 0.0 ? sqrt(vnf) : 1.0;
                    }
                    int32_t ptr = eval ? (int32_t)stages[0].order : -3;   // -1 accepted, -2 rejected, -3 not evaluated
                    for (uint32_t oi = 0; oi < prefix; ++oi) {
                        const uint32_t s = stages[oi].order;
                        const bool here = ptr == (int32_t)s;
                        if (__ballot(here) == 0ull) break;   // (nobody left in the prefix)
                        if (here) {
                            const bool pass = cv_stage_sum<TREES, false>(img, timg, table + stages[s].first_node, stages[s].n_nodes, off, vnf) >=
                                              (double)stages[s].threshold;
                            ptr = pass ? stages[s].on_pass : stages[s].on_fail;
                        }
                    }
                    const unsigned long long fm = __ballot(ptr == -2), am = __ballot(ptr == -1), wm = __ballot(ptr >= 0);
                    if (lane == 0) {
                        Fw[(ix0 - seg0) >> 6] = fm;
                        Aw[(ix0 - seg0) >> 6] = am;
                    }
                    if (ptr >= 0) q[n_q + mbcnt(wm)] = CvQEntry{off, ix | ((uint32_t)ptr << 16), vnf};
                    n_q += (uint32_t)__popcll(wm);
                    __builtin_amdgcn_wave_barrier();
                    if (n_q > (uint32_t)CV_QCAP - 64u) drain(seg0);
                }
                drain(seg0);
                // every position of the segment has its verdict: the sequential walk, group by group
                for (uint32_t ix0 = seg0; ix0 < seg_end; ix0 += 64u) {
                    const uint32_t ix = ix0 + lane;
                    const unsigned long long F = Fw[(ix0 - seg0) >> 6], A = Aw[(ix0 - seg0) >> 6];
                    const bool visited = cv_visited(F, lane, min(64u, end_x - ix0), carry);
                    const bool hit = visited && ((A >> lane) & 1ull) != 0ull;
                    const unsigned long long hm = __ballot(hit);
                    if (hm != 0ull) {
                        uint32_t g = 0;
                        if (lane == 0) g = atomicAdd(a.det_count, (uint32_t)__popcll(hm));
                        g = __builtin_amdgcn_readfirstlane(g);
                        const uint32_t pos = g + mbcnt(hm);
                        if (hit && pos < a.det_cap) a.det[pos] = CvDet{(uint32_t)cv_round((double)ix * ystep), y, slot, frame};
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }
        for (uint32_t ix0 = 0; ix0 < end_x; ix0 += 64u) {
            const uint32_t ix = ix0 + lane;
            const bool valid = ix < end_x;
            const uint32_t x = (uint32_t)cv_round((double)(valid ? ix : 0u) * ystep);
            const bool border = row_border || x + win_w >= a.stride;
            const uint32_t po = y * a.stride + x;
            const uint32_t off = frame_bytes + po * 4u;
            double vnf = 1.0;
            const bool eval = valid && !border;
            if (eval) {
                const int32_t isum = (int32_t)(ld_u32(img, off, q0 * 4u) - ld_u32(img, off, q1 * 4u) - ld_u32(img, off, q2 * 4u) +
                                               ld_u32(img, off, q3 * 4u));
                const uint64_t qq = ld_u64(sq_f, po * 8u, q0 * 8u) - ld_u64(sq_f, po * 8u, q1 * 8u) - ld_u64(sq_f, po * 8u, q2 * 8u) +
                                    ld_u64(sq_f, po * 8u, q3 * 8u);
                const double mean = (double)isum * inv_area;
                vnf = (double)qq;
                vnf = vnf * inv_area - mean * mean;
                vnf = vnf >= 0.0 ? sqrt(vnf) : 1.0;
            }
            const uint32_t n_valid = min(64u, end_x - ix0);
            if (STAGE_TREE) {
                // the whole stage tree for every grid position (tempcv.cpp:834-861): every queued lane carries the
                // stage it visits next; stages are swept once in a topological order of the pass / fail graph
                int32_t ptr = eval ? (int32_t)stages[0].order : -3;   // -1 accepted, -2 rejected, -3 not evaluated
                unsigned long long entered = 0ull;
                for (uint32_t oi = 0; oi < a.n_order; ++oi) {
                    const uint32_t s = stages[oi].order;
                    const bool here = ptr == (int32_t)s;
                    if (__ballot(here) == 0ull) continue;
                    if (here) {
                        const bool pass = cv_stage_sum<TREES, false>(img, timg, table + stages[s].first_node, stages[s].n_nodes, off, vnf) >=
                                          (double)stages[s].threshold;
                        ptr = pass ? stages[s].on_pass : stages[s].on_fail;
                        entered |= 1ull << s;
                    }
                }
                const unsigned long long F = __ballot(ptr == -2);
                const bool visited = cv_visited(F, lane, n_valid, carry);
                if (COUNT) {
                    const unsigned long long vm = __ballot(visited);
                    if (lane == 0) atomicAdd(a.stage_entered + VJ_MAX_STAGES_DEV, (unsigned long long)__popcll(vm));
                    for (uint32_t s = 0; s < a.n_stages; ++s) {
                        const unsigned long long em = __ballot(visited && ((entered >> s) & 1ull) != 0ull);
                        if (lane == 0 && em != 0ull) atomicAdd(a.stage_entered + s, (unsigned long long)__popcll(em));
                    }
                }
                const unsigned long long am = __ballot(visited && ptr == -1);
                if (am != 0ull) {
                    uint32_t g = 0;
                    if (lane == 0) g = atomicAdd(a.det_count, (uint32_t)__popcll(am));
                    g = __builtin_amdgcn_readfirstlane(g);
                    const uint32_t pos = g + mbcnt(am);
                    if (visited && ptr == -1 && pos < a.det_cap) a.det[pos] = CvDet{x, y, slot, frame};
                }
                continue;
            }
            bool fail0 = false;
            if (eval)
                fail0 = !(cv_stage_sum_mode<TREES>(img, timg, table + stages[0].first_node, stages[0].n_nodes, off, vnf, stages[0].cv_f64, a.tree2) >= thr0);
            // which positions does the sequential walk visit?  parity of the reject run below each lane
            const unsigned long long F = __ballot(fail0);
            const bool visited = cv_visited(F, lane, n_valid, carry);
            const bool pass0 = visited && !border && !fail0;
            if (COUNT) {
                const unsigned long long vm = __ballot(visited), em = __ballot(visited && !border);
                if (lane == 0) {
                    atomicAdd(a.stage_entered + VJ_MAX_STAGES_DEV, (unsigned long long)__popcll(vm));
                    atomicAdd(a.stage_entered + 0, (unsigned long long)__popcll(em));
                }
            }
            const unsigned long long pm = __ballot(pass0);
            if (pass0) q[n_q + mbcnt(pm)] = CvQEntry{off, x | (y << 16), vnf};
            n_q += (uint32_t)__popcll(pm);
            __builtin_amdgcn_wave_barrier();
            if (n_q > (uint32_t)CV_QCAP - 64u) cv_flush<TREES, COUNT>(a, img, timg, table, q, n_q, slot, frame, lane);
        }
        if (!STAGE_TREE && n_q != 0u) cv_flush<TREES, COUNT>(a, img, timg, table, q, n_q, slot, frame, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool TREES, bool STAGE_TREE>
static void cv_launch(const CvArgs& a, bool count, dim3 g, dim3 b, hipStream_t stream) {
    if (count) hipLaunchKernelGGL((cv_profile_pass<TREES, true, STAGE_TREE>), g, b, 0, stream, a);
    else       hipLaunchKernelGGL((cv_profile_pass<TREES, false, STAGE_TREE>), g, b, 0, stream, a);
}

int launch_cv_profile_pass(const CvArgs& a, bool trees, bool count, bool stage_tree, int n_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    dim3 g(n_blocks), b(CV_WAVES_PER_BLOCK * 64);
    if (stage_tree) {
        if (trees) cv_launch<true, true>(a, count, g, b, stream);
        else       cv_launch<false, true>(a, count, g, b, stream);
    } else {
        if (trees) cv_launch<true, false>(a, count, g, b, stream);
        else       cv_launch<false, false>(a, count, g, b, stream);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------ stage trees whose prefix ran on LDS tiles
// cv_tile_pass<2> (vj_cv_tile.hip) evaluated the tree's linear prefix on every grid window of the small scales and queued
// the survivors.  cv_tree_walk takes them the rest of the way (tempcv.cpp:834-861: pass -> child, fail -> the next
// sibling up the tree, else reject; the stages are swept once in topological order, every lane carrying the stage it
// visits next) and sets each window's reject or accept bit; after skip_resolve has turned the reject bits of every
// window row into visited bits, cv_tree_emit reports the accepted windows the sequential walk visits.
__global__ __launch_bounds__(256) void cv_tree_walk(CvTreeArgs a) {
    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    kptr<StageDev> stages = as_k(a.stages);
    kptr<CvScaleDev> scales = as_k(a.scales);
    const rsrc_t img = make_rsrc(a.sum, a.sum_bytes);
    const uint32_t n = min(*a.tq_count, a.tq_cap);
    for (uint32_t base = wave * 64u; base < n; base += n_waves * 64u) {
        const bool act = base + lane < n;
        const CvTreeEntry e = a.tq[act ? base + lane : base];
        const uint32_t my_slot = e.bit_slot >> 8;
        int32_t ptr = act ? stages[stages[a.prefix - 1u].order].on_pass : -3;   // what passing the prefix's last stage leads to
        unsigned long long todo = __ballot(act);
        while (todo != 0ull) {   // the entries of one scale at a time: the feature table must be wave-uniform
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)my_slot, (int)__builtin_ctzll(todo));
            const bool mine = act && my_slot == slot;
            todo &= ~__ballot(mine);
            kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + scales[slot].table_first;
            for (uint32_t oi = a.prefix; oi < a.n_order; ++oi) {
                const uint32_t s = stages[oi].order;
                const bool here = mine && ptr == (int32_t)s;
                if (__ballot(here) == 0ull) continue;
                if (here) {
                    const bool pass = cv_stage_sum<false, false>(img, img, table + stages[s].first_node, stages[s].n_nodes, e.off, e.vnf) >=
                                      (double)stages[s].threshold;
                    ptr = pass ? stages[s].on_pass : stages[s].on_fail;
                }
            }
        }
        if (act) {
            const unsigned long long bit = 1ull << (e.bit_slot & 63u);
            if (ptr == -1) atomicOr(a.accept + e.word, bit);
            else atomicOr(a.reject + e.word, bit);   // -2, or a stage the sweep order never reaches: rejected
        }
    }
}

__global__ __launch_bounds__(256) void cv_tree_emit(CvTreeArgs a) {
    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    kptr<UnitDev> segs = as_k(a.segs);
    kptr<CvScaleDev> scales = as_k(a.scales);
    const uint32_t total = a.n_segs * a.n_frames;
    for (uint32_t u = wave; u < total; u += n_waves) {
        const uint32_t frame = u / a.n_segs, r = u - frame * a.n_segs;
        const uint32_t slot = segs[r].scale, first = segs[r].first, wpr = segs[r].count;
        const uint32_t iy = (first - scales[slot].bits_base) / wpr;
        const double ystep = scales[slot].ystep;
        const uint32_t y = (uint32_t)cv_round((double)iy * ystep);
        const size_t w0 = (size_t)frame * a.bits_frame_words + first;
        for (uint32_t w = 0; w < wpr; ++w) {
            const unsigned long long hits = a.accept[w0 + w] & a.reject[w0 + w];   // (reject holds the VISITED bits by now)
            if (hits == 0ull) continue;   // uniform
            const bool hit = ((hits >> lane) & 1ull) != 0ull;
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(a.det_count, (uint32_t)__popcll(hits));
            g = __builtin_amdgcn_readfirstlane(g);
            const uint32_t pos = g + mbcnt(hits);
            if (hit && pos < a.det_cap) a.det[pos] = CvDet{(uint32_t)cv_round((double)(w * 64u + lane) * ystep), y, slot, frame};
        }
    }
}

// The same for a tree made of chains (CvTreeArgs::chains): one sub-queue per scale, so a wave draws chunks of CV_TQ_CHUNK windows
// of ONE scale and sweeps the chains over them like a linear cascade (cv_chain_sweep).
__global__ __launch_bounds__(256) void cv_tree_chain_pass(CvTreeArgs a) {
    __shared__ CvTreeEntry lds_q[4 * CV_TQ_CHUNK];
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CvTreeEntry* q = lds_q + wib * CV_TQ_CHUNK;
    CvTreeEntry* fail_g = reinterpret_cast<CvTreeEntry*>(a.fail_scratch) + (size_t)(blockIdx.x * 4u + wib) * CV_TQ_CHUNK;
    kptr<StageDev> stages = as_k(a.stages);
    kptr<CvScaleDev> scales = as_k(a.scales);
    kptr<uint32_t> counts = as_k(a.tq_count);
    const rsrc_t img = make_rsrc(a.sum, a.sum_bytes);
    while (true) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(a.ticket, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        // ticket -> (scale, chunk): the tile scales in order
        uint32_t slot = 0, n_here = 0;
        uint64_t first = 0;
        for (; slot < a.n_scales; ++slot) {
            if (scales[slot].tile_th == 0u) continue;
            const uint64_t cap = cv_tq_cap(scales[slot].end_x, scales[slot].end_y, a.n_frames, a.tq_shift);
            const uint32_t cnt = (uint32_t)min((uint64_t)counts[scales[slot].tq_slot], cap);
            const uint32_t n_chunks = (cnt + a.chunk - 1u) / a.chunk;
            if (t < n_chunks) {
                first = cv_tq_first(scales[slot].tq_win_first, scales[slot].tq_slot, a.n_frames, a.tq_shift) + (uint64_t)t * a.chunk;
                n_here = min(a.chunk, cnt - t * a.chunk);
                break;
            }
            t -= n_chunks;
        }
        if (slot == a.n_scales) return;   // every chunk is taken
        for (uint32_t i = lane; i < n_here; i += 64u) q[i] = a.tq[first + i];
        __builtin_amdgcn_wave_barrier();
        kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + scales[slot].table_first;
        cv_chain_sweep<false>(a.chains, stages, img, img, true, a.table + (size_t)scales[slot].table_first * 16u, table, q, n_here, fail_g,
                              reinterpret_cast<unsigned long long*>(q + CV_TAIL_MAX), lane, [&](const CvTreeEntry& e, bool mine, bool accepted) {
                                  if (mine) atomicOr((accepted ? a.accept : a.reject) + e.word, 1ull << (e.bit_slot & 63u));
                              });
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_cv_tree_chain_pass(const CvTreeArgs& a, int n_blocks, void* stream_) {
    hipLaunchKernelGGL(cv_tree_chain_pass, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

int launch_cv_tree_walk(const CvTreeArgs& a, int n_blocks, void* stream_) {
    hipLaunchKernelGGL(cv_tree_walk, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}
int launch_cv_tree_emit(const CvTreeArgs& a, int n_blocks, void* stream_) {
    hipLaunchKernelGGL(cv_tree_emit, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------ tilted integral
// cvIntegral's tilted sum (OpenCV 2.4.2 imgproc; tilted(X, Y) = sum of gray(x, y) over y < Y, |x - X + 1| <= Y - y - 1)
// by its row recurrence
//   T[Y][X] = T[Y-1][X-1] + T[Y-1][X+1] - T[Y-2][X] + I(X-1, Y-1) + I(X-1, Y-2),
//   T[Y][-1] = T[Y-1][0],  T[Y][W+1] = T[Y-1][W]   (a triangle whose apex lies outside the image),
// exact in 32 bits modulo 2^32 like CV_32S.  Rows depend on the two rows above, columns do not depend on each other:
// one workgroup per frame walks the rows with the last three rows in LDS, 1024 columns at a time.  Only cascades with
// tilted features in the OpenCV profile ask for it; it is not on the headline path.
__device__ __forceinline__ uint32_t gray_at(const uint8_t* row, uint32_t x, uint32_t ch) {
    if (ch <= 1u) return row[x];
    const uint8_t* p = row + (size_t)x * ch;
    return (p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + 8192u) >> 14;   // OpenCV's 8-bit BGR2GRAY, as the integral kernels
}

__global__ __launch_bounds__(1024) void tilted_rows(TiltedArgs a) {
    extern __shared__ uint32_t lds_rows[];   // 3 rows of (W + 1)
    const uint32_t frame = blockIdx.x;
    const uint32_t ow = a.width + 1u;
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    uint32_t* out = a.tilted + (size_t)frame * a.frame_elems;
    for (uint32_t x = threadIdx.x; x < ow; x += 1024u) {
        lds_rows[x] = 0u;   // row 0
        out[x] = 0u;
    }
    __syncthreads();
    for (uint32_t Y = 1; Y <= a.height; ++Y) {
        const uint32_t* t1 = lds_rows + ((Y - 1u) % 3u) * ow;
        const uint32_t* t2 = lds_rows + ((Y + 1u) % 3u) * ow;   // (Y - 2) mod 3
        uint32_t* cur = lds_rows + (Y % 3u) * ow;
        const uint8_t* i1 = img + (size_t)(Y - 1u) * a.gray_stride;
        const uint8_t* i2 = img + (size_t)(Y >= 2u ? Y - 2u : 0u) * a.gray_stride;
        const bool has2 = Y >= 2u;
        for (uint32_t X = threadIdx.x; X < ow; X += 1024u) {
            const uint32_t left = X >= 1u ? t1[X - 1u] : (has2 ? t2[0] : 0u);
            const uint32_t right = X + 1u < ow ? t1[X + 1u] : (has2 ? t2[a.width] : 0u);
            const uint32_t up2 = has2 ? t2[X] : 0u;
            uint32_t px = 0u;
            if (X >= 1u) px = gray_at(i1, X - 1u, a.channels) + (has2 ? gray_at(i2, X - 1u, a.channels) : 0u);
            const uint32_t v = left + right - up2 + px;
            cur[X] = v;
            out[(size_t)Y * ow + X] = v;
        }
        __syncthreads();
    }
}

int launch_tilted_integral(const TiltedArgs& a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const size_t lds = (size_t)3 * (a.width + 1u) * sizeof(uint32_t);
    if (lds > 160u * 1024u) return (int)hipErrorInvalidValue;
    if (lds > 64u * 1024u) {
        const hipError_t e = hipFuncSetAttribute((const void*)tilted_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(tilted_rows, dim3(a.n_frames), dim3(1024), lds, stream, a);
    return (int)hipGetLastError();
}

// The same image without the row-by-row dependency (1080 barriers deep: 1.25 ms whatever the batch).  With zero outside the frame,
//   A(x, y) = sum_k I(x - k, y - k)   (the "\" diagonal ending at (x, y)),    B(x, y) = sum_k I(x + k, y - k)   (the "/" diagonal),
// the triangle with its apex at pixel (x, y) is the one with its apex at (x, y - 1) plus both diagonals through (x, y):
//   Tri(x, y) = Tri(x, y - 1) + A(x, y) + B(x, y) - I(x, y),        tilted(X, Y) = Tri(X - 1, Y - 1)
// — a column prefix of C = A + B - I, and A and B are column prefixes of the image sheared one way and the other.  Three prefix sums
// over the rows, each in bands of 8 rows like the upright integral: per-band totals, an exclusive scan over the bands, the rows of a
// band from its prefix.  Integer adds mod 2^32 in another order: the same image, bit for bit (vj_integral_tilted's tests compare
// both kernels with the oracle's direct sum).  diag: [frame][band][2][W + H] (A by x - y + H - 1, B by x + y), col: [frame][band][W + 1].
constexpr uint32_t TILT_ROWS = 8;
__device__ __forceinline__ uint32_t tilt_px(const TiltedArgs& a, const uint8_t* img, int32_t x, int32_t y) {
    if (x < 0 || y < 0 || x >= (int32_t)a.width || y >= (int32_t)a.height) return 0u;
    return gray_at(img + (size_t)y * a.gray_stride, (uint32_t)x, a.channels);
}

__global__ __launch_bounds__(256) void tilt_diag_sums(TiltedArgs a, uint32_t* diag) {
    const uint32_t D = a.width + a.height, i = blockIdx.x * 256u + threadIdx.x, band = blockIdx.y, frame = blockIdx.z;
    if (i >= D) return;
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    const uint32_t n_bands = (a.height + TILT_ROWS - 1u) / TILT_ROWS;
    uint32_t sa = 0, sb = 0;
    for (uint32_t r = 0; r < TILT_ROWS; ++r) {
        const int32_t y = (int32_t)(band * TILT_ROWS + r);
        sa += tilt_px(a, img, (int32_t)i - (int32_t)(a.height - 1u) + y, y);
        sb += tilt_px(a, img, (int32_t)i - y, y);
    }
    uint32_t* d = diag + ((size_t)frame * n_bands + band) * 2u * D;
    d[i] = sa;
    d[D + i] = sb;
}

// exclusive prefix over the bands, in place: `n` values per band, `per_band` the distance between bands
__global__ __launch_bounds__(256) void tilt_band_scan(uint32_t* v, uint32_t n, uint32_t per_band, uint32_t n_bands) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, frame = blockIdx.y;
    if (i >= n) return;
    uint32_t* p = v + (size_t)frame * n_bands * per_band + i;
    uint32_t s = 0;
    for (uint32_t b0 = 0; b0 < n_bands; b0 += 8u) {   // eight bands' loads in flight per step
        uint32_t t[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) t[k] = b0 + k < n_bands ? p[(size_t)(b0 + k) * per_band] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            if (b0 + k < n_bands) p[(size_t)(b0 + k) * per_band] = s;
            s += t[k];
        }
    }
}

// C(xc, y) for the rows of one band: the diagonals' prefixes at the band's top edge plus their part inside the band
template <typename F>
__device__ __forceinline__ void tilt_band_rows(const TiltedArgs& a, const uint8_t* img, const uint32_t* d /* this band's diag block */, int32_t xc,
                                               uint32_t band, F row) {
    const uint32_t D = a.width + a.height;
    const int32_t y0 = (int32_t)(band * TILT_ROWS);
    for (uint32_t r = 0; r < TILT_ROWS && y0 + (int32_t)r < (int32_t)a.height; ++r) {
        const int32_t y = y0 + (int32_t)r;
        const int32_t ia = xc - y + (int32_t)(a.height - 1u), ib = xc + y;
        uint32_t A = ia >= 0 && ia < (int32_t)D ? d[ia] : 0u, B = ib >= 0 && ib < (int32_t)D ? d[D + ib] : 0u;
        for (int32_t k = 0; k <= (int32_t)r; ++k) {
            A += tilt_px(a, img, xc - k, y - k);
            B += tilt_px(a, img, xc + k, y - k);
        }
        row(y, A + B - tilt_px(a, img, xc, y));
    }
}

__global__ __launch_bounds__(256) void tilt_col_sums(TiltedArgs a, const uint32_t* diag, uint32_t* col) {
    const uint32_t ow = a.width + 1u, X = blockIdx.x * 256u + threadIdx.x, band = blockIdx.y, frame = blockIdx.z;
    if (X >= ow) return;
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    const uint32_t n_bands = (a.height + TILT_ROWS - 1u) / TILT_ROWS, D = a.width + a.height;
    uint32_t s = 0;
    tilt_band_rows(a, img, diag + ((size_t)frame * n_bands + band) * 2u * D, (int32_t)X - 1, band, [&](int32_t, uint32_t c) { s += c; });
    col[((size_t)frame * n_bands + band) * ow + X] = s;
}

__global__ __launch_bounds__(256) void tilt_out_rows(TiltedArgs a, const uint32_t* diag, const uint32_t* col) {
    const uint32_t ow = a.width + 1u, X = blockIdx.x * 256u + threadIdx.x, band = blockIdx.y, frame = blockIdx.z;
    if (X >= ow) return;
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    const uint32_t n_bands = (a.height + TILT_ROWS - 1u) / TILT_ROWS, D = a.width + a.height;
    uint32_t* out = a.tilted + (size_t)frame * a.frame_elems;
    if (band == 0u) out[X] = 0u;   // row 0
    uint32_t acc = col[((size_t)frame * n_bands + band) * ow + X];
    tilt_band_rows(a, img, diag + ((size_t)frame * n_bands + band) * 2u * D, (int32_t)X - 1, band, [&](int32_t y, uint32_t c) {
        acc += c;
        out[(size_t)(y + 1) * ow + X] = acc;   // tilted(X, Y = y + 1) = Tri(X - 1, y)
    });
}

int launch_tilted_bands(const TiltedArgs& a, uint32_t* diag, uint32_t* col, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t n_bands = (a.height + TILT_ROWS - 1u) / TILT_ROWS, D = a.width + a.height, ow = a.width + 1u;
    hipLaunchKernelGGL(tilt_diag_sums, dim3((D + 255u) / 256u, n_bands, a.n_frames), dim3(256), 0, stream, a, diag);
    hipLaunchKernelGGL(tilt_band_scan, dim3((2u * D + 255u) / 256u, a.n_frames), dim3(256), 0, stream, diag, 2u * D, 2u * D, n_bands);
    hipLaunchKernelGGL(tilt_col_sums, dim3((ow + 255u) / 256u, n_bands, a.n_frames), dim3(256), 0, stream, a, (const uint32_t*)diag, col);
    hipLaunchKernelGGL(tilt_band_scan, dim3((ow + 255u) / 256u, a.n_frames), dim3(256), 0, stream, col, ow, ow, n_bands);
    hipLaunchKernelGGL(tilt_out_rows, dim3((ow + 255u) / 256u, n_bands, a.n_frames), dim3(256), 0, stream, a, (const uint32_t*)diag, (const uint32_t*)col);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------ gray image
// clifGrayscale (clif.h:55-58 -> cvCvtColor BGR2GRAY): the gray image the integral kernels see, as its own output.
__global__ __launch_bounds__(256) void gray_rows(TiltedArgs a, uint8_t* dst, uint32_t dst_stride) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
    if (x >= a.width) return;
    dst[(size_t)y * dst_stride + x] = (uint8_t)gray_at(a.gray + (size_t)y * a.gray_stride, x, a.channels);
}

int launch_grayscale(const TiltedArgs& a, uint8_t* dst, uint32_t dst_stride, void* stream_) {
    hipLaunchKernelGGL(gray_rows, dim3((a.width + 255u) / 256u, a.height), dim3(256), 0, (hipStream_t)stream_, a, dst, dst_stride);
    return (int)hipGetLastError();
}

}  // namespace vj
