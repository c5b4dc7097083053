// OpenCV arithmetic profile on LDS tiles (gfx950): the small scales of cvHaarDetectObjects' scale-cascade path
// (tempcv.cpp:1116-1185 -> cvRunHaarClassifierCascadeSum :795-972; stumps and two-node trees, linear stages and stage-tree prefixes,
// upright features and — the tilted integral staged behind the sum — tilted ones)
// with the rectangle corners gathered from a tile of the sum image staged in LDS instead of through the
// texture-address unit — the machinery of the clod profile's tile kernel (vj_kernels.hip: cascade_tile_pass) on this
// profile's arithmetic: f64 variance norm factor and stage sums, node products per stage as tempcv.cpp's scalar
// branches write them (cv_node terms below), threshold bias (host), border rule (:817-820).  The large scales stay on
// cv_profile_pass (vj_cv_profile.hip) and run concurrently on a second stream.  MUST be compiled with -ffp-contract=off.
//
// The sequential "ixstep = result != 0 ? 1 : 2" rule (:1163) makes the set of evaluated windows data-dependent: window
// i of a row is visited iff the run of stage-0 rejects that ends at i-1 has even length.  Two launches per LDS class:
//   cv_tile_pass<0>  stage 0 for EVERY grid window of the tile (dense, 3-20 stumps) -> one reject bit per window,
//   skip_resolve     (vj_kernels.hip, shared with the clod profile's CPU-variant window sets) reject bits -> visited bits,
//   cv_tile_pass<1>  the whole cascade on the visited windows: dense multi-chunk sweeps with the tile's survivors
//                    re-packed across the waves, then the wave-split finish once few are left.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "vj_device.hpp"
#include "vj_devutil.hpp"

namespace vj {

__device__ __forceinline__ void cvt_barrier() { __syncthreads(); }
constexpr uint32_t CVT_OFF_MASK = 0xfffffu;   // a queue entry: tile-local byte offset (< 2^18) | window index inside the tile << 20 (mode 2)
__device__ __forceinline__ int cvt_round(double v) { return __double2int_rn(v); }   // cvRound: half to even

// CvNodeRec as 16 scalar dwords: lt[3] 0-2, da[3] 3-5, db[3] 6-8, w[3] 9-11, thr 12, left 13, right 14, flags 15
__device__ __forceinline__ uint32_t cvt_ld(const char* img, uint32_t lane_off, uint32_t uni_off) {
    return *reinterpret_cast<const uint32_t*>(img + (lane_off + uni_off));
}

// One stump stage on NC chunks of 64 windows (lane l holds window l of every chunk): the record arrives once through the
// scalar cache, the gathers of the first two rectangles of all chunks are in flight together.  Per window:
//   F64 (a stage flagged two_rects, tempcv.cpp:872-888):  rect0 = (double)calc_sum * (double)w0, ..., sum = rect1 + rect0
//   else (:907-911):  sum = (double)((float)calc_sum * w0); sum += (double)((float)calc_sum1 * w1); [+ third]
//   stage_sum += sum < thr * vnf ? left : right                                   (alpha[sum >= t], :913)
template <int NC, bool F64>
__device__ __forceinline__ void cvt_stage_sum_multi(const char* img, kptr<NodeRecDev> tab, uint32_t n_nodes, const uint32_t (&off)[NC],
                                                    const double (&vnf)[NC], double (&stage_sum)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) stage_sum[c] = 0.0;
    NodeRecDev r = tab[0];
    for (uint32_t j = 0; j < n_nodes; ++j) {
        const NodeRecDev rn = tab[j + 1 < n_nodes ? j + 1 : j];
        const uint32_t lt0 = r[0], lt1 = r[1], lt2 = r[2], da0 = r[3], da1 = r[4], da2 = r[5], db0 = r[6], db1 = r[7], db2 = r[8];
        const float w0 = __uint_as_float(r[9]), w1 = __uint_as_float(r[10]), w2 = __uint_as_float(r[11]);
        const double thr = (double)__uint_as_float(r[12]);
        const double left = (double)__uint_as_float(r[13]), right = (double)__uint_as_float(r[14]);
        uint32_t c0[NC][4], c1[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            c0[c][0] = cvt_ld(img, off[c], lt0);
            c0[c][1] = cvt_ld(img, off[c], lt0 + da0);
            c0[c][2] = cvt_ld(img, off[c], lt0 + db0);
            c0[c][3] = cvt_ld(img, off[c], lt0 + da0 + db0);
            c1[c][0] = cvt_ld(img, off[c], lt1);
            c1[c][1] = cvt_ld(img, off[c], lt1 + da1);
            c1[c][2] = cvt_ld(img, off[c], lt1 + db1);
            c1[c][3] = cvt_ld(img, off[c], lt1 + da1 + db1);
        }
        double s[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int32_t r0 = (int32_t)(c0[c][0] - c0[c][1] - c0[c][2] + c0[c][3]);   // calc_sum: p0 - p1 - p2 + p3 (:118-121)
            const int32_t r1 = (int32_t)(c1[c][0] - c1[c][1] - c1[c][2] + c1[c][3]);
            if (F64) {
                const double rect0 = (double)r0 * (double)w0;
                const double rect1 = (double)r1 * (double)w1;
                s[c] = rect1 + rect0;
            } else {
                s[c] = (double)((float)r0 * w0);
                s[c] += (double)((float)r1 * w1);
            }
        }
        if (!F64 && w2 != 0.0f) {   // uniform (node->feature.rect[2].p0 != 0)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                c0[c][0] = cvt_ld(img, off[c], lt2);
                c0[c][1] = cvt_ld(img, off[c], lt2 + da2);
                c0[c][2] = cvt_ld(img, off[c], lt2 + db2);
                c0[c][3] = cvt_ld(img, off[c], lt2 + da2 + db2);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int32_t r2 = (int32_t)(c0[c][0] - c0[c][1] - c0[c][2] + c0[c][3]);
                s[c] += (double)((float)r2 * w2);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) stage_sum[c] += s[c] < thr * vnf[c] ? left : right;
        r = rn;
    }
}

template <int NC>
__device__ __forceinline__ void cvt_stage_sum_mode(const char* img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t f64,
                                                   const uint32_t (&off)[NC], const double (&vnf)[NC], double (&sum)[NC]) {
    if (f64 != 0u) cvt_stage_sum_multi<NC, true>(img, tab, n_nodes, off, vnf, sum);    // uniform per stage
    else cvt_stage_sum_multi<NC, false>(img, tab, n_nodes, off, vnf, sum);
}

// Two consecutive stumps on one window with all their gathers in flight (the wave-split finish: one chunk per wave, so
// the only parallelism a wave has is across stumps).  Returns the two node sums with the stage's arithmetic.
template <bool F64>
__device__ __forceinline__ void cvt_node_sum_pair(const char* img, const NodeRecDev& ra, const NodeRecDev& rb, uint32_t off, double& sa,
                                                  double& sb) {
    uint32_t v[2][3][4];
    const NodeRecDev* rr[2] = {&ra, &rb};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const NodeRecDev& r = *rr[p];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            v[p][q][0] = cvt_ld(img, off, r[q]);
            v[p][q][1] = cvt_ld(img, off, r[q] + r[3 + q]);
            v[p][q][2] = cvt_ld(img, off, r[q] + r[6 + q]);
            v[p][q][3] = cvt_ld(img, off, r[q] + r[3 + q] + r[6 + q]);
        }
    }
    double out[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const NodeRecDev& r = *rr[p];
        const int32_t r0 = (int32_t)(v[p][0][0] - v[p][0][1] - v[p][0][2] + v[p][0][3]);
        const int32_t r1 = (int32_t)(v[p][1][0] - v[p][1][1] - v[p][1][2] + v[p][1][3]);
        const float w0 = __uint_as_float(r[9]), w1 = __uint_as_float(r[10]);
        if (F64) {
            const double rect0 = (double)r0 * (double)w0;
            const double rect1 = (double)r1 * (double)w1;
            out[p] = rect1 + rect0;
        } else {
            out[p] = (double)((float)r0 * w0);
            out[p] += (double)((float)r1 * w1);
        }
    }
    if (!F64) {
        const float wa2 = __uint_as_float(ra[11]), wb2 = __uint_as_float(rb[11]);
        if (wa2 != 0.0f || wb2 != 0.0f) {   // uniform; an absent third rectangle has lt = da = db = 0: four reads of the origin
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const NodeRecDev& r = *rr[p];
                v[p][2][0] = cvt_ld(img, off, r[2]);
                v[p][2][1] = cvt_ld(img, off, r[2] + r[5]);
                v[p][2][2] = cvt_ld(img, off, r[2] + r[8]);
                v[p][2][3] = cvt_ld(img, off, r[2] + r[5] + r[8]);
            }
            if (wa2 != 0.0f) out[0] += (double)((float)(int32_t)(v[0][2][0] - v[0][2][1] - v[0][2][2] + v[0][2][3]) * wa2);
            if (wb2 != 0.0f) out[1] += (double)((float)(int32_t)(v[1][2][0] - v[1][2][1] - v[1][2][2] + v[1][2][3]) * wb2);
        }
    }
    sa = out[0];
    sb = out[1];
}

// A two-node tree (root r0, its only node child r1 — every tree of frontalface_alt2) on one window: icvEvalHidHaarClassifier's
// walk (tempcv.cpp:771-792: idx = sum < t ? left : right until idx <= 0; node sums are int * float products widened to
// double, :783-788) with both nodes' gathers in flight.  Returns the leaf value; `code` names the leaf for an ordered replay
// (bit 1: reached through the child, bit 0: right side).
__device__ __forceinline__ double cvt_tree2_value(const char* img, const NodeRecDev& r0, const NodeRecDev& r1, uint32_t off, double vnf,
                                                  uint32_t& code) {
    double s0, s1;
    cvt_node_sum_pair<false>(img, r0, r1, off, s0, s1);
    const uint32_t flags0 = r0[15];
    const bool left0 = s0 < (double)__uint_as_float(r0[12]) * vnf;
    const bool left1 = s1 < (double)__uint_as_float(r1[12]) * vnf;
    const bool to_child = left0 ? (flags0 & 1u) != 0u : (flags0 & 2u) != 0u;
    const float leaf0 = left0 ? __uint_as_float(r0[13]) : __uint_as_float(r0[14]);
    const float leaf1 = left1 ? __uint_as_float(r1[13]) : __uint_as_float(r1[14]);
    code = to_child ? (left1 ? 2u : 3u) : (left0 ? 0u : 1u);
    return (double)(to_child ? leaf1 : leaf0);
}

template <int NC>
__device__ __forceinline__ void cvt_stage_sum_tree2_multi(const char* img, kptr<NodeRecDev> tab, uint32_t n_trees, const uint32_t (&off)[NC],
                                                          const double (&vnf)[NC], double (&stage_sum)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) stage_sum[c] = 0.0;
    NodeRecDev r0 = tab[0], r1 = tab[1];
    for (uint32_t t = 0; t < n_trees; ++t) {
        const uint32_t tn = t + 1u < n_trees ? t + 1u : t;
        const NodeRecDev n0 = tab[2u * tn], n1 = tab[2u * tn + 1u];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint32_t code;
            stage_sum[c] += cvt_tree2_value(img, r0, r1, off[c], vnf[c], code);
        }
        r0 = n0;
        r1 = n1;
    }
}

// One stage on NC chunks with the cascade's shape: stumps (with the stage's product type) or two-node trees.
template <bool TREE2, int NC>
__device__ __forceinline__ void cvt_stage_sum_any(const char* img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t f64,
                                                  const uint32_t (&off)[NC], const double (&vnf)[NC], double (&sum)[NC]) {
    if (TREE2) cvt_stage_sum_tree2_multi<NC>(img, tab, n_nodes >> 1, off, vnf, sum);
    else cvt_stage_sum_mode<NC>(img, tab, n_nodes, f64, off, vnf, sum);
}

// Dense sweep of one stage over a wave's queue (qo / qv: tile-local byte offsets and norm factors), 4 / 2 / 1 chunks of 64
// windows per stump; survivors are rewritten in place at their ballot ranks.  Returns the survivors.
template <bool TREE2>
__device__ __forceinline__ uint32_t cvt_sweep(const char* img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t f64, double thr,
                                              uint32_t* qo, double* qv, uint32_t n, uint32_t lane) {
    uint32_t m = 0, base = 0;
    auto group = [&](auto nc_tag) {
        constexpr int NC = decltype(nc_tag)::value;
        uint32_t raw[NC], off[NC];
        double vnf[NC], sum[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t i = base + (uint32_t)c * 64u + lane;
            raw[c] = qo[i < n ? i : 0u];
            off[c] = raw[c] & CVT_OFF_MASK;   // (stage trees keep the window's index inside the tile above the offset)
            vnf[c] = qv[i < n ? i : 0u];
        }
        cvt_stage_sum_any<TREE2, NC>(img, tab, n_nodes, f64, off, vnf, sum);
        __builtin_amdgcn_wave_barrier();   // every entry of the group is in registers
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool pass = base + (uint32_t)c * 64u + lane < n && sum[c] >= thr;
            const unsigned long long mask = __ballot(pass);
            if (pass) {
                qo[m + mbcnt(mask)] = raw[c];
                qv[m + mbcnt(mask)] = vnf[c];
            }
            m += (uint32_t)__popcll(mask);
        }
        __builtin_amdgcn_wave_barrier();
        base += (uint32_t)NC * 64u;
    };
    while (base + 192u < n) group(std::integral_constant<int, 4>{});
    if (base + 64u < n) group(std::integral_constant<int, 2>{});
    if (base < n) group(std::integral_constant<int, 1>{});
    return m;
}

// Wave-split finish (tile_wave_split of the clod profile on this profile's arithmetic).  T <= CVT_WS_MAX packed survivors
// sit in lds_off / lds_vnf [0, T): c = ceil(T / 64) chunks.  The stage's stumps are split into K = 8 / c contiguous
// ranges; wave w evaluates range w / c on chunk w % c and leaves, per window, the f64 sum of its range and one verdict
// bit per stump.  The K range sums added together are the stage's leaf values in another association than the
// reference's running sum (stage_sum += alpha, in stump order, :913): the comparison with the stage threshold is only
// taken from it when it clears the threshold by more than `delta`, an a-priori bound on the difference between any two
// f64 summation orders of the stage's leaf values (4 n 2^-53 sum max|leaf|, from the host's f32 bound: sp_delta * 2^-28
// keeps a factor 2 in hand); windows inside the band replay their verdict bits in stump order.  Bit-identical results.
template <bool COUNT, bool TREE2>
__device__ __forceinline__ uint32_t cvt_wave_split(const CvTileArgs& a, const char* img, kptr<NodeRecDev> table, uint32_t* lds_off,
                                                   double* lds_vnf, uint32_t* lds_x, uint32_t* lds_cnt, uint32_t T, uint32_t pos0,
                                                   uint32_t lane, uint32_t wib) {
    kptr<StageDev> stages = as_k(a.stages);
    constexpr uint32_t XW = 384u;   // dwords per producing wave: 64 f64 range sums, then 4 x 64 verdict words
    for (uint32_t s = pos0; s < a.n_stages && T != 0u; ++s) {
        if (COUNT && threadIdx.x == 0) atomicAdd(a.stage_entered + s, (unsigned long long)T);
        // items of a stage: stumps, or two-node trees (records 2t and 2t + 1; a 2-bit leaf code per tree instead of a verdict bit)
        constexpr uint32_t PER_WORD = TREE2 ? 16u : 32u;
        const uint32_t n = TREE2 ? stages[s].n_nodes >> 1 : stages[s].n_nodes, f64 = stages[s].cv_f64;
        const double thr_s = (double)stages[s].threshold;
        const double delta = (double)stages[s].sp_delta * 3.725290298461914e-09;   // 2^-28
        kptr<NodeRecDev> tab = table + stages[s].first_node;
        const uint32_t c = (T + 63u) >> 6;
        uint32_t K = (uint32_t)CVT_WAVES / c;
        uint32_t rs = (n + K - 1u) / K;
        rs = (rs + 1u) & ~1u;                      // even: pairs never straddle two ranges
        if (rs > 4u * PER_WORD || K == 1u) { K = 1u; rs = n; }
        const uint32_t chunk = wib % c, range = wib / c;   // uniform
        const uint32_t i = chunk * 64u + lane;
        const bool valid = i < T;
        const uint32_t off = lds_off[valid ? i : 0u];
        const double vnf = lds_vnf[valid ? i : 0u];
        bool pass = false;
        if (range < K) {
            if (K == 1u) {
                const uint32_t off1[1] = {off};
                const double vnf1[1] = {vnf};
                double sum1[1];
                cvt_stage_sum_any<TREE2, 1>(img, tab, stages[s].n_nodes, f64, off1, vnf1, sum1);
                pass = valid && sum1[0] >= thr_s;
            } else {
                const uint32_t j0 = min(range * rs, n), j1 = min(j0 + rs, n);
                double psum = 0.0;
                uint32_t* xw = lds_x + wib * XW;
                for (uint32_t w0 = j0, wd = 0; TREE2 && w0 < j1; w0 += PER_WORD, ++wd) {
                    const uint32_t m = min(PER_WORD, j1 - w0);
                    uint32_t bw = 0u;
                    NodeRecDev r0 = tab[2u * w0], r1 = tab[2u * w0 + 1u];
                    for (uint32_t k = 0; k < m; ++k) {
                        const uint32_t tn = min(w0 + k + 1u, j1 - 1u);
                        const NodeRecDev n0 = tab[2u * tn], n1 = tab[2u * tn + 1u];
                        uint32_t code;
                        psum += cvt_tree2_value(img, r0, r1, off, vnf, code);
                        bw |= code << (2u * k);
                        r0 = n0;
                        r1 = n1;
                    }
                    xw[128u + wd * 64u + lane] = bw;
                }
                for (uint32_t w0 = j0, wd = 0; !TREE2 && w0 < j1; w0 += 32u, ++wd) {
                    const uint32_t m = min(32u, j1 - w0);
                    uint32_t bw = 0u, k = 0;
                    NodeRecDev ra = tab[w0], rb = tab[min(w0 + 1u, j1 - 1u)];
                    for (; k + 1u < m; k += 2u) {
                        const NodeRecDev na = tab[min(w0 + k + 2u, j1 - 1u)], nb = tab[min(w0 + k + 3u, j1 - 1u)];
                        double sa, sb;
                        if (f64 != 0u) cvt_node_sum_pair<true>(img, ra, rb, off, sa, sb);
                        else cvt_node_sum_pair<false>(img, ra, rb, off, sa, sb);
                        const bool la = sa < (double)__uint_as_float(ra[12]) * vnf, lb = sb < (double)__uint_as_float(rb[12]) * vnf;
                        psum += (double)(la ? __uint_as_float(ra[13]) : __uint_as_float(ra[14]));
                        psum += (double)(lb ? __uint_as_float(rb[13]) : __uint_as_float(rb[14]));
                        bw |= (la ? 0u : 1u) << k;
                        bw |= (lb ? 0u : 2u) << k;
                        ra = na;
                        rb = nb;
                    }
                    if (k < m) {   // odd tail (only the last word of the stage's last range)
                        double sa, sb;
                        if (f64 != 0u) cvt_node_sum_pair<true>(img, ra, ra, off, sa, sb);
                        else cvt_node_sum_pair<false>(img, ra, ra, off, sa, sb);
                        const bool la = sa < (double)__uint_as_float(ra[12]) * vnf;
                        psum += (double)(la ? __uint_as_float(ra[13]) : __uint_as_float(ra[14]));
                        bw |= (la ? 0u : 1u) << k;
                    }
                    xw[128u + wd * 64u + lane] = bw;
                }
                reinterpret_cast<double*>(xw)[lane] = psum;
            }
        }
        if (K > 1u) {
            cvt_barrier();   // every range sum and verdict word of the stage is in LDS
            if (wib < c) {   // range 0's wave decides its chunk
                double approx = 0.0;
                for (uint32_t r = 0; r < K; ++r) approx += reinterpret_cast<const double*>(lds_x + (r * c + wib) * XW)[lane];
                const double d = approx - thr_s;
                const bool clear = d > delta || d < -delta;
                pass = valid && d > delta;
                if (__ballot(valid && !clear) != 0ull) {
                    // replay in stump order; the leaf values come through the scalar cache
                    double sum = 0.0;
                    for (uint32_t r = 0; r < K; ++r) {
                        const uint32_t j0 = min(r * rs, n), j1 = min(j0 + rs, n);
                        const uint32_t* xw = lds_x + (r * c + wib) * XW + 128u;
                        for (uint32_t w0 = j0, wd = 0; w0 < j1; w0 += PER_WORD, ++wd) {
                            const uint32_t m = min(PER_WORD, j1 - w0);
                            const uint32_t bw = xw[wd * 64u + lane];
                            if (TREE2) {
                                kptr<uint32_t> lr = reinterpret_cast<kptr<uint32_t>>(tab + 2u * w0);
                                for (uint32_t k = 0; k < m; ++k) {
                                    const uint32_t code = (bw >> (2u * k)) & 3u;
                                    const float l0 = __uint_as_float(lr[k * 32u + 13u]), r0 = __uint_as_float(lr[k * 32u + 14u]);
                                    const float l1 = __uint_as_float(lr[k * 32u + 29u]), r1 = __uint_as_float(lr[k * 32u + 30u]);
                                    sum += (double)(code & 2u ? (code & 1u ? r1 : l1) : (code & 1u ? r0 : l0));
                                }
                            } else {
                                kptr<uint32_t> lr = reinterpret_cast<kptr<uint32_t>>(tab + w0);
                                for (uint32_t k = 0; k < m; ++k)
                                    sum += (double)((bw >> k) & 1u ? __uint_as_float(lr[k * 16u + 14u]) : __uint_as_float(lr[k * 16u + 13u]));
                            }
                        }
                    }
                    if (!clear) pass = valid && sum >= thr_s;
                }
            }
        }
        // survivors: compact the packed queue across the deciding waves
        const unsigned long long mask = __ballot(pass);
        if (lane == 0) lds_cnt[1u + wib] = (uint32_t)__popcll(mask);
        cvt_barrier();   // every entry is in registers, every wave's count is published
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)CVT_WAVES; ++w) {
            const uint32_t cw = lds_cnt[1u + w];
            before += w < wib ? cw : 0u;
            total += cw;
        }
        if (pass) {
            lds_off[before + mbcnt(mask)] = off;
            lds_vnf[before + mbcnt(mask)] = vnf;
        }
        T = __builtin_amdgcn_readfirstlane(total);
        cvt_barrier();   // the queue is repacked; lds_cnt and the scratch may be rewritten
    }
    return T;
}

// MODE 0: reject bits of stage 0 for every grid window of the tile.  MODE 1: the cascade on the visited windows.
// MODE 2 (stage trees; tempcv.cpp:834-861 returns 0 on ANY reject, so the skip rule needs the whole tree's verdict for
// every grid window): the tree's linear prefix — a.n_stages stages, 95 % of the rejects — on every grid window of the
// tile; a window's reject bit starts set and is cleared when it survives the prefix; the survivors go to a global queue
// for cv_tree_walk (vj_cv_profile.hip), which sets their reject or accept bit; skip_resolve and cv_tree_emit follow.
template <int MODE, bool COUNT, bool TREE2>
__global__ __launch_bounds__(CVT_WAVES * 64) void cv_tile_pass(CvTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
    uint32_t* lds_off = lds_dyn;                                                              // [CVT_WAVES][CVT_WAVE_CAP]
    double* lds_vnf = reinterpret_cast<double*>(lds_dyn + CVT_WAVES * CVT_WAVE_CAP);          // [CVT_WAVES][CVT_WAVE_CAP]
    uint32_t* lds_cnt = lds_dyn + CVT_WAVES * CVT_WAVE_CAP * 3;                               // 64 dwords
    unsigned long long* lds_F = reinterpret_cast<unsigned long long*>(lds_cnt + 64);           // mode 2: reject bits of the tile's <= 32 rows
    uint32_t* lds_img = lds_dyn + CVT_LDS_HEADER / 4;
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    kptr<CvScaleDev> scales = as_k(a.scales);
    kptr<StageDev> stages = as_k(a.stages);
    kptr<UnitDev> tiles = as_k(a.tiles);
    const uint32_t total_units = a.n_tiles * a.n_frames;
    const uint32_t frame_bytes4 = a.frame_elems * 4u;

    // tiles are handed out through eight ticket counters, one per contiguous part of the (frame, tile) list; workgroups
    // that share an XCD (blockIdx & 7 under the observed placement; speed only) start on the same part, so neighbouring
    // tiles of a frame are staged through the same L2 (as cascade_tile_pass does)
    const uint32_t my_xcd = blockIdx.x & 7u;
    auto part_begin = [&](uint32_t x) { return (uint32_t)((unsigned long long)total_units * x / 8u); };
    auto seeds = [&](uint32_t x) { return gridDim.x > x ? (gridDim.x - x + 7u) >> 3 : 0u; };
    uint32_t cur_part = my_xcd;
    auto draw = [&]() -> uint32_t {   // thread 0 only
        for (uint32_t tries = 0; tries < 8u; ++tries) {
            const uint32_t x = (cur_part + tries) & 7u;
            const uint32_t t = atomicAdd(a.ticket + x, 1u);
            const uint32_t cand = part_begin(x) + seeds(x) + t;
            if (cand < part_begin(x + 1u)) {
                cur_part = x;
                return cand;
            }
        }
        return total_units;
    };
    uint32_t u = part_begin(my_xcd) + (blockIdx.x >> 3);
    if (u >= part_begin(my_xcd + 1u)) {
        if (threadIdx.x == 0) lds_cnt[40] = draw();
        cvt_barrier();
        u = __builtin_amdgcn_readfirstlane(lds_cnt[40]);
    }
    while (u < total_units) {
        uint32_t next_u = 0;
        if (threadIdx.x == 0) next_u = draw();
        const uint32_t frame = u / a.n_tiles;
        const uint32_t r = u - frame * a.n_tiles;
        const uint32_t slot = tiles[r].scale;
        const uint32_t ix0 = tiles[r].first & 0xffffu, iy0 = tiles[r].first >> 16;
        const double ystep = scales[slot].ystep, inv_area = scales[slot].inv_area;
        const uint32_t win_w = scales[slot].win_w, win_h = scales[slot].win_h;
        const uint32_t end_x = scales[slot].end_x, end_y = scales[slot].end_y;
        const uint32_t tw = scales[slot].tile_tw, th = scales[slot].tile_th;
        const uint32_t pitch = scales[slot].tile_pitch, rows = scales[slot].tile_rows;
        // equRect in the tile's pitch (q0 = ex * stride + ex, q1 = q0 + ew, q2 = (ex + eh) * stride + ex: vj_cv.cpp)
        const uint32_t ex = scales[slot].q0 / (a.stride + 1u), ew = scales[slot].q1 - scales[slot].q0;
        const uint32_t eh = (scales[slot].q2 - scales[slot].q0) / a.stride;
        const uint32_t t0 = (ex * pitch + ex) * 4u, t1 = t0 + ew * 4u, t2 = t0 + eh * pitch * 4u, t3 = t2 + ew * 4u;
        const uint32_t q0 = scales[slot].q0, q1 = scales[slot].q1, q2 = scales[slot].q2, q3 = scales[slot].q3;
        const uint32_t wpr = (end_x + 63u) >> 6;
        const size_t frame_off = (size_t)frame * a.frame_elems;
        const rsrc_t sum_f = make_rsrc(a.sum + frame_off, frame_bytes4);
        const rsrc_t sq_f = make_rsrc(a.sqsum + frame_off, frame_bytes4 * 2u);
        unsigned long long* bits = a.bits + (size_t)frame * a.bits_frame_words + scales[slot].bits_base;
        const uint32_t x0 = (uint32_t)cvt_round((double)ix0 * ystep), y0 = (uint32_t)cvt_round((double)iy0 * ystep);

        cvt_barrier();   // the previous tile's gathers are finished
        // stage the tile's footprint of the sum image: 16 bytes per lane straight into LDS (buffer_load ... lds); the
        // host keeps the pitch a multiple of four dwords.  Reads past the frame return 0 (no window gathers them).
        for (uint32_t rr = wib; rr < rows; rr += (uint32_t)CVT_WAVES) {
            const uint32_t g_row = ((y0 + rr) * a.stride + x0) * 4u;   // uniform
            for (uint32_t c0 = 0; c0 < pitch; c0 += 256u) {
                const uint32_t soff = __builtin_amdgcn_readfirstlane(g_row + c0 * 4u);
                if (c0 + lane * 4u < pitch)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(sum_f, (__attribute__((address_space(3))) uint32_t*)(lds_img + rr * pitch + c0),
                                                             16, lane * 16u, soff, 0, 0);
            }
        }
        if (a.tilted != nullptr) {   // ... and of the tilted integral, right behind it (tilted nodes' records carry the distance)
            const rsrc_t tilt_f = make_rsrc(a.tilted + frame_off, frame_bytes4);
            for (uint32_t rr = wib; rr < rows; rr += (uint32_t)CVT_WAVES) {
                const uint32_t g_row = ((y0 + rr) * a.stride + x0) * 4u;
                for (uint32_t c0 = 0; c0 < pitch; c0 += 256u) {
                    const uint32_t soff = __builtin_amdgcn_readfirstlane(g_row + c0 * 4u);
                    if (c0 + lane * 4u < pitch)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(tilt_f, (__attribute__((address_space(3))) uint32_t*)(lds_img + (rows + rr) * pitch + c0),
                                                                 16, lane * 16u, soff, 0, 0);
                }
            }
        }
        // while the tile is in flight: this wave's four window rows (one per chunk, lane = column), the windows the walk
        // visits, and the four squared-sum corners of each (8-byte gathers from HBM / L2)
        constexpr int NCH = CVT_WAVE_CAP / 64;
        uint32_t w_off[NCH];
        uint64_t w_q[NCH];
        bool w_eval[NCH];
        uint32_t n_visited = 0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const uint32_t ty = wib * (uint32_t)NCH + (uint32_t)k;
            const uint32_t iy = iy0 + ty, ix = ix0 + lane;
            const bool valid = lane < tw && ty < th && ix < end_x && iy < end_y;
            const uint32_t x = (uint32_t)cvt_round((double)ix * ystep), y = (uint32_t)cvt_round((double)iy * ystep);
            const bool border = y + win_h >= a.sum_h || x + win_w >= a.stride;   // pt + real_window_size >= sum size -> -1 (:817-820)
            bool take = valid;
            if (MODE == 1 && ty < th && iy < end_y) {
                const unsigned long long V = bits[iy * wpr + (ix0 >> 6)];   // uniform: a tile row never straddles a word
                take = valid && ((V >> (ix & 63u)) & 1ull) != 0ull;
                if (COUNT) n_visited += (uint32_t)__popcll(__ballot(take));
            }
            w_eval[k] = take && !border;
            if (MODE == 2) {   // every evaluated window counts as rejected until it survives the prefix
                const unsigned long long E = __ballot(w_eval[k]);
                if (lane == 0 && ty < 32u) lds_F[ty] = E;
            }
            w_off[k] = 0u;
            w_q[k] = 0ull;
            if (w_eval[k]) {
                const uint32_t po = y * a.stride + x;
                w_off[k] = ((y - y0) * pitch + (x - x0)) * 4u;
                w_q[k] = ld_u64(sq_f, po * 8u, q0 * 8u) - ld_u64(sq_f, po * 8u, q1 * 8u) - ld_u64(sq_f, po * 8u, q2 * 8u) +
                         ld_u64(sq_f, po * 8u, q3 * 8u);
            }
        }
        if (MODE == 1 && COUNT && lane == 0 && n_visited != 0u) atomicAdd(a.stage_entered + VJ_MAX_STAGES_DEV, (unsigned long long)n_visited);
        cvt_barrier();   // the tile is in LDS

        const char* img = reinterpret_cast<const char*>(lds_img);
        kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + scales[slot].tile_table_first;
        // variance_norm_factor (:822-831): mean and squared mean in f64 from the window's sum (LDS) and squared sum
        double w_vnf[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            w_vnf[k] = 1.0;
            if (w_eval[k]) {
                const int32_t isum = (int32_t)(cvt_ld(img, w_off[k], t0) - cvt_ld(img, w_off[k], t1) - cvt_ld(img, w_off[k], t2) +
                                               cvt_ld(img, w_off[k], t3));
                const double mean = (double)isum * inv_area;
                double v = (double)w_q[k];
                v = v * inv_area - mean * mean;
                w_vnf[k] = v >= 0.0 ? sqrt(v) : 1.0;
            }
        }
        if (MODE == 0) {
            // stage 0 on the wave's four rows at once; a window outside the grid or on the border is "not a reject"
            double sum[NCH];
            cvt_stage_sum_any<TREE2, NCH>(img, table + stages[0].first_node, stages[0].n_nodes, stages[0].cv_f64, w_off, w_vnf, sum);
            const double thr0 = (double)stages[0].threshold;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const uint32_t ty = wib * (uint32_t)NCH + (uint32_t)k;
                const unsigned long long F = __ballot(w_eval[k] && !(sum[k] >= thr0));
                if (lane == 0 && ty < th && iy0 + ty < end_y) {
                    unsigned long long* word = bits + (iy0 + ty) * wpr + (ix0 >> 6);
                    if (tw == 64u) *word = F;
                    else if (F != 0ull) atomicOr(word, F << (ix0 & 63u));   // narrower tiles share a word (zeroed by the host)
                }
            }
        } else {
            // queue the visited windows, then the cascade stage by stage
            uint32_t* qo = lds_off + wib * CVT_WAVE_CAP;
            double* qv = lds_vnf + wib * CVT_WAVE_CAP;
            uint32_t n = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const unsigned long long mask = __ballot(w_eval[k]);
                if (w_eval[k]) {
                    const uint32_t tidx = (wib * (uint32_t)NCH + (uint32_t)k) * 64u + lane;   // row * 64 + column inside the tile
                    qo[n + mbcnt(mask)] = MODE == 2 ? w_off[k] | (tidx << 20) : w_off[k];
                    qv[n + mbcnt(mask)] = w_vnf[k];
                }
                n += (uint32_t)__popcll(mask);
            }
            __builtin_amdgcn_wave_barrier();
            auto flush = [&](const uint32_t* fo, uint32_t nn) {   // survivors of the last stage are detections
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(a.det_count, nn);
                g = __builtin_amdgcn_readfirstlane(g);
                for (uint32_t i = lane; i < nn; i += 64u) {
                    const uint32_t lo = fo[i] >> 2;
                    const uint32_t ly = lo / pitch, lx = lo - ly * pitch;
                    if (g + i < a.det_cap) a.det[g + i] = CvDet{x0 + lx, y0 + ly, slot, frame};
                }
            };
            bool finished = false;
            for (uint32_t st = 0; st < a.n_stages; ++st) {
                if (st != 0u && ((a.repack_mask >> st) & 1ull)) {
                    // pool the tile's survivors into contiguous runs of whole chunks (entries move through registers)
                    if (lane == 0) lds_cnt[wib] = n;
                    cvt_barrier();
                    uint32_t before = 0, total = 0;
#pragma unroll
                    for (uint32_t w = 0; w < (uint32_t)CVT_WAVES; ++w) {
                        const uint32_t c = lds_cnt[w];
                        before += w < wib ? c : 0u;
                        total += c;
                    }
                    before = __builtin_amdgcn_readfirstlane(before);
                    total = __builtin_amdgcn_readfirstlane(total);
                    uint32_t hold_o[NCH];
                    double hold_v[NCH];
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if ((uint32_t)k * 64u + lane < n) {
                            hold_o[k] = qo[(uint32_t)k * 64u + lane];
                            hold_v[k] = qv[(uint32_t)k * 64u + lane];
                        }
                    cvt_barrier();   // every wave holds its survivors in registers
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if ((uint32_t)k * 64u + lane < n) {
                            lds_off[before + (uint32_t)k * 64u + lane] = hold_o[k];
                            lds_vnf[before + (uint32_t)k * 64u + lane] = hold_v[k];
                        }
                    cvt_barrier();
                    if (st >= a.ws_begin && total != 0u && total <= a.ws_max) {
                        // few windows left: the rest of the cascade with every stage's stumps split over the waves
                        uint32_t* lds_x = reinterpret_cast<uint32_t*>(lds_vnf + CVT_WS_MAX);
                        const uint32_t left = cvt_wave_split<COUNT, TREE2>(a, img, table, lds_off, lds_vnf, lds_x, lds_cnt, total, st, lane, wib);
                        if (wib == 0u && left != 0u) flush(lds_off, left);
                        finished = true;
                        break;
                    }
                    const uint32_t share = 64u * (((total + 63u) / 64u + CVT_WAVES - 1u) / CVT_WAVES);
                    const uint32_t first = min(wib * share, total);
                    qo = lds_off + first;
                    qv = lds_vnf + first;
                    n = min(share, total - first);
                }
                if (n != 0u) {
                    if (COUNT && lane == 0) atomicAdd(a.stage_entered + st, (unsigned long long)n);
                    n = cvt_sweep<TREE2>(img, table + stages[st].first_node, stages[st].n_nodes, stages[st].cv_f64, (double)stages[st].threshold, qo, qv,
                                  n, lane);
                }
            }
            if (MODE == 2) {
                // survivors of the prefix: clear their reject bits, hand them to the tree walk; then the rows' words go out
                cvt_barrier();   // (lds_F is complete: every wave wrote its rows before the first sweep barrier... and none re-packs any more)
                if (n != 0u) {
                    // one sub-queue per scale (a chunk of cv_tree_chain_pass then holds windows of one scale); tq_shift = ~0: one
                    // flat queue (trees that are not made of chains: cv_tree_walk)
                    const bool flat = a.tq_shift == 0xffffffffu;
                    const uint64_t q_first = flat ? 0ull : cv_tq_first(scales[slot].tq_win_first, scales[slot].tq_slot, a.n_frames, a.tq_shift);
                    const uint64_t q_cap = flat ? (uint64_t)a.tq_cap : cv_tq_cap(scales[slot].end_x, scales[slot].end_y, a.n_frames, a.tq_shift);
                    uint32_t g = 0;
                    if (lane == 0) g = atomicAdd(a.tq_count + (flat ? 0u : scales[slot].tq_slot), n);
                    g = __builtin_amdgcn_readfirstlane(g);
                    for (uint32_t i = lane; i < n; i += 64u) {
                        const uint32_t raw = qo[i], tidx = raw >> 20, lo = (raw & CVT_OFF_MASK) >> 2;
                        const uint32_t ty = tidx >> 6, tx = tidx & 63u;
                        atomicAnd(&lds_F[ty], ~(1ull << tx));
                        const uint32_t ly = lo / pitch, lx = lo - ly * pitch;
                        const uint32_t ix = ix0 + tx, iy = iy0 + ty;
                        if ((uint64_t)g + i < q_cap && q_first + g + i < (uint64_t)a.tq_cap)
                            a.tq[q_first + g + i] = CvTreeEntry{frame * frame_bytes4 + ((y0 + ly) * a.stride + (x0 + lx)) * 4u,
                                                      frame * a.bits_frame_words + scales[slot].bits_base + iy * wpr + (ix >> 6), (ix & 63u) | (slot << 8),
                                                      0u, qv[i]};
                    }
                }
                cvt_barrier();
                for (uint32_t ty = threadIdx.x; ty < th; ty += (uint32_t)CVT_WAVES * 64u)
                    if (iy0 + ty < end_y) {
                        unsigned long long* word = bits + (iy0 + ty) * wpr + (ix0 >> 6);
                        const unsigned long long F = lds_F[ty];
                        if (tw == 64u) *word = F;
                        else if (F != 0ull) atomicOr(word, F << (ix0 & 63u));
                    }
            } else if (!finished && n != 0u) flush(qo, n);
        }
        cvt_barrier();   // the tile is finished: lds_cnt may carry the next ticket
        if (threadIdx.x == 0) lds_cnt[40] = next_u;
        cvt_barrier();
        u = __builtin_amdgcn_readfirstlane(lds_cnt[40]);
    }
}

int prepare_cv_tile_kernels() {
    const int max_lds = 160 * 1024;
    const void* fns[] = {(const void*)cv_tile_pass<0, false, false>, (const void*)cv_tile_pass<1, false, false>,
                         (const void*)cv_tile_pass<1, true, false>,  (const void*)cv_tile_pass<2, false, false>,
                         (const void*)cv_tile_pass<0, false, true>,  (const void*)cv_tile_pass<1, false, true>,
                         (const void*)cv_tile_pass<1, true, true>};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int launch_cv_tile_pass(const CvTileArgs& a, int mode, bool count, bool tree2, int n_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    dim3 g(n_blocks), b(CVT_WAVES * 64);
    if (tree2) {
        if (mode == 0) hipLaunchKernelGGL((cv_tile_pass<0, false, true>), g, b, a.lds_bytes, stream, a);
        else if (mode == 2) return (int)hipErrorInvalidValue;      // stage trees on tiles are stump cascades
        else if (count) hipLaunchKernelGGL((cv_tile_pass<1, true, true>), g, b, a.lds_bytes, stream, a);
        else hipLaunchKernelGGL((cv_tile_pass<1, false, true>), g, b, a.lds_bytes, stream, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((cv_tile_pass<0, false, false>), g, b, a.lds_bytes, stream, a);
        else if (mode == 2) hipLaunchKernelGGL((cv_tile_pass<2, false, false>), g, b, a.lds_bytes, stream, a);
        else if (count) hipLaunchKernelGGL((cv_tile_pass<1, true, false>), g, b, a.lds_bytes, stream, a);
        else hipLaunchKernelGGL((cv_tile_pass<1, false, false>), g, b, a.lds_bytes, stream, a);
    }
    return (int)hipGetLastError();
}

}  // namespace vj
