// gfx950 kernel of the OpenCV arithmetic profile: cvHaarDetectObjects' scale-cascade path as the
// reference keeps it in tempcv.cpp (a private copy of OpenCV 2.4.2 haar.cpp) —
//   cvRunHaarClassifierCascadeSum   tempcv.cpp:795-972   f64 variance, f64 node and stage sums, border rule
//   icvEvalHidHaarClassifier        tempcv.cpp:771-792   sum < t ? left : right
//   HaarDetectObjects_ScaleCascade_Invoker  :1116-1185    x = cvRound(ix * ystep), ixstep = result != 0 ? 1 : 2
// Second arithmetic profile of the library; the clod profile (vj_kernels.hip) is the
// contract of the headline path.  MUST be compiled with -ffp-contract=off.
//
// One wave walks one window row.  The sequential "skip the next window after a stage-0 reject" rule is a
// recurrence over the row, e[i] = !(e[i-1] && f[i-1]); because a window that follows a non-reject is always
// visited, e[i] only depends on the PARITY of the run of stage-0 rejects that ends at i-1 (f computed for
// every grid position), which a wave gets from one __ballot per 64 positions plus one carry bit.
#include <hip/hip_runtime.h>
#include "vj_device.hpp"
#include "vj_devutil.hpp"

namespace vj {

struct CvQEntry {
    uint32_t off;   // byte offset of the window origin in the batch sum image
    uint32_t xy;    // x | y << 16
    double vnf;     // variance_norm_factor
};

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }   // cvRound: half to even

// One node: sum of the weighted rectangles in f64 (tempcv.cpp:868-888): (double)int * (double)float, added in order.
__device__ __forceinline__ double cv_node_sum(rsrc_t img, const NodeRecDev& r, uint32_t off) {
    const uint32_t dw0 = (uint32_t)(int32_t)(int16_t)(r[6] & 0xffffu), dw1 = (uint32_t)((int32_t)r[6] >> 16),
                   dw2 = (uint32_t)(int32_t)(int16_t)(r[7] & 0xffffu);
    const int32_t r0 = (int32_t)(ld_u32(img, off, r[0]) - ld_u32(img, off, r[0] + dw0) - ld_u32(img, off, r[0] + r[3]) +
                                 ld_u32(img, off, r[0] + r[3] + dw0));
    const int32_t r1 = (int32_t)(ld_u32(img, off, r[1]) - ld_u32(img, off, r[1] + dw1) - ld_u32(img, off, r[1] + r[4]) +
                                 ld_u32(img, off, r[1] + r[4] + dw1));
    double s = (double)r0 * (double)__uint_as_float(r[8]);
    s += (double)r1 * (double)__uint_as_float(r[9]);
    if (__uint_as_float(r[10]) != 0.0f) {   // uniform
        const int32_t r2 = (int32_t)(ld_u32(img, off, r[2]) - ld_u32(img, off, r[2] + dw2) - ld_u32(img, off, r[2] + r[5]) +
                                     ld_u32(img, off, r[2] + r[5] + dw2));
        s += (double)r2 * (double)__uint_as_float(r[10]);
    }
    return s;
}

// One stage on one window: stumps through the scalar cache; multi-node trees visit their records in index
// order under the lanes whose walk sits on them (a child always follows its parent), as stage_sum_trees does.
template <bool TREES>
__device__ __forceinline__ double cv_stage_sum(rsrc_t img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off, double vnf) {
    double stage_sum = 0.0;
    if (!TREES) {
        NodeRecDev r = tab[0];
        for (uint32_t j = 0; j < n_nodes; ++j) {
            const NodeRecDev rn = tab[j + 1 < n_nodes ? j + 1 : j];
            const double t = (double)__uint_as_float(r[11]) * vnf;
            const double s = cv_node_sum(img, r, off);
            stage_sum += (double)(s < t ? __uint_as_float(r[12]) : __uint_as_float(r[13]));
            r = rn;
        }
        return stage_sum;
    }
    uint32_t cur = 0, k = 0;
    float value = 0.0f;
    bool done = false;
    for (uint32_t j = 0; j < n_nodes; ++j) {
        const NodeRecDev r = tab[j];
        const uint32_t flags = r[7] >> 16;
        if (!done && cur == k) {
            const double t = (double)__uint_as_float(r[11]) * vnf;
            const bool go_left = cv_node_sum(img, r, off) < t;
            const uint32_t nxt = go_left ? r[12] : r[13];
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

template <bool TREES, bool COUNT>
__device__ __forceinline__ void cv_flush(const CvArgs& a, rsrc_t img, kptr<NodeRecDev> table, CvQEntry* q, uint32_t& n,
                                         uint32_t slot, uint32_t frame, uint32_t lane) {
    kptr<StageDev> stages = as_k(a.stages);
    for (uint32_t s = 1; s < a.n_stages && n != 0u; ++s) {
        if (COUNT && lane == 0) atomicAdd(a.stage_entered + s, (unsigned long long)n);
        kptr<NodeRecDev> tab = table + stages[s].first_node;
        const uint32_t n_nodes = stages[s].n_nodes;
        const double thr = (double)stages[s].threshold;
        uint32_t m = 0;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + lane;
            const bool act = i < n;
            const CvQEntry e = q[act ? i : 0u];
            bool pass = false;
            if (act) pass = cv_stage_sum<TREES>(img, tab, n_nodes, e.off, e.vnf) >= thr;
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

template <bool TREES, bool COUNT>
__global__ __launch_bounds__(CV_WAVES_PER_BLOCK * 64) void cv_profile_pass(CvArgs a) {
    __shared__ CvQEntry lds_q[CV_WAVES_PER_BLOCK * CV_QCAP];
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CvQEntry* q = lds_q + wib * CV_QCAP;
    const uint32_t rank = blockIdx.x * CV_WAVES_PER_BLOCK + wib;
    kptr<CvScaleDev> scales = as_k(a.scales);
    kptr<UnitDev> rows = as_k(a.rows);
    kptr<StageDev> stages = as_k(a.stages);
    const uint32_t frame_bytes4 = a.frame_elems * 4u;
    const rsrc_t img = make_rsrc(a.sum, a.n_frames * frame_bytes4);
    const uint32_t total = a.n_rows * a.n_frames;
    const unsigned long long below = (1ull << lane) - 1ull;

    for (uint32_t u = rank; u < total; u += a.total_waves) {
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
        uint32_t carry = 0;   // parity of the run of stage-0 rejects that ends at the last position seen
        uint32_t n_q = 0;
        for (uint32_t ix0 = 0; ix0 < end_x; ix0 += 64u) {
            const uint32_t ix = ix0 + lane;
            const bool valid = ix < end_x;
            const uint32_t x = (uint32_t)cv_round((double)(valid ? ix : 0u) * ystep);
            const bool border = row_border || x + win_w >= a.stride;
            const uint32_t po = y * a.stride + x;
            const uint32_t off = frame_bytes + po * 4u;
            bool fail0 = false;
            double vnf = 1.0;
            if (valid && !border) {
                const int32_t isum = (int32_t)(ld_u32(img, off, q0 * 4u) - ld_u32(img, off, q1 * 4u) - ld_u32(img, off, q2 * 4u) +
                                               ld_u32(img, off, q3 * 4u));
                const uint64_t qq = ld_u64(sq_f, po * 8u, q0 * 8u) - ld_u64(sq_f, po * 8u, q1 * 8u) - ld_u64(sq_f, po * 8u, q2 * 8u) +
                                    ld_u64(sq_f, po * 8u, q3 * 8u);
                const double mean = (double)isum * inv_area;
                vnf = (double)qq;
                vnf = vnf * inv_area - mean * mean;
                vnf = vnf >= 0.0 ? sqrt(vnf) : 1.0;
                fail0 = !(cv_stage_sum<TREES>(img, table + stages[0].first_node, stages[0].n_nodes, off, vnf) >= thr0);
            }
            // which positions does the sequential walk visit?  parity of the reject run below each lane
            const unsigned long long F = __ballot(fail0);
            const unsigned long long zeros = ~F & below;
            uint32_t parity;
            if (zeros == 0ull) parity = (lane & 1u) ^ carry;
            else parity = (lane - 1u - (63u - (uint32_t)__clzll((long long)zeros))) & 1u;
            const bool visited = valid && parity == 0u;
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
            // carry for the next 64 positions
            const uint32_t n_valid = min(64u, end_x - ix0);
            const unsigned long long vmask = n_valid == 64u ? ~0ull : (1ull << n_valid) - 1ull;
            const unsigned long long zall = ~F & vmask;
            if (zall == 0ull) carry ^= n_valid & 1u;
            else carry = (n_valid - 1u - (63u - (uint32_t)__clzll((long long)zall))) & 1u;
            __builtin_amdgcn_wave_barrier();
            if (n_q > (uint32_t)CV_QCAP - 64u) cv_flush<TREES, COUNT>(a, img, table, q, n_q, slot, frame, lane);
        }
        if (n_q != 0u) cv_flush<TREES, COUNT>(a, img, table, q, n_q, slot, frame, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_cv_profile_pass(const CvArgs& a, bool trees, bool count, int n_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    dim3 g(n_blocks), b(CV_WAVES_PER_BLOCK * 64);
    if (trees) {
        if (count) hipLaunchKernelGGL((cv_profile_pass<true, true>), g, b, 0, stream, a);
        else       hipLaunchKernelGGL((cv_profile_pass<true, false>), g, b, 0, stream, a);
    } else {
        if (count) hipLaunchKernelGGL((cv_profile_pass<false, true>), g, b, 0, stream, a);
        else       hipLaunchKernelGGL((cv_profile_pass<false, false>), g, b, 0, stream, a);
    }
    return (int)hipGetLastError();
}

}  // namespace vj
