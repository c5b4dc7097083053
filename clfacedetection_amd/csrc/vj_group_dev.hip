// cv::groupRectangles on the device, per frame — what vj_group.cpp does on the host (tempcv.cpp:130-243; the
// reference's filterResult, clod.cpp:182-357, is its broken port) — so that the GROUPED faces of a first cascade can
// become the regions of a second one without leaving the device (vj_detect_chain).
//
// The result must equal the host's on the candidates in their canonical order (frame, scale, y, x):
//   * cv::partition labels the connected components of the "similar" graph in order of first appearance, i.e. by
//     their smallest member index.  Here: the frame's candidates are sorted in LDS (bitonic, 64-bit keys), every
//     candidate starts as its own label, and labels are lowered to the smallest label among similar candidates
//     (followed by pointer jumping) until nothing changes: a label then is the smallest index of its component, and the
//     classes in the order of their labels are partition()'s classes.
//   * the class sums are integer (LDS atomics: order-free), the averages use the same f32 operations, and the
//     containment filter the same integer / f64 comparisons as the host code.
// -ffp-contract=off as everywhere.
#include <hip/hip_runtime.h>
#include <climits>
#include "vj_device.hpp"
#include "vj_devutil.hpp"

namespace vj {

constexpr uint32_t GROUP_THREADS = 1024;

__global__ __launch_bounds__(256) void group_count(GroupArgs g) {
    const uint32_t n = min(*g.det_count, g.det_cap);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t frame = g.det[i].off / g.frame_bytes;
        if (frame < g.n_frames) atomicAdd(g.frame_count + frame, 1u);
    }
}

// Exclusive prefix of `in[0..n)` into out[0..n], out[n] = total; one workgroup.
__device__ __forceinline__ void block_exclusive_prefix(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* lds /* >= 17 */) {
    const uint32_t lane = lane_id(), wib = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += t;
        }
        if (lane == 63u) lds[wib] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < n_waves; ++w) {
            const uint32_t c = lds[w];
            before += w < wib ? c : 0u;
            total += c;
        }
        if (i < n) out[i] = carry + before + incl - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

__global__ __launch_bounds__(GROUP_THREADS) void group_offsets(GroupArgs g) {
    __shared__ uint32_t lds[GROUP_THREADS / 64 + 1];
    block_exclusive_prefix(g.frame_count, g.frame_first, g.n_frames, lds);
}

__global__ __launch_bounds__(256) void group_scatter(GroupArgs g) {
    const uint32_t n = min(*g.det_count, g.det_cap);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const DetEntry d = g.det[i];
        const uint32_t frame = d.off / g.frame_bytes;
        if (frame >= g.n_frames) continue;
        const uint32_t el = (d.off - frame * g.frame_bytes) >> 2;
        const uint32_t pos = g.frame_first[frame] + atomicAdd(g.frame_cursor + frame, 1u);
        g.keys[pos] = (uint64_t)d.scale << 32 | el;   // (scale, y, x) order == (scale, element) order
    }
}

// ASimilarRects (tempcv.cpp:130-143), as vj_group.cpp evaluates it
__device__ __forceinline__ bool similar(int x1, int y1, int w1, int h1, int x2, int y2, int w2, int h2, double eps) {
    const double delta = eps * (double)(min(w1, w2) + min(h1, h2)) * 0.5;
    return (double)abs(x1 - x2) <= delta && (double)abs(y1 - y2) <= delta && (double)abs(x1 + w1 - x2 - w2) <= delta &&
           (double)abs(y1 + h1 - y2 - h2) <= delta;
}

// Order-preserving ranks of the set flags among items [0, n): rank_out[i] = number of set flags below i; returns the
// total.  Every thread of the workgroup calls it; scratch >= 33 words.
__device__ __forceinline__ uint32_t block_rank(const uint32_t* flag, uint32_t* rank_out, uint32_t n, uint32_t* scratch) {
    const uint32_t lane = lane_id(), wib = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += GROUP_THREADS) {   // at most two rounds
        const uint32_t i = i0 + threadIdx.x;
        const bool f = i < n && flag[i] != 0u;
        const unsigned long long m = __ballot(f);
        if (lane == 0u) scratch[wib] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < GROUP_THREADS / 64; ++w) {
            const uint32_t c = scratch[w];
            before += w < wib ? c : 0u;
            total += c;
        }
        if (i < n) rank_out[i] = carry + before + mbcnt(m);
        carry += total;
        __syncthreads();
    }
    return carry;
}

__global__ __launch_bounds__(GROUP_THREADS) void group_frame(GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // layout (words): keys / class sums share the front
    uint64_t* keys = reinterpret_cast<uint64_t*>(lds);              // [GROUP_MAX] u64 — dead after the decode
    int32_t* cx = reinterpret_cast<int32_t*>(lds);                  // class sums, then averages: 5 arrays of GROUP_MAX
    int32_t* cy = cx + GROUP_MAX;
    int32_t* cw = cy + GROUP_MAX;
    int32_t* ch = cw + GROUP_MAX;
    uint32_t* cn = reinterpret_cast<uint32_t*>(ch + GROUP_MAX);
    int32_t* rx = reinterpret_cast<int32_t*>(cn + GROUP_MAX);       // the candidates
    int32_t* ry = rx + GROUP_MAX;
    int32_t* rw = ry + GROUP_MAX;
    int32_t* rh = rw + GROUP_MAX;
    uint32_t* label = reinterpret_cast<uint32_t*>(rh + GROUP_MAX);
    uint32_t* aux = label + GROUP_MAX;                              // root flags -> class ranks; keep flags -> output ranks
    uint32_t* scratch = aux + GROUP_MAX;                            // 40 words
    const uint32_t frame = blockIdx.x;
    const uint32_t first = g.frame_first[frame];
    const uint32_t n = g.frame_first[frame + 1u] - first;
    const uint32_t tid = threadIdx.x;
    if (n == 0u) return;
    if (n > min(g.group_max, GROUP_MAX)) {
        if (tid == 0u) atomicAdd(g.overflow, 1u);
        return;
    }
    // ---- canonical order: sort the frame's keys
    uint32_t P = 2;
    while (P < n) P <<= 1;
    for (uint32_t i = tid; i < P; i += GROUP_THREADS) keys[i] = i < n ? g.keys[first + i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j != 0u; j >>= 1) {
            for (uint32_t i = tid; i < P; i += GROUP_THREADS) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = keys[i], b = keys[l];
                    if (((i & k) == 0u) == (a > b)) {
                        keys[i] = b;
                        keys[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < n; i += GROUP_THREADS) {
        const uint64_t key = keys[i];
        const uint32_t slot = (uint32_t)(key >> 32), el = (uint32_t)key;
        const uint32_t y = el / g.stride;
        rx[i] = (int32_t)(el - y * g.stride);
        ry[i] = (int32_t)y;
        rw[i] = (int32_t)g.scales[slot].win_w;
        rh[i] = (int32_t)g.scales[slot].win_h;
        label[i] = i;
    }
    __syncthreads();
    // ---- connected components: label = smallest index of the component
    for (;;) {
        if (tid == 0u) scratch[36] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += GROUP_THREADS) {
            const int x1 = rx[i], y1 = ry[i], w1 = rw[i], h1 = rh[i];
            uint32_t m = label[i];
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t lj = label[j];   // racing with j's own update: any value read is a member of j's component
                if (lj < m && similar(x1, y1, w1, h1, rx[j], ry[j], rw[j], rh[j], g.eps)) m = lj;   // (symmetric in value)
            }
            if (m < label[i]) {
                label[i] = m;
                scratch[36] = 1u;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += GROUP_THREADS) {   // pointer jumping
            uint32_t l = label[i];
            while (label[l] < l) l = label[l];
            label[i] = l;
        }
        __syncthreads();
        if (scratch[36] == 0u) break;
        __syncthreads();
    }
    // ---- classes in order of first appearance
    for (uint32_t i = tid; i < n; i += GROUP_THREADS) aux[i] = label[i] == i ? 1u : 0u;
    __syncthreads();
    // a root's rank among the roots is its class index (aux is overwritten in place; only root positions are meaningful)
    const uint32_t ncls = block_rank(aux, aux, n, scratch);
    __syncthreads();
    for (uint32_t i = tid; i < ncls; i += GROUP_THREADS) {   // (the key array is dead: the sums live there)
        cx[i] = 0; cy[i] = 0; cw[i] = 0; ch[i] = 0; cn[i] = 0u;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += GROUP_THREADS) {
        const uint32_t c = aux[label[i]];
        // int accumulators as in the original (tempcv.cpp:167-172); atomics on int wrap like the host's unsigned adds
        atomicAdd(cx + c, rx[i]);
        atomicAdd(cy + c, ry[i]);
        atomicAdd(cw + c, rw[i]);
        atomicAdd(ch + c, rh[i]);
        atomicAdd(cn + c, 1u);
    }
    __syncthreads();
    auto sat = [](float v) { return v > (float)INT_MAX ? INT_MAX : (int)v; };
    for (uint32_t i = tid; i < ncls; i += GROUP_THREADS) {
        const float s = 1.f / (float)(int)cn[i];
        cx[i] = sat((float)cx[i] * s);
        cy[i] = sat((float)cy[i] * s);
        cw[i] = sat((float)cw[i] * s);
        ch[i] = sat((float)ch[i] * s);
    }
    __syncthreads();
    // ---- drop weak classes and small rectangles inside larger, better supported ones (tempcv.cpp:205-242)
    for (uint32_t i = tid; i < ncls; i += GROUP_THREADS) {
        const int n1 = (int)cn[i];
        uint32_t keep = n1 > g.threshold ? 1u : 0u;
        if (keep) {
            const int x1 = cx[i], y1 = cy[i], w1 = cw[i], h1 = ch[i];
            for (uint32_t j = 0; j < ncls; ++j) {
                const int n2 = (int)cn[j];
                if (j == i || n2 <= g.threshold) continue;
                const int x2 = cx[j], y2 = cy[j], w2 = cw[j], h2 = ch[j];
                const int dx = (double)w2 * g.eps > (double)INT_MAX ? INT_MAX : (int)((double)w2 * g.eps);
                const int dy = (double)h2 * g.eps > (double)INT_MAX ? INT_MAX : (int)((double)h2 * g.eps);
                typedef long long ll;
                if (x1 >= (ll)x2 - dx && y1 >= (ll)y2 - dy && (ll)x1 + w1 <= (ll)x2 + w2 + dx && (ll)y1 + h1 <= (ll)y2 + h2 + dy &&
                    (n2 > max(3, n1) || n1 < 3)) {
                    keep = 0u;
                    break;
                }
            }
        }
        label[i] = keep;   // (the labels are no longer needed)
    }
    __syncthreads();
    const uint32_t n_out = block_rank(label, aux, ncls, scratch);
    for (uint32_t i = tid; i < ncls; i += GROUP_THREADS)
        if (label[i]) {
            g.grouped[first + aux[i]] = RoiDev{(int32_t)frame, cx[i], cy[i], cw[i], ch[i]};
            g.grouped_weight[first + aux[i]] = cn[i];
        }
    if (tid == 0u) g.grouped_count[frame] = n_out;
}

// Concatenate the frames' grouped rectangles in frame order: the region list of the second cascade.
__global__ __launch_bounds__(GROUP_THREADS) void group_collect(GroupArgs g) {
    __shared__ uint32_t lds[GROUP_THREADS / 64 + 1];
    // frame_cursor is free again: it takes the output offsets (n_frames + 1 would overrun it by one: the total goes to lds)
    const uint32_t lane = lane_id(), wib = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < g.n_frames; i0 += GROUP_THREADS) {
        const uint32_t f = i0 + threadIdx.x;
        const uint32_t v = f < g.n_frames ? g.grouped_count[f] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += t;
        }
        if (lane == 63u) lds[wib] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < GROUP_THREADS / 64; ++w) {
            const uint32_t c = lds[w];
            before += w < wib ? c : 0u;
            total += c;
        }
        if (f < g.n_frames) {
            const uint32_t dst = carry + before + incl - v, src = g.frame_first[f];
            for (uint32_t k = 0; k < v; ++k)
                if (dst + k < g.max_rois) {
                    g.rois[dst + k] = g.grouped[src + k];
                    g.roi_weight[dst + k] = g.grouped_weight[src + k];
                }
        }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *g.n_rois = carry;
}

static constexpr size_t GROUP_LDS_BYTES = ((size_t)GROUP_MAX * 11u + 40u) * 4u;

int prepare_group_kernels() {
    return (int)hipFuncSetAttribute((const void*)group_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GROUP_LDS_BYTES);
}

int launch_group_rois(const GroupArgs& g, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // frame_count | frame_cursor | grouped_count | overflow are one block
    hipError_t e = hipMemsetAsync(g.frame_count, 0, ((size_t)3u * g.n_frames + 1u) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(group_count, dim3(256), dim3(256), 0, stream, g);
    hipLaunchKernelGGL(group_offsets, dim3(1), dim3(GROUP_THREADS), 0, stream, g);
    hipLaunchKernelGGL(group_scatter, dim3(256), dim3(256), 0, stream, g);
    hipLaunchKernelGGL(group_frame, dim3(g.n_frames), dim3(GROUP_THREADS), GROUP_LDS_BYTES, stream, g);
    hipLaunchKernelGGL(group_collect, dim3(1), dim3(GROUP_THREADS), 0, stream, g);
    return (int)hipGetLastError();
}

}  // namespace vj
