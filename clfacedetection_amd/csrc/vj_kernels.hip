// gfx950 (CDNA4, wave64) kernels of the Viola–Jones detect path.
//
//  * integral + squared-integral image  (what cvIntegral computes for the reference,
//    clif.cpp:280-285, 326-335; the reference's own integral kernels clif.cl:79-120
//    are unused by detect and wrong — SURVEY.md §2.1)
//  * the cascade evaluator, replacing runStage (clod.cl:32-93) and its host driver
//    loop (clod.cpp:1212-1322): one window per lane, per-wave survivor queue in LDS
//    compacted with __ballot after every stage, node records fetched through the
//    scalar cache (they are uniform across the wave), a few launches per batch
//    instead of one launch + host round trip per (scale, stage).
//
// Arithmetic contract: IEEE binary32, no contraction, operations in
// the reference's order.  This file MUST be compiled with -ffp-contract=off; HIP's
// default correctly-rounded f32 divide/sqrt is relied upon.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include "vj_device.hpp"
#include "vj_devutil.hpp"

#ifndef VJ_STAMPS
#define VJ_STAMPS 0
#endif
// Experiment switch (tools/ab_build.sh): pin the scalar loads of the NEXT node record(s) at the top of a stump iteration,
// ahead of the iteration's LDS gathers.  hipcc otherwise sinks them to the end of the iteration — LDS and scalar loads share
// lgkmcnt, and an outstanding scalar load turns every counted LDS wait into lgkmcnt(0) — where their latency is exposed
// (profiles/r04_notes.md #3).
#ifndef VJ_SCHED_PREFETCH
#define VJ_SCHED_PREFETCH 0
#endif
#define VJ_PIN_PREFETCH() do { } while (0)
namespace vj {

// The next node record, fetched through the scalar cache WITHOUT telling the compiler's wait-count pass: the load is issued
// where it stands (hipcc sinks an ordinary load of `tab[j + 1]` to the end of the iteration, next to its first use, because a
// scalar load in flight turns every counted LDS wait into lgkmcnt(0)), the iteration's LDS gathers keep their counted waits
// (an extra operation in flight only makes a counted wait stricter), and rec_arrived() — an explicit lgkmcnt(0) that the
// record's uses depend on — stands where the record is needed.
__device__ __forceinline__ NodeRecDev rec_fetch(kptr<NodeRecDev> p) {
    NodeRecDev r;
    // (the address is wave-uniform by construction; where the compiler cannot prove it, it must still sit in SGPRs)
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint64_t u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(r) : "s"(u));
    return r;
}
__device__ __forceinline__ void rec_arrived(NodeRecDev& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }
__device__ __forceinline__ void rec_arrived(NodeRecDev& a, NodeRecDev& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

// Sum of a float over the 64 lanes, left in lane 63, with DPP adds only (no LDS crossbar traffic):
// butterflies inside quads and rows of 16, then row_bcast15 / row_bcast31 across the rows.  The
// order of the additions is NOT lane order — callers must not need that.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    auto dpp = [](float x, auto ctrl, auto row_mask) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});   // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});   // row_mirror
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});   // row_bcast15 -> rows 1, 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});   // row_bcast31 -> rows 2, 3
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load (s_waitcnt vmcnt(0)), which would serialise the record prefetches of the stump-parallel
// finish behind a full memory round trip per block.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ===================================================================== integral
// Three launches per batch:
//   1. band_colsum : per 8-row band, per column: sum and sum of squares (reads u8 once)
//   2. band_scan   : per column, exclusive prefix of those over the bands
//   3. band_rows   : one wave per band; top edge = prefix along x of the column totals above, then
//                    per row a wave prefix scan along x (DPP) + carry of the chunks to the left.
// All integers, so the result is exact: sum wraps mod 2^32 like CV_32S, sqsum is u64.

__device__ __forceinline__ uint32_t load_px4(const uint8_t* row, uint32_t x, uint32_t width) {
    // four pixels x..x+3 packed little-endian; pixels beyond the row read as 0
    const uint8_t* p = row + x;
    if (x + 4 <= width && ((uintptr_t)p & 3u) == 0) return *reinterpret_cast<const uint32_t*>(p);
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (x + c < width) v |= (uint32_t)p[c] << (8 * c);
    return v;
}

// Image ingest: 3-channel BGR / 4-channel BGRA frames are converted on the fly with OpenCV's 8-bit
// fixed-point BGR2GRAY, (1868 B + 9617 G + 4899 R + 8192) >> 14 (OpenCV 2.4.2 imgproc, the cvCvtColor the
// reference calls at clif.cpp:328 — third-party arithmetic), fused into both pixel reads
// of the integral so that no gray copy is written.
__device__ __forceinline__ uint32_t bgr2gray(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14;
}
__device__ __forceinline__ uint32_t load_gray4(const uint8_t* row, uint32_t x, uint32_t width, uint32_t ch) {
    if (ch <= 1u) return load_px4(row, x, width);
    const uint8_t* p = row + (size_t)x * ch;
    uint32_t v = 0;
    if (x + 4 <= width && ((uintptr_t)p & 3u) == 0) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
        if (ch == 3u) {   // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            v = bgr2gray(w0 & 0xffu, (w0 >> 8) & 0xffu, (w0 >> 16) & 0xffu) |
                bgr2gray(w0 >> 24, w1 & 0xffu, (w1 >> 8) & 0xffu) << 8 |
                bgr2gray((w1 >> 16) & 0xffu, w1 >> 24, w2 & 0xffu) << 16 |
                bgr2gray((w2 >> 8) & 0xffu, (w2 >> 16) & 0xffu, w2 >> 24) << 24;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v |= bgr2gray(w[c] & 0xffu, (w[c] >> 8) & 0xffu, (w[c] >> 16) & 0xffu) << (8 * c);
        }
        return v;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (x + c < width) v |= bgr2gray(p[c * ch], p[c * ch + 1u], p[c * ch + 2u]) << (8 * c);
    return v;
}

__global__ __launch_bounds__(256) void band_colsum(IntegralArgs a) {
    const uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 4u;
    const uint32_t band = blockIdx.y, frame = blockIdx.z;
    if (x >= a.band_pitch) return;
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    uint32_t s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const uint32_t y0 = band * BAND_ROWS;
#pragma unroll
    for (int r = 0; r < BAND_ROWS; ++r) {
        const uint32_t y = y0 + r;
        if (y < a.height && x < a.width) {
            const uint32_t v = load_gray4(img + (size_t)y * a.gray_stride, x, a.width, a.channels);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t p = (v >> (8 * c)) & 0xffu;
                s[c] += p;
                q[c] += p * p;
            }
        }
    }
    const size_t o = ((size_t)frame * a.n_bands + band) * a.band_pitch + x;
    *reinterpret_cast<uint4*>(a.band_sum + o) = make_uint4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<uint4*>(a.band_sq + o) = make_uint4(q[0], q[1], q[2], q[3]);
}

__global__ __launch_bounds__(256) void band_scan(IntegralArgs a) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    const uint32_t frame = blockIdx.y;
    if (x >= a.band_pitch) return;
    uint32_t s = 0;
    uint64_t q = 0;
    size_t o = (size_t)frame * a.n_bands * a.band_pitch + x;
    // eight bands per step: the loads of a step are independent of its stores (the prefix is written in place)
    for (uint32_t b0 = 0; b0 < a.n_bands; b0 += 8u) {
        uint32_t ts[8], tq[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const bool in = b0 + k < a.n_bands;
            ts[k] = in ? a.band_sum[o + (size_t)k * a.band_pitch] : 0u;
            tq[k] = in ? a.band_sq[o + (size_t)k * a.band_pitch] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            if (b0 + k < a.n_bands) {
                a.band_sum[o + (size_t)k * a.band_pitch] = s;         // exclusive prefix, in place
                a.band_sq_prefix[o + (size_t)k * a.band_pitch] = q;
            }
            s += ts[k];
            q += tq[k];
        }
        o += (size_t)8u * a.band_pitch;
    }
}

// The same prefix over the bands in TWO levels, for latency: one workgroup takes 64 columns, wave g the g-th group of BPG bands —
// every lane loads its column's BPG band totals at once (one memory round trip instead of n_bands / 8 dependent ones: the
// single-frame integral is latency-bound, 135 bands at 1080p), scans them in registers, and adds the totals of the groups
// above from LDS.  Same arithmetic (integer adds), same outputs as band_scan.
template <int BPG>
__global__ __launch_bounds__(1024) void band_scan2(IntegralArgs a) {
    __shared__ uint32_t tot_s[16][64];
    __shared__ uint64_t tot_q[16][64];
    const uint32_t lane = threadIdx.x & 63u, g = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * 64u + lane, frame = blockIdx.y;
    const bool in = x < a.band_pitch;
    const uint32_t b0 = g * (uint32_t)BPG;
    const size_t o = ((size_t)frame * a.n_bands + b0) * a.band_pitch + x;
    uint32_t ts[BPG], tq[BPG];
#pragma unroll
    for (int k = 0; k < BPG; ++k) {
        const bool have = in && b0 + (uint32_t)k < a.n_bands;
        ts[k] = have ? a.band_sum[o + (size_t)k * a.band_pitch] : 0u;
        tq[k] = have ? a.band_sq[o + (size_t)k * a.band_pitch] : 0u;
    }
    uint32_t s = 0;
    uint64_t q = 0;
#pragma unroll
    for (int k = 0; k < BPG; ++k) {
        s += ts[k];
        q += tq[k];
    }
    tot_s[g][lane] = s;
    tot_q[g][lane] = q;
    __syncthreads();
    s = 0;
    q = 0;
    for (uint32_t h = 0; h < g; ++h) {   // (uniform trip count per wave)
        s += tot_s[h][lane];
        q += tot_q[h][lane];
    }
#pragma unroll
    for (int k = 0; k < BPG; ++k) {
        if (in && b0 + (uint32_t)k < a.n_bands) {
            a.band_sum[o + (size_t)k * a.band_pitch] = s;         // exclusive prefix, in place
            a.band_sq_prefix[o + (size_t)k * a.band_pitch] = q;
        }
        s += ts[k];
        q += tq[k];
    }
}

// Inclusive prefix sums over the 64 lanes with DPP adds only (no LDS crossbar traffic): shifts by 1, 2, 4, 8 inside
// the rows of 16, then row_bcast15 / row_bcast31 carry the row totals on.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp_or_zero<0x111, 0xf>(v);   // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);   // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);   // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);   // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);   // row_bcast15 -> rows 1, 3
    v += dpp_or_zero<0x143, 0xc>(v);   // row_bcast31 -> rows 2, 3
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_or_zero(uint64_t v) {
    return (uint64_t)dpp_or_zero<CTRL, ROW_MASK>((uint32_t)v) | (uint64_t)dpp_or_zero<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32;
}
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v) {
    v += dpp_or_zero<0x111, 0xf>(v);
    v += dpp_or_zero<0x112, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa>(v);
    v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_last(T v);
template <>
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
template <>
__device__ __forceinline__ uint64_t wave_last(uint64_t v) {
    return (uint64_t)wave_last((uint32_t)v) | (uint64_t)wave_last((uint32_t)(v >> 32)) << 32;
}

typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(4)));
typedef uint64_t u64x2_unaligned __attribute__((ext_vector_type(2), aligned(8)));
// NT: non-temporal stores — the batch variant's outputs (1.6 GB per 64 x 1080p) pass through the caches once and are read much
// later; a call whose outputs fit the Infinity Cache keeps ordinary stores (profiles/r04_notes.md #6).
template <bool NT, typename V>
__device__ __forceinline__ void out_store(V* p, V v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// One wave per band of 8 rows, walking it in chunks of 256 columns (4 per lane).  The integral at (y, x) is the
// integral at the band's top edge — a prefix along x of the column totals above the band, scanned once per chunk —
// plus the prefixes along x of the band's rows down to y.  A row's prefix of squares stays below 2^32 for rows of up
// to 66051 pixels (Q = uint32_t: one 32-bit DPP scan per row and image); wider images scan in 64 bits.  All integer
// arithmetic: the sum wraps mod 2^32 like CV_32S whatever the order of the additions, the squared sum is exact.
template <typename Q, bool NT = false>
__device__ __forceinline__ void band_rows_body(const IntegralArgs& a) {
    const uint32_t lane = lane_id();
    const uint32_t band = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t frame = blockIdx.y;
    if (band >= a.n_bands) return;  // whole wave exits together
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    uint32_t* sum = a.sum + (size_t)frame * a.frame_elems;
    uint64_t* sqs = a.sqsum + (size_t)frame * a.frame_elems;
    const uint32_t ow = a.width + 1u;   // output row length
    const uint32_t y0 = band * BAND_ROWS;
    const size_t bo = ((size_t)frame * a.n_bands + band) * a.band_pitch;

    if (band == 0) {  // row 0 of both outputs is zero
        for (uint32_t x = lane; x < ow; x += 64u) {
            sum[x] = 0u;
            sqs[x] = 0ull;
        }
    }
    uint32_t rs[BAND_ROWS];  // per row: total of the row left of the current chunk
    Q rq[BAND_ROWS];
#pragma unroll
    for (int r = 0; r < BAND_ROWS; ++r) {
        rs[r] = 0;
        rq[r] = 0;
    }
    uint32_t top_s = 0;      // integral at the band's top edge, left of the current chunk
    uint64_t top_q = 0;
    for (uint32_t x0 = 0; x0 < a.width; x0 += 256u) {
        const uint32_t x = x0 + lane * 4u;
        const bool in = x < a.width;
        uint32_t v[BAND_ROWS];
#pragma unroll
        for (int r = 0; r < BAND_ROWS; ++r)   // every row of the chunk is in flight before the first scan
            v[r] = in && y0 + r < a.height ? load_gray4(img + (size_t)(y0 + r) * a.gray_stride, x, a.width, a.channels) : 0u;
        uint32_t as[4] = {0, 0, 0, 0};   // running integral of the lane's four columns
        uint64_t aq[4] = {0, 0, 0, 0};
        if (in) {  // column totals above the band
            const uint4 t = *reinterpret_cast<const uint4*>(a.band_sum + bo + x);
            as[0] = t.x; as[1] = as[0] + t.y; as[2] = as[1] + t.z; as[3] = as[2] + t.w;
            const ulonglong2 u0 = *reinterpret_cast<const ulonglong2*>(a.band_sq_prefix + bo + x);
            const ulonglong2 u1 = *reinterpret_cast<const ulonglong2*>(a.band_sq_prefix + bo + x + 2);
            aq[0] = u0.x; aq[1] = aq[0] + u0.y; aq[2] = aq[1] + u1.x; aq[3] = aq[2] + u1.y;
        }
        {
            const uint32_t is = wave_incl_scan(as[3]);
            const uint64_t iq = wave_incl_scan(aq[3]);
            const uint32_t base_s = top_s + (is - as[3]);
            const uint64_t base_q = top_q + (iq - aq[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                as[c] += base_s;
                aq[c] += base_q;
            }
            top_s += wave_last(is);
            top_q += wave_last(iq);
        }
#pragma unroll
        for (int r = 0; r < BAND_ROWS; ++r) {
            const uint32_t y = y0 + r;
            if (y < a.height) {  // uniform
                uint32_t ls[4];
                Q lq[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t p = (v[r] >> (8 * c)) & 0xffu;
                    ls[c] = p + (c ? ls[c - 1] : 0u);
                    lq[c] = (Q)(p * p) + (c ? lq[c - 1] : (Q)0);
                }
                const uint32_t is = wave_incl_scan(ls[3]);
                const Q iq = wave_incl_scan(lq[3]);
                const uint32_t base_s = rs[r] + (is - ls[3]);
                const Q base_q = rq[r] + (iq - lq[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    as[c] += base_s + ls[c];
                    aq[c] += (uint64_t)(Q)(base_q + lq[c]);
                }
                const size_t ro = (size_t)(y + 1u) * ow;
                if (x0 == 0 && lane == 0) {  // column 0 is zero
                    sum[ro] = 0u;
                    sqs[ro] = 0ull;
                }
                if (x + 4u <= a.width) {
                    // 16 (sum) and 32 (squared sum) contiguous bytes per lane: the wave writes
                    // contiguous 1 KiB / 2 KiB runs; rows are only 4-byte aligned (odd stride)
                    out_store<NT>(reinterpret_cast<u32x4_unaligned*>(sum + ro + x + 1u), u32x4_unaligned{as[0], as[1], as[2], as[3]});
                    out_store<NT>(reinterpret_cast<u64x2_unaligned*>(sqs + ro + x + 1u), u64x2_unaligned{aq[0], aq[1]});
                    out_store<NT>(reinterpret_cast<u64x2_unaligned*>(sqs + ro + x + 3u), u64x2_unaligned{aq[2], aq[3]});
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (x + c < a.width) {
                            sum[ro + x + c + 1u] = as[c];
                            sqs[ro + x + c + 1u] = aq[c];
                        }
                }
                rs[r] += wave_last(is);
                rq[r] += wave_last(iq);
            }
        }
    }
}

// Two register budgets of the same body: the compiler's own choice (178-192 VGPRs, two waves per SIMD: the fastest for one
// frame, which is latency-bound) and three waves per SIMD (168 VGPRs, 44-92 bytes of scratch per lane: 8-10 % faster on batches,
// which are bound by stores in flight — 64 x 1080p 0.721 -> 0.664 ms, 256 x 720p 1.325 -> 1.199; four waves per SIMD spill 200-250 bytes
// and lose it again: profiles/r04_notes.md #6).  The batch variant also stores non-temporally (64 x 1080p -7 %, 256 x 720p -6 %; a single
// 4096 x 4096 frame or 8 x 1080p, whose outputs the caches still hold when the cascade starts, +20-30 % with such stores: they keep the ordinary ones).
template <typename Q>
__global__ __launch_bounds__(256) void band_rows(IntegralArgs a) { band_rows_body<Q>(a); }
template <typename Q>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void band_rows_w3(IntegralArgs a) { band_rows_body<Q, true>(a); }

// The same rows with the CHUNKS of a band side by side instead of one after the other: a workgroup of up to eight waves takes one
// band, wave w the 256 columns [x0 + 256 w, x0 + 256 (w + 1)) of every row of the band.  Each wave scans its chunk of the top
// edge and of the eight rows on its own, the chunk totals (9 per wave, u32 + u64) meet in LDS, and a row's base is the totals
// of the chunks to its left — so a band takes one chunk's latency instead of eight chunks' (one 1080p frame: 135 waves, each a
// chain of 8 dependent chunks, were the whole launch), and a wave carries no per-row state from chunk to chunk (fewer registers:
// more bands' stores in flight on a batch).  Images wider than 2048 pixels loop with a carry.  Same integer arithmetic, same
// outputs as band_rows_body.
template <typename Q>
__global__ __launch_bounds__(512) void band_rows_par(IntegralArgs a) {
    __shared__ uint32_t tot_s[2][BAND_ROWS + 1][8];
    __shared__ uint64_t tot_q[2][BAND_ROWS + 1][8];
    const uint32_t lane = lane_id();
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint32_t band = blockIdx.x, frame = blockIdx.y;
    const uint8_t* img = a.gray + (size_t)frame * a.gray_frame_bytes;
    uint32_t* sum = a.sum + (size_t)frame * a.frame_elems;
    uint64_t* sqs = a.sqsum + (size_t)frame * a.frame_elems;
    const uint32_t ow = a.width + 1u;
    const uint32_t y0 = band * BAND_ROWS;
    const size_t bo = ((size_t)frame * a.n_bands + band) * a.band_pitch;
    if (band == 0) {  // row 0 of both outputs is zero
        for (uint32_t x = threadIdx.x; x < ow; x += blockDim.x) {
            sum[x] = 0u;
            sqs[x] = 0ull;
        }
    }
    uint32_t carry_s[BAND_ROWS + 1];   // [0]: the band's top edge, [1 + r]: row r — totals left of the current group of chunks (uniform)
    uint64_t carry_q[BAND_ROWS + 1];
#pragma unroll
    for (int r = 0; r <= BAND_ROWS; ++r) {
        carry_s[r] = 0u;
        carry_q[r] = 0ull;
    }
    uint32_t buf = 0;
    for (uint32_t x0 = 0; x0 < a.width; x0 += 256u * nw, buf ^= 1u) {
        const uint32_t x = x0 + w * 256u + lane * 4u;
        const bool in = x < a.width;
        uint32_t v[BAND_ROWS];
#pragma unroll
        for (int r = 0; r < BAND_ROWS; ++r)
            v[r] = in && y0 + r < a.height ? load_gray4(img + (size_t)(y0 + r) * a.gray_stride, x, a.width, a.channels) : 0u;
        uint32_t as[4] = {0, 0, 0, 0};
        uint64_t aq[4] = {0, 0, 0, 0};
        if (in) {  // column totals above the band
            const uint4 t = *reinterpret_cast<const uint4*>(a.band_sum + bo + x);
            as[0] = t.x; as[1] = as[0] + t.y; as[2] = as[1] + t.z; as[3] = as[2] + t.w;
            const ulonglong2 u0 = *reinterpret_cast<const ulonglong2*>(a.band_sq_prefix + bo + x);
            const ulonglong2 u1 = *reinterpret_cast<const ulonglong2*>(a.band_sq_prefix + bo + x + 2);
            aq[0] = u0.x; aq[1] = aq[0] + u0.y; aq[2] = aq[1] + u1.x; aq[3] = aq[2] + u1.y;
        }
        // this chunk's own scans: exclusive prefixes per lane, totals to LDS
        uint32_t ex_s[BAND_ROWS + 1], t_s[BAND_ROWS + 1];
        uint64_t ex_top_q, t_q[BAND_ROWS + 1];
        Q ex_q[BAND_ROWS];
        {
            const uint32_t is = wave_incl_scan(as[3]);
            const uint64_t iq = wave_incl_scan(aq[3]);
            ex_s[0] = is - as[3];
            ex_top_q = iq - aq[3];
            t_s[0] = wave_last(is);
            t_q[0] = wave_last(iq);
        }
#pragma unroll
        for (int r = 0; r < BAND_ROWS; ++r) {
            uint32_t ls = 0;
            Q lq = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t p = (v[r] >> (8 * c)) & 0xffu;
                ls += p;
                lq += (Q)(p * p);
            }
            const uint32_t is = wave_incl_scan(ls);
            const Q iq = wave_incl_scan(lq);
            ex_s[1 + r] = is - ls;
            ex_q[r] = iq - lq;
            t_s[1 + r] = wave_last(is);
            t_q[1 + r] = (uint64_t)wave_last(iq);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r <= BAND_ROWS; ++r) {
                tot_s[buf][r][w] = t_s[r];
                tot_q[buf][r][w] = t_q[r];
            }
        }
        __syncthreads();   // (one barrier per group: the totals are double-buffered)
        // lane r adds up row r's totals: of the chunks to the left of this wave's, and of the whole group
        uint32_t pre_s = 0, all_s = 0;
        uint64_t pre_q = 0, all_q = 0;
        if (lane <= (uint32_t)BAND_ROWS) {
            for (uint32_t k = 0; k < nw; ++k) {
                const uint32_t ts = tot_s[buf][lane][k];
                const uint64_t tq = tot_q[buf][lane][k];
                if (k < w) {
                    pre_s += ts;
                    pre_q += tq;
                }
                all_s += ts;
                all_q += tq;
            }
        }
        auto lane_u32 = [&](uint32_t val, int r) { return (uint32_t)__builtin_amdgcn_readlane((int)val, r); };
        auto lane_u64 = [&](uint64_t val, int r) { return (uint64_t)lane_u32((uint32_t)val, r) | (uint64_t)lane_u32((uint32_t)(val >> 32), r) << 32; };
        {
            const uint32_t base_s = carry_s[0] + lane_u32(pre_s, 0) + ex_s[0];
            const uint64_t base_q = carry_q[0] + lane_u64(pre_q, 0) + ex_top_q;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                as[c] += base_s;
                aq[c] += base_q;
            }
            carry_s[0] += lane_u32(all_s, 0);
            carry_q[0] += lane_u64(all_q, 0);
        }
#pragma unroll
        for (int r = 0; r < BAND_ROWS; ++r) {
            const uint32_t y = y0 + r;
            if (y < a.height) {  // uniform
                const uint32_t base_s = carry_s[1 + r] + lane_u32(pre_s, 1 + r) + ex_s[1 + r];
                const Q base_q = (Q)(carry_q[1 + r] + lane_u64(pre_q, 1 + r)) + ex_q[r];
                uint32_t ls = 0;
                Q lq = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t p = (v[r] >> (8 * c)) & 0xffu;
                    ls += p;
                    lq += (Q)(p * p);
                    as[c] += base_s + ls;
                    aq[c] += (uint64_t)(Q)(base_q + lq);
                }
                const size_t ro = (size_t)(y + 1u) * ow;
                if (x == 0u) {  // column 0 is zero
                    sum[ro] = 0u;
                    sqs[ro] = 0ull;
                }
                if (x + 4u <= a.width) {
                    out_store<false>(reinterpret_cast<u32x4_unaligned*>(sum + ro + x + 1u), u32x4_unaligned{as[0], as[1], as[2], as[3]});
                    out_store<false>(reinterpret_cast<u64x2_unaligned*>(sqs + ro + x + 1u), u64x2_unaligned{aq[0], aq[1]});
                    out_store<false>(reinterpret_cast<u64x2_unaligned*>(sqs + ro + x + 3u), u64x2_unaligned{aq[2], aq[3]});
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (x + c < a.width) {
                            sum[ro + x + c + 1u] = as[c];
                            sqs[ro + x + c + 1u] = aq[c];
                        }
                }
                carry_s[1 + r] += lane_u32(all_s, 1 + r);
                carry_q[1 + r] += lane_u64(all_q, 1 + r);
            }
        }
    }
}

int launch_integral(const IntegralArgs& a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t quads = a.band_pitch / 4u;
    dim3 g1((quads + 255u) / 256u, a.n_bands, a.n_frames);
    hipLaunchKernelGGL(band_colsum, g1, dim3(256), 0, stream, a);
    // prefix over the bands: two levels (16 groups per workgroup) while a group stays <= 32 bands (images up to 4096 rows)
    const dim3 g2b((a.band_pitch + 63u) / 64u, a.n_frames, 1);
    if (a.n_bands <= 16u * 8u) hipLaunchKernelGGL(band_scan2<8>, g2b, dim3(1024), 0, stream, a);
    else if (a.n_bands <= 16u * 16u) hipLaunchKernelGGL(band_scan2<16>, g2b, dim3(1024), 0, stream, a);
    else if (a.n_bands <= 16u * 32u) hipLaunchKernelGGL(band_scan2<32>, g2b, dim3(1024), 0, stream, a);
    else {
        dim3 g2((a.band_pitch + 255u) / 256u, a.n_frames, 1);
        hipLaunchKernelGGL(band_scan, g2, dim3(256), 0, stream, a);
    }
    if (a.rows_mode != 0u) {   // chunks of a band side by side (band_rows_par): 1 always, 2 for calls that are not batches
        const bool batch_p = (uint64_t)a.n_frames * a.n_bands >= 2048u;
        if (a.rows_mode == 1u || !batch_p) {
            const uint32_t nw = min(8u, (a.width + 255u) / 256u);
            dim3 gp(a.n_bands, a.n_frames, 1);
            if ((uint64_t)a.width * 65025ull < (1ull << 32)) hipLaunchKernelGGL(band_rows_par<uint32_t>, gp, dim3(64u * nw), 0, stream, a);
            else hipLaunchKernelGGL(band_rows_par<uint64_t>, gp, dim3(64u * nw), 0, stream, a);
            return (int)hipGetLastError();
        }
    }
    dim3 g3((a.n_bands + 3u) / 4u, a.n_frames, 1);
    const bool batch = (uint64_t)a.n_frames * a.n_bands >= 2048u;   // enough bands in flight to fill the chip several times
    if ((uint64_t)a.width * 65025ull < (1ull << 32)) {
        if (batch) hipLaunchKernelGGL(band_rows_w3<uint32_t>, g3, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(band_rows<uint32_t>, g3, dim3(256), 0, stream, a);
    } else {
        if (batch) hipLaunchKernelGGL(band_rows_w3<uint64_t>, g3, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(band_rows<uint64_t>, g3, dim3(256), 0, stream, a);
    }
    return (int)hipGetLastError();
}

// ====================================================================== cascade

// Window position of grid index i: precomputeWindows' lrint(i * step) (clod.cpp:514, half to even; the contract of the
// OpenCL path) or the plain CPU loop's round(i * step) (clod.cpp:1416, half away from zero; VJ_FLAG_SKIP_ROW).  The
// block variant keeps `step` as a double (clod.cpp:862), so its product is the f64 one — exact, where the f32 product
// is rounded to 24 bits first and may land on the other side of a half (VJ_FLAG_GRID_F64; lrint at :941-942, round()
// at :1034): those positions come from a per-scale table the host fills with the reference's own f64 expression
// (pos_base != 0; f64 arithmetic here cost the tile kernel a register allocation granule and the gather chain next to
// it 2 ms of config 4, profiles/r03_notes.md #9).  pos_mode / pos_base are wave-uniform.
__device__ __forceinline__ uint32_t window_pos(const CascadeArgs& a, uint32_t pos_base, uint32_t i, float step) {
    if (pos_base != 0u) return a.pos_tab[pos_base + i];
    const float v = (float)i * step;
    return (a.pos_mode & 1u) ? (uint32_t)roundf(v) : (uint32_t)__float2int_rn(v);
}

// P2 skip modes: is grid window (ix, iy) of this scale one the reference's sequential loop visits?
__device__ __forceinline__ bool window_visited(const CascadeArgs& a, uint32_t frame, uint32_t skip_base, uint32_t skip_wpr,
                                               uint32_t nx, uint32_t ix, uint32_t iy) {
    uint32_t word, bit;
    if (skip_wpr != 0u) {
        word = skip_base + iy * skip_wpr + (ix >> 6);
        bit = ix & 63u;
    } else {
        const uint32_t i = iy * nx + ix;
        word = skip_base + (i >> 6);
        bit = i & 63u;
    }
    return ((a.skip_bits[(size_t)frame * a.skip_frame_words + word] >> bit) & 1ull) != 0ull;
}

// computeVariance (clod.cpp:418-446) for the window whose origin is element `e` of the
// frame described by (sum_f, sq_f).
__device__ __forceinline__ float window_variance(rsrc_t sum_f, rsrc_t sq_f, uint32_t e, uint32_t e_lt, uint32_t e_dw,
                                                 uint32_t e_dh, float area, bool signed_mean) {
    // corner offsets are uniform (scalar adds); the lane contributes e
    const uint32_t c0 = e_lt, c1 = e_lt + e_dw, c2 = e_lt + e_dh, c3 = e_lt + e_dh + e_dw;
    const uint32_t s = ld_u32(sum_f, e * 4u, c0 * 4u) - ld_u32(sum_f, e * 4u, c1 * 4u) - ld_u32(sum_f, e * 4u, c2 * 4u) +
                       ld_u32(sum_f, e * 4u, c3 * 4u);
    const uint64_t q = ld_u64(sq_f, e * 8u, c0 * 8u) - ld_u64(sq_f, e * 8u, c1 * 8u) - ld_u64(sq_f, e * 8u, c2 * 8u) +
                       ld_u64(sq_f, e * 8u, c3 * 8u);
    const float mean = (signed_mean ? (float)(int32_t)s : (float)s) / area;
    float variance = (float)q;                       // u64 -> f32, round to nearest even
    variance = (variance / area) - (mean * mean);    // separate divide, multiply, subtract
    return variance >= 0.0f ? sqrtf(variance) : 1.0f;
}

// Where a window's corner values come from.  Both sources take the lane's window
// offset (bytes) plus a wave-uniform corner offset (bytes).
//  * GlobalImg: the batch sum image in HBM/L2 through buffer loads (no VALU address
//    arithmetic; every lane is its own request in the texture-address unit, which is
//    what bounds this path: ~40-64 cycles per wave-load, measured).
//  * LdsImg: a tile of the sum image staged in LDS (2-4 cycles per wave-load).
// The stump-parallel finish turns the roles around — the window offset is wave-uniform, the corner
// offset differs per lane — which only matters for buffer loads (the scalar operand must be the
// uniform one): `by_stump()` gives that view.
struct GlobalImgByStump {
    rsrc_t r;
    __device__ __forceinline__ uint32_t ld(uint32_t uni_window_off, uint32_t lane_corner_off) const {
        return ld_u32(r, lane_corner_off, __builtin_amdgcn_readfirstlane(uni_window_off));
    }
};
struct GlobalImg {
    rsrc_t r;
    __device__ __forceinline__ uint32_t ld(uint32_t lane_off, uint32_t uni_off) const { return ld_u32(r, lane_off, uni_off); }
    __device__ __forceinline__ GlobalImgByStump by_stump() const { return GlobalImgByStump{r}; }
};
struct LdsImg {
    const char* base;  // LDS
    __device__ __forceinline__ uint32_t ld(uint32_t lane_off, uint32_t uni_off) const {
        return *reinterpret_cast<const uint32_t*>(base + (lane_off + uni_off));
    }
    __device__ __forceinline__ LdsImg by_stump() const { return *this; }
};

// The weighted rectangle sums of one node (clod.cl:60-76) for the lane's window.
template <typename Img>
__device__ __forceinline__ float node_rect_sum(const Img& img, const NodeRecDev& r, uint32_t off) {
    const uint32_t lt0 = r[0], lt1 = r[1], lt2 = r[2];
    const uint32_t dh0 = r[3], dh1 = r[4], dh2 = r[5];
    // left->right distances are signed 16-bit (negative only in de-interleaved LDS tiles)
    const uint32_t dw0 = (uint32_t)(int32_t)(int16_t)(r[6] & 0xffffu), dw1 = (uint32_t)((int32_t)r[6] >> 16),
                   dw2 = (uint32_t)(int32_t)(int16_t)(r[7] & 0xffffu);
    const float w0 = __uint_as_float(r[8]), w1 = __uint_as_float(r[9]), w2 = __uint_as_float(r[10]);
    // u32 wrap-around on the four corners, one cast, one multiply per rectangle
    const uint32_t r0 = img.ld(off, lt0) - img.ld(off, lt0 + dw0) - img.ld(off, lt0 + dh0) + img.ld(off, lt0 + dh0 + dw0);
    const uint32_t r1 = img.ld(off, lt1) - img.ld(off, lt1 + dw1) - img.ld(off, lt1 + dh1) + img.ld(off, lt1 + dh1 + dw1);
    // rect_sum = 0; rect_sum += t0; — the leading "0 +" only maps -0 to +0, which no
    // comparison or later sum can observe, so it is elided.
    float rect_sum = (float)r0 * w0;
    rect_sum += (float)r1 * w1;
    if (w2 != 0.0f) {  // uniform branch (clod.cl:70)
        const uint32_t r2 =
            img.ld(off, lt2) - img.ld(off, lt2 + dw2) - img.ld(off, lt2 + dh2) + img.ld(off, lt2 + dh2 + dw2);
        rect_sum += (float)r2 * w2;
    }
    return rect_sum;
}

template <typename Img>
__device__ __forceinline__ void node_rect_sum_pair(const Img& img, const NodeRecDev& ra, const NodeRecDev& rb, uint32_t off,
                                                   float& sum_a, float& sum_b);
// (COUNT: cnt[c] += {nodes below the root that window c's walk visits, their rectangles} — the oracle's stump_evals / rect_evals
// count visited nodes; the roots are every entering window's and are priced on the host)
template <int NC, bool COUNT = false, typename Img>
__device__ __forceinline__ void stage_sum_tree2_multi(const Img& img, kptr<NodeRecDev> tab, uint32_t n_trees,
                                                      const uint32_t (&off)[NC], const float (&var)[NC],
                                                      float (&stage_sum)[NC], uint32_t (*cnt)[2] = nullptr);   // (defined with the tile kernel's sweeps)

// Adds the lanes' visited-node counts of a counted sweep to CascadeArgs::tree_ctr (lanes outside the population pass 0).
__device__ __forceinline__ void tree_count_flush(const CascadeArgs& a, uint32_t nodes, uint32_t rects, uint32_t lane) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nodes += (uint32_t)__shfl_xor((int)nodes, d, 64);
        rects += (uint32_t)__shfl_xor((int)rects, d, 64);
    }
    if (lane == 0 && (nodes | rects) != 0u) {
        atomicAdd(a.tree_ctr, (unsigned long long)nodes);
        atomicAdd(a.tree_ctr + 1, (unsigned long long)rects);
    }
}

// One stump-based stage on one window (clod.cl:49-82).  `tab` points at the stage's
// first node record of the wave's scale; every table value is wave-uniform.  Two stumps per step: their 16 (24)
// gathers are issued together, so a wave that walks a stage alone — the thin late stages of the queue passes, a
// single frame — pays one memory round trip per PAIR instead of one or two per stump (ISA before: 8 loads,
// s_waitcnt vmcnt(0), 4 loads, s_waitcnt vmcnt(0) per stump); the values are still added in stump order.
template <typename Img>
__device__ __forceinline__ float stage_sum_stumps(const Img& img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off,
                                                  float var, bool pairs = true) {
    float stage_sum = 0.0f;
    uint32_t j = 0;
    if (!pairs) {   // uniform: one stump per step (fewer gathers in flight: gentler on the tile chain next door)
        NodeRecDev r = tab[0];
        for (; j < n_nodes; ++j) {
            const NodeRecDev rn = tab[j + 1 < n_nodes ? j + 1 : j];
            const float rect_sum = node_rect_sum(img, r, off);
            stage_sum += (rect_sum >= __uint_as_float(r[11]) * var) ? __uint_as_float(r[13]) : __uint_as_float(r[12]);
            r = rn;
        }
        return stage_sum;
    }
    if (n_nodes >= 2u) {
        NodeRecDev ra = tab[0], rb = tab[1];
        for (; j + 1u < n_nodes; j += 2u) {
            // fetch the next pair while this one is evaluated (scalar loads are long)
            const uint32_t ja = j + 2u < n_nodes ? j + 2u : j, jb = j + 3u < n_nodes ? j + 3u : j + 1u;
#if VJ_SCHED_PREFETCH
            NodeRecDev na = rec_fetch(tab + ja), nb = rec_fetch(tab + jb);
#else
            const NodeRecDev na = tab[ja], nb = tab[jb];
#endif
            float sa, sb;
            node_rect_sum_pair(img, ra, rb, off, sa, sb);
            // alpha[rect_sum >= norm_threshold]: alpha[0] = left_val, alpha[1] = right_val
            stage_sum += (sa >= __uint_as_float(ra[11]) * var) ? __uint_as_float(ra[13]) : __uint_as_float(ra[12]);
            stage_sum += (sb >= __uint_as_float(rb[11]) * var) ? __uint_as_float(rb[13]) : __uint_as_float(rb[12]);
#if VJ_SCHED_PREFETCH
            rec_arrived(na, nb);
#endif
            ra = na;
            rb = nb;
        }
    }
    if (j < n_nodes) {   // odd tail
        const NodeRecDev r = tab[j];
        const float rect_sum = node_rect_sum(img, r, off);
        stage_sum += (rect_sum >= __uint_as_float(r[11]) * var) ? __uint_as_float(r[13]) : __uint_as_float(r[12]);
    }
    return stage_sum;
}

// The same stage on NC chunks of 64 windows at once (lane l holds window l of every chunk): the
// record fetch, its wait and the scalar corner arithmetic are paid once per stump instead of once per
// chunk, and the NC independent gather groups overlap each other's LDS latency.  Per window the
// operations and their order are exactly those of stage_sum_stumps.
template <int NC, typename Img>
__device__ __forceinline__ void stage_sum_stumps_multi(const Img& img, kptr<NodeRecDev> tab, uint32_t n_nodes,
                                                       const uint32_t (&off)[NC], const float (&var)[NC],
                                                       float (&stage_sum)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) stage_sum[c] = 0.0f;
    NodeRecDev r = tab[0];
    for (uint32_t j = 0; j < n_nodes; ++j) {
#if VJ_SCHED_PREFETCH
        NodeRecDev rn = rec_fetch(tab + (j + 1 < n_nodes ? j + 1 : j));
#else
        const NodeRecDev rn = tab[j + 1 < n_nodes ? j + 1 : j];
#endif
        const float thr = __uint_as_float(r[11]), left = __uint_as_float(r[12]), right = __uint_as_float(r[13]);
        // node_rect_sum, rectangle-major across the chunks so that all 8*NC gathers of the first two
        // rectangles are in flight together (the uniform third-rectangle branch would otherwise cut the
        // chunks apart)
        const uint32_t lt0 = r[0], lt1 = r[1], lt2 = r[2];
        const uint32_t dh0 = r[3], dh1 = r[4], dh2 = r[5];
        const uint32_t dw0 = (uint32_t)(int32_t)(int16_t)(r[6] & 0xffffu), dw1 = (uint32_t)((int32_t)r[6] >> 16),
                       dw2 = (uint32_t)(int32_t)(int16_t)(r[7] & 0xffffu);
        const float w0 = __uint_as_float(r[8]), w1 = __uint_as_float(r[9]), w2 = __uint_as_float(r[10]);
        uint32_t c0[NC][4], c1[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            c0[c][0] = img.ld(off[c], lt0);
            c0[c][1] = img.ld(off[c], lt0 + dw0);
            c0[c][2] = img.ld(off[c], lt0 + dh0);
            c0[c][3] = img.ld(off[c], lt0 + dh0 + dw0);
            c1[c][0] = img.ld(off[c], lt1);
            c1[c][1] = img.ld(off[c], lt1 + dw1);
            c1[c][2] = img.ld(off[c], lt1 + dh1);
            c1[c][3] = img.ld(off[c], lt1 + dh1 + dw1);
        }
        float rect_sum[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t r0 = c0[c][0] - c0[c][1] - c0[c][2] + c0[c][3];
            const uint32_t r1 = c1[c][0] - c1[c][1] - c1[c][2] + c1[c][3];
            rect_sum[c] = (float)r0 * w0;
            rect_sum[c] += (float)r1 * w1;
        }
        if (w2 != 0.0f) {  // uniform branch (clod.cl:70)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                c0[c][0] = img.ld(off[c], lt2);
                c0[c][1] = img.ld(off[c], lt2 + dw2);
                c0[c][2] = img.ld(off[c], lt2 + dh2);
                c0[c][3] = img.ld(off[c], lt2 + dh2 + dw2);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const uint32_t r2 = c0[c][0] - c0[c][1] - c0[c][2] + c0[c][3];
                rect_sum[c] += (float)r2 * w2;
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) stage_sum[c] += (rect_sum[c] >= thr * var[c]) ? right : left;
#if VJ_SCHED_PREFETCH
        rec_arrived(rn);
#endif
        r = rn;
    }
}

// Multi-node trees: icvEvalHidHaarClassifier's walk (tempcv.cpp:771-792) on the clod
// f32 arithmetic.  Nodes of a tree are stored consecutively and a child always has a
// larger index than its parent, so a tree is evaluated by visiting its records in
// order, each with the lanes whose walk currently sits on it; the table stays uniform.
template <bool COUNT = false, typename Img>
__device__ __forceinline__ float stage_sum_trees(const Img& img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off,
                                                 float var, uint32_t* cnt = nullptr) {
    float stage_sum = 0.0f;
    uint32_t cur = 0;     // node (inside the current tree) this lane evaluates next
    uint32_t k = 0;       // position of the record inside its tree (uniform)
    float value = 0.0f;
    bool done = false;
    for (uint32_t j = 0; j < n_nodes; ++j) {
        const NodeRecDev r = tab[j];
        const uint32_t flags = r[7] >> 16;
        if (!done && cur == k) {
            if (COUNT && k != 0u) {
                cnt[0] += 1u;
                cnt[1] += __uint_as_float(r[10]) != 0.0f ? 3u : 2u;
            }
            const float t = __uint_as_float(r[11]) * var;
            const float sum = node_rect_sum(img, r, off);
            const bool go_left = sum < t;  // idx = sum < t ? left : right
            const uint32_t nxt = go_left ? r[12] : r[13];
            const bool is_node = go_left ? (flags & 1u) != 0u : (flags & 2u) != 0u;
            if (is_node) {
                cur = nxt;
            } else {
                value = __uint_as_float(nxt);
                done = true;
            }
        }
        ++k;
        if (flags & 4u) {  // last record of the tree (uniform)
            stage_sum += value;
            cur = 0;
            k = 0;
            done = false;
        }
    }
    return stage_sum;
}

template <bool TREES, bool COUNT = false, typename Img>
__device__ __forceinline__ float stage_sum_of(const Img& img, kptr<NodeRecDev> tab, uint32_t n_nodes, uint32_t off,
                                              float var, bool pairs = true, uint32_t* cnt = nullptr) {
    if (TREES) return stage_sum_trees<COUNT>(img, tab, n_nodes, off, var, cnt);
    return stage_sum_stumps(img, tab, n_nodes, off, var, pairs);
}

// Compacting stage sweep shared by every pass: runs stages [a.stage_begin, a.stage_end)
// over the wave's LDS queue q[0..n) (all entries belong to one scale and sit at the same
// stage — linear cascades), compacting survivors in place after every stage.  Returns
// the number of survivors left at q[0..).
// Where the windows a stage REJECTS go, for the segments of a stage tree whose failures continue in another
// chain (frontalface_alt_tree: failing anywhere in the first chain after stage 4 starts the second one).
struct FailSink {
    QEntry* base = nullptr;      // the (scale, part) sub-queue of the chain that takes them
    uint32_t* count = nullptr;
};

// Thin-wave tail of a global-gather sweep (stump cascades).  Once a wave is down to a handful of windows, walking a
// stage stump by stump costs one scalar record fetch and one memory round trip per stump (pair) with 1-8 of its 64
// lanes busy — a late stage of 100-400 stumps takes 0.1-1 ms per wave whatever the population, which is what bounds a
// single frame and the late passes of a stage tree.  So the roles are turned around: lane j takes stump j of a block of 64
// (its record arrives with four coalesced 16-byte loads), every window is evaluated by all lanes at once (the window
// offset is wave-uniform, the corner offsets differ per lane: GlobalImgByStump), a __ballot gives the block's verdict
// bits, and afterwards lane w adds window w's leaf values IN STUMP ORDER from the bits — the reference's sequence of f32
// additions (clod.cl:81; the leaf values come through the scalar cache).  Scratch: the unused part of the wave's LDS
// queue (the verdict words).  Per stage the cost is windows x blocks instead of stumps, so it pays below ~64 windows.
constexpr uint32_t SP_TAIL_MAX = 48;          // windows at most: 48 entries (384 B) + 48 x 9 verdict words (3456 B) fit the wave's 4 KiB queue
constexpr uint32_t SP_TAIL_MAX_BLOCKS = 9;    // blocks of 64 stumps per stage at most (stages of <= 576 nodes)
constexpr uint32_t SP_TAIL_MAX_NODES = SP_TAIL_MAX_BLOCKS * 64u;
#ifndef VJ_TAIL_NW
#define VJ_TAIL_NW 2
#endif
constexpr int TAIL_NW = VJ_TAIL_NW;            // windows per step of the tail in the queue passes of small batches (CascadeArgs::wide_tail)

template <bool COUNT, int NW = 1>
__device__ __forceinline__ uint32_t sweep_tail_stump_parallel(const CascadeArgs& a, rsrc_t img_r, kptr<NodeRecDev> table, QEntry* q, uint32_t n,
                                                              uint32_t lane, uint32_t begin, uint32_t end, FailSink fail) {
    kptr<StageDev> stages = as_k(a.stages);
    const GlobalImgByStump img{img_r};
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(q + SP_TAIL_MAX);          // [window][block]
    for (uint32_t pos = begin; pos < end && n != 0u; ++pos) {
        const uint32_t s = a.identity_order != 0u ? pos : stages[pos].order;
        const uint32_t first_node = stages[s].first_node, n_nodes = stages[s].n_nodes;
        const float threshold = stages[s].threshold;
        if (COUNT && lane == 0) atomicAdd(a.stage_entered + s, (unsigned long long)n);
        const uint32_t n_blocks = (n_nodes + 63u) >> 6;
        const unsigned long long t_a = VJ_STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
        // A: verdict bits, lanes = stumps
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const uint32_t j = b * 64u + lane;
            const bool active = j < n_nodes;
            const uint4* rp = reinterpret_cast<const uint4*>((uintptr_t)(table + first_node + (active ? j : 0u)));
            const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
            NodeRecDev r;
            r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w;
            r[8] = r2.x; r[9] = r2.y; r[10] = r2.z; r[11] = r2.w; r[12] = r3.x; r[13] = r3.y; r[14] = r3.z; r[15] = r3.w;
            const float thr_node = __uint_as_float(r[11]);
            uint32_t w = 0;
            if constexpr (NW > 1) {
                // NW windows per step, all of their gathers in flight before the first verdict (instantiated for the queue
                // passes of small batches: a cluster of detections sits in ONE wave, which then pays windows x blocks
                // memory round trips per stage)
                const uint32_t lt0 = r[0], lt1 = r[1], lt2 = r[2], dh0 = r[3], dh1 = r[4], dh2 = r[5];
                const uint32_t dw0 = (uint32_t)(int32_t)(int16_t)(r[6] & 0xffffu), dw1 = (uint32_t)((int32_t)r[6] >> 16),
                               dw2 = (uint32_t)(int32_t)(int16_t)(r[7] & 0xffffu);
                const float w0 = __uint_as_float(r[8]), w1 = __uint_as_float(r[9]), w2 = __uint_as_float(r[10]);
                for (; w + (uint32_t)NW <= n; w += (uint32_t)NW) {
                    uint32_t c[NW][12];
                    float var[NW];
#pragma unroll
                    for (int k = 0; k < NW; ++k) {   // (an absent third rectangle reads four in-range dwords)
                        const QEntry e = q[w + (uint32_t)k];
                        var[k] = e.var;
                        c[k][0] = img.ld(e.off, lt0); c[k][1] = img.ld(e.off, lt0 + dw0); c[k][2] = img.ld(e.off, lt0 + dh0); c[k][3] = img.ld(e.off, lt0 + dh0 + dw0);
                        c[k][4] = img.ld(e.off, lt1); c[k][5] = img.ld(e.off, lt1 + dw1); c[k][6] = img.ld(e.off, lt1 + dh1); c[k][7] = img.ld(e.off, lt1 + dh1 + dw1);
                        c[k][8] = img.ld(e.off, lt2); c[k][9] = img.ld(e.off, lt2 + dw2); c[k][10] = img.ld(e.off, lt2 + dh2); c[k][11] = img.ld(e.off, lt2 + dh2 + dw2);
                    }
#pragma unroll
                    for (int k = 0; k < NW; ++k) {   // node_rect_sum's arithmetic (clod.cl:60-76)
                        float rect_sum = (float)(c[k][0] - c[k][1] - c[k][2] + c[k][3]) * w0;
                        rect_sum += (float)(c[k][4] - c[k][5] - c[k][6] + c[k][7]) * w1;
                        const float with2 = rect_sum + (float)(c[k][8] - c[k][9] - c[k][10] + c[k][11]) * w2;
                        rect_sum = w2 != 0.0f ? with2 : rect_sum;
                        const unsigned long long m = __ballot(active && rect_sum >= thr_node * var[k]);
                        if (lane == 0) masks[(w + (uint32_t)k) * SP_TAIL_MAX_BLOCKS + b] = m;
                    }
                }
            }
            for (; w < n; ++w) {
                const QEntry e = q[w];   // broadcast
                const float sum = node_rect_sum(img, r, e.off);
                const unsigned long long m = __ballot(active && sum >= thr_node * e.var);
                if (lane == 0) masks[w * SP_TAIL_MAX_BLOCKS + b] = m;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long t_b = VJ_STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
        // B: lanes = windows; leaf values added in stump order
        const bool have = lane < n;
        const QEntry mine = q[have ? lane : 0u];
        float stage_sum = 0.0f;
        kptr<uint32_t> leaf = reinterpret_cast<kptr<uint32_t>>(table + first_node);   // record k: dwords 12 / 13 = left / right value
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const unsigned long long m = masks[(have ? lane : 0u) * SP_TAIL_MAX_BLOCKS + b];
            const uint32_t jn = min(64u, n_nodes - b * 64u);
#pragma unroll 4
            for (uint32_t k = 0; k < jn; ++k) {
                const uint32_t j = b * 64u + k;
                const float l = __uint_as_float(leaf[j * 16u + 12u]), rr = __uint_as_float(leaf[j * 16u + 13u]);
                stage_sum += ((m >> k) & 1ull) != 0ull ? rr : l;
            }
        }
        const bool pass = have && stage_sum >= threshold;
        if (fail.base != nullptr) {   // uniform: this segment's rejects continue elsewhere
            const unsigned long long fm = __ballot(have && !pass);
            if (fm != 0ull) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(fail.count, (uint32_t)__popcll(fm));
                g = __builtin_amdgcn_readfirstlane(g);
                if (have && !pass) fail.base[g + mbcnt(fm)] = mine;
            }
        }
        const unsigned long long pm = __ballot(pass);
        __builtin_amdgcn_wave_barrier();   // every lane holds its entry and has read its masks
        if (pass) q[mbcnt(pm)] = mine;
        if (VJ_STAMPS && lane == 0) {   // diagnostic build: verdict phase, leaf-sum phase, (window, block) pairs, stages
            const unsigned long long t_e = __builtin_amdgcn_s_memtime();
            atomicAdd(a.stage_entered + 51, t_b - t_a);
            atomicAdd(a.stage_entered + 52, t_e - t_b);
            atomicAdd(a.stage_entered + 53, (unsigned long long)n * n_blocks);
            atomicAdd(a.stage_entered + 54, 1ull);
        }
        n = (uint32_t)__popcll(pm);
        __builtin_amdgcn_wave_barrier();
    }
    return n;
}

template <bool TREES, bool COUNT, bool MULTI = false, int NW = 1, typename Img>
__device__ __forceinline__ uint32_t sweep_stages(const CascadeArgs& a, const Img& img, kptr<NodeRecDev> table,
                                                 QEntry* q, uint32_t n, uint32_t lane, uint32_t begin, uint32_t end,
                                                 FailSink fail = FailSink()) {
    kptr<StageDev> stages = as_k(a.stages);
    for (uint32_t pos = begin; pos < end && n != 0u; ++pos) {
        // [begin, end) are positions in the sweep order (StageDev::order); a linear cascade's order is 0, 1, 2, ...
        const uint32_t s = a.identity_order != 0u ? pos : stages[pos].order;   // (no dependent load for linear cascades)
        const uint32_t first_node = stages[s].first_node;
        const uint32_t n_nodes = stages[s].n_nodes;
        if constexpr (!TREES && !MULTI && std::is_same<Img, GlobalImg>::value) {
            // a handful of windows left: the rest of the sweep stump-parallel (uniform decision)
            if (a.sp_tail_max != 0u && n <= a.sp_tail_max && a.max_stage_nodes <= SP_TAIL_MAX_NODES)
                return sweep_tail_stump_parallel<COUNT, NW>(a, img.r, table, q, n, lane, pos, end, fail);
        }
        const float threshold = stages[s].threshold;
        if (COUNT && lane == 0) atomicAdd(a.stage_entered + s, (unsigned long long)n);
        kptr<NodeRecDev> tab = table + first_node;
        uint32_t m = 0;
        uint32_t base = 0;
        if (TREES && MULTI && a.tree2) {
            // two-node trees: pairs of chunks share the record fetches
            while (base + 64u < n) {
                QEntry e[2];
                uint32_t off[2];
                float var[2], sum[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint32_t i = base + (uint32_t)c * 64u + lane;
                    e[c] = q[i < n ? i : 0u];
                    off[c] = e[c].off;
                    var[c] = e[c].var;
                }
                uint32_t cnt2[2][2] = {{0u, 0u}, {0u, 0u}};
                stage_sum_tree2_multi<2, COUNT>(img, tab, n_nodes >> 1, off, var, sum, cnt2);
                if (COUNT) {
                    const bool a0 = base + lane < n, a1 = base + 64u + lane < n;
                    tree_count_flush(a, (a0 ? cnt2[0][0] : 0u) + (a1 ? cnt2[1][0] : 0u), (a0 ? cnt2[0][1] : 0u) + (a1 ? cnt2[1][1] : 0u), lane);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool pass = base + (uint32_t)c * 64u + lane < n && sum[c] >= threshold;
                    const unsigned long long mask = __ballot(pass);
                    if (pass) q[m + mbcnt(mask)] = e[c];
                    m += (uint32_t)__popcll(mask);
                }
                __builtin_amdgcn_wave_barrier();
                base += 128u;
            }
        }
        if (!TREES && MULTI) {
            // groups of 4, then 2 full-or-partial chunks; a last single chunk falls through to the loop below
            auto group = [&](auto nc_tag) {
                constexpr int NC = decltype(nc_tag)::value;
                QEntry e[NC];
                uint32_t off[NC];
                float var[NC], sum[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const uint32_t i = base + (uint32_t)c * 64u + lane;
                    e[c] = q[i < n ? i : 0u];
                    off[c] = e[c].off;
                    var[c] = e[c].var;
                }
                stage_sum_stumps_multi<NC>(img, tab, n_nodes, off, var, sum);
                __builtin_amdgcn_wave_barrier();   // every entry of the group is in registers
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const bool pass = base + (uint32_t)c * 64u + lane < n && sum[c] >= threshold;
                    const unsigned long long mask = __ballot(pass);
                    if (pass) q[m + mbcnt(mask)] = e[c];
                    m += (uint32_t)__popcll(mask);
                }
                __builtin_amdgcn_wave_barrier();
                base += (uint32_t)NC * 64u;
            };
            while (base + 192u < n) group(std::integral_constant<int, 4>{});
            if (base + 64u < n) group(std::integral_constant<int, 2>{});
        }
        for (; base < n; base += 64u) {
            const uint32_t i = base + lane;
            const bool act = i < n;
            const QEntry e = q[act ? i : 0u];
            bool pass = false;
            uint32_t cnt1[1][2] = {{0u, 0u}};
            if (act) {
                if (TREES && a.tree2) {   // two-node trees: both nodes' gathers in flight (tile sweeps and global-gather sweeps alike)
                    const uint32_t off1[1] = {e.off};
                    const float var1[1] = {e.var};
                    float sum1[1];
                    stage_sum_tree2_multi<1, COUNT>(img, tab, n_nodes >> 1, off1, var1, sum1, cnt1);
                    pass = sum1[0] >= threshold;
                } else {
                    pass = stage_sum_of<TREES, COUNT>(img, tab, n_nodes, e.off, e.var, a.gather_pairs == 2u || (a.gather_pairs == 1u && n <= 64u), cnt1[0]) >= threshold;
                }
            }
            if (TREES && COUNT) tree_count_flush(a, cnt1[0][0], cnt1[0][1], lane);
            if (!MULTI && fail.base != nullptr) {   // uniform: this segment's rejects continue elsewhere
                const unsigned long long fm = __ballot(act && !pass);
                if (fm != 0ull) {
                    uint32_t g = 0;
                    if (lane == 0) g = atomicAdd(fail.count, (uint32_t)__popcll(fm));
                    g = __builtin_amdgcn_readfirstlane(g);
                    if (act && !pass) fail.base[g + mbcnt(fm)] = e;
                }
            }
            const unsigned long long mask = __ballot(pass);
            __builtin_amdgcn_wave_barrier();   // every lane has read its entry before any lane overwrites
            if (pass) q[m + mbcnt(mask)] = e;  // m + rank <= i: never ahead of the read cursor
            m += (uint32_t)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
        }
        n = m;
    }
    return n;
}

// Which part of a scale's queue segment the windows of `frame` go to: frames are grouped in order, at most
// ceil(n_frames / Q_PARTS) per part (the host sizes the parts for that), so that a part holds few frames — the
// working set of whoever drains it.
__device__ __forceinline__ uint32_t frame_part(const CascadeArgs& a, uint32_t frame) {
    return a.n_frames >= Q_PARTS ? (uint32_t)((unsigned long long)frame * Q_PARTS / a.n_frames) : frame;
}

// Global-gather pass body: sweep the stages, then hand the survivors to the next pass's
// global queue (or to the detection list).
template <bool TREES, bool LAST, bool COUNT, int NW = 1>
__device__ __forceinline__ void run_stages_linear(const CascadeArgs& a, rsrc_t img_r, QEntry* q, uint32_t n,
                                                  uint32_t scale_slot, uint32_t table_first, uint32_t q_base,
                                                  uint32_t lane, uint32_t begin, uint32_t part, uint32_t run_idx = 0xffffffffu) {
    kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + table_first;
    const GlobalImg img{img_r};
    FailSink fail;
    if (a.q_fail != nullptr) {
        fail.base = a.q_fail + ((size_t)q_base + (size_t)part * as_k(a.scales)[scale_slot].q_cap);
        fail.count = a.q_fail_count + scale_slot * Q_PARTS + part;
    }
    n = sweep_stages<TREES, COUNT, false, NW>(a, img, table, q, n, lane, begin, a.stage_end, fail);
    if (n == 0u) {
        if (!LAST && run_idx != 0xffffffffu && lane == 0) *reinterpret_cast<uint2*>(a.run_table + 2u * (size_t)run_idx) = make_uint2(0u, 0u);
        return;
    }
    if (LAST) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(a.det_count, n);
        g = __builtin_amdgcn_readfirstlane(g);
        for (uint32_t i = lane; i < n; i += 64u)
            if (g + i < a.det_cap) a.det[g + i] = DetEntry{q[i].off, scale_slot};
    } else {
        uint32_t g = 0;
        // the scale's queue segment has Q_PARTS parts of q_part_cap entries: one per group of frames
        if (lane == 0) g = atomicAdd(a.q_out_count + scale_slot * Q_PARTS + part, n);
        g = __builtin_amdgcn_readfirstlane(g);
        const size_t base = (size_t)q_base + (size_t)part * as_k(a.scales)[scale_slot].q_cap;
        for (uint32_t i = lane; i < n; i += 64u) a.q_out[base + g + i] = q[i];
        // where this unit's survivors sit: the band-major queue pass finds them by unit
        if (run_idx != 0xffffffffu && lane == 0) *reinterpret_cast<uint2*>(a.run_table + 2u * (size_t)run_idx) = make_uint2(g, n);
    }
}

// Stage-tree cascades (e.g. frontalface_alt_tree: stage 4 has two child chains): a
// window's next stage depends on whether it passed (on_pass) or failed (on_fail), so
// every queued window carries its target stage.  Stages are visited once, in a
// topological order of the pass/fail graph computed on the host (StageDev::order —
// e.g. 0..4, chain 5,7,..,39, then chain 6,8,..,46: failing in the first chain jumps
// BACK to stage 6); at stage s only the lanes whose target is s evaluate, the others
// ride along.  Whole cascade in one pass.
// `emit(mask, mine, off)`: the lanes of `mask` fell off the tree's end (accepted); `mine` says whether this lane is one.
template <bool TREES, bool COUNT, typename Emit>
__device__ __forceinline__ void run_stages_general_to(const CascadeArgs& a, rsrc_t img, QEntry* q, int32_t* tgt,
                                                      uint32_t n, uint32_t table_first, uint32_t lane, Emit emit) {
    kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + table_first;
    kptr<StageDev> stages = as_k(a.stages);
    for (uint32_t i = lane; i < n; i += 64u) tgt[i] = (int32_t)stages[a.stage_begin].order;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t oi = a.stage_begin; oi < a.stage_end && n != 0u; ++oi) {
        const uint32_t s = stages[oi].order;
        const uint32_t first_node = stages[s].first_node;
        const uint32_t n_nodes = stages[s].n_nodes;
        const float threshold = stages[s].threshold;
        const int32_t on_pass = stages[s].on_pass, on_fail = stages[s].on_fail;
        kptr<NodeRecDev> tab = table + first_node;
        uint32_t m = 0, entered = 0;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + lane;
            const bool act = i < n;
            const QEntry e = q[act ? i : 0u];
            int32_t t = tgt[act ? i : 0u];
            const bool here = act && t == (int32_t)s;
            uint32_t cntg[2] = {0u, 0u};
            if (here) t = (stage_sum_of<TREES, COUNT>(GlobalImg{img}, tab, n_nodes, e.off, e.var, true, cntg) >= threshold) ? on_pass : on_fail;
            if (TREES && COUNT) tree_count_flush(a, cntg[0], cntg[1], lane);
            const bool keep = act && t >= 0;
            const unsigned long long acc_mask = __ballot(act && t == -1);  // accepted: falls off the tree's end
            if (acc_mask != 0ull) emit(acc_mask, act && t == -1, e.off);
            if (COUNT) entered += (uint32_t)__popcll(__ballot(here));
            const unsigned long long mask = __ballot(keep);
            __builtin_amdgcn_wave_barrier();
            if (keep) {
                const uint32_t pos = m + mbcnt(mask);
                q[pos] = e;
                tgt[pos] = t;
            }
            m += (uint32_t)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
        }
        if (COUNT && lane == 0 && entered != 0u) atomicAdd(a.stage_entered + s, (unsigned long long)entered);
        n = m;
    }
}

template <bool TREES, bool COUNT>
__device__ __forceinline__ void run_stages_general(const CascadeArgs& a, rsrc_t img, QEntry* q, int32_t* tgt,
                                                   uint32_t n, uint32_t scale_slot, uint32_t table_first,
                                                   uint32_t lane) {
    run_stages_general_to<TREES, COUNT>(a, img, q, tgt, n, table_first, lane, [&](unsigned long long acc_mask, bool mine, uint32_t off) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(a.det_count, (uint32_t)__popcll(acc_mask));
        g = __builtin_amdgcn_readfirstlane(g);
        const uint32_t pos = g + mbcnt(acc_mask);
        if (mine && pos < a.det_cap) a.det[pos] = DetEntry{off, scale_slot};
    });
}

template <bool FROM_GRID, bool TREES, bool LAST, bool COUNT, bool GENERAL, int NW = 1>
__global__ __launch_bounds__(GATHER_WAVES_MAX * 64) void cascade_pass(CascadeArgs a) {
    // (stage trees are launched with WAVES_PER_BLOCK waves: their second array would not fit next to the tiles otherwise)
    __shared__ QEntry lds_q[(GENERAL ? WAVES_PER_BLOCK : GATHER_WAVES_MAX) * UNIT_WINDOWS];
    __shared__ int32_t lds_tgt[GENERAL ? WAVES_PER_BLOCK * UNIT_WINDOWS : 1];
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wpb = blockDim.x >> 6;     // the host's choice (CascadeArgs::gather_waves, clamped by the launcher)
    QEntry* q = lds_q + wib * UNIT_WINDOWS;
    const uint32_t rank = blockIdx.x * wpb + wib;
    kptr<ScaleDev> scales = as_k(a.scales);
    // the whole batch of sum images behind one descriptor (host keeps it below 4 GiB)
    const rsrc_t img = make_rsrc(a.sum, a.sum_bytes);

    if (FROM_GRID) {
        kptr<UnitDev> units = as_k(a.units);
        const uint32_t total_units = a.n_units * a.n_frames;
        const uint32_t frame_bytes4 = a.frame_elems * 4u;
        // Blocks are dealt round-robin over the 8 XCDs (observed placement; used for speed only): the waves that
        // share an XCD — and its 4 MiB L2 — take one contiguous eighth of the (frame, unit) list, i.e. they work on
        // the same frame's sum image at the same time instead of on every frame in flight.
        uint32_t u_begin = 0, u_end = total_units, u_step = a.total_waves, u_first = rank;
        if (a.xcd_affinity != 0u && gridDim.x >= 8u) {
            const uint32_t xcd = blockIdx.x & 7u;
            u_begin = (uint32_t)((unsigned long long)total_units * xcd / 8u);
            u_end = (uint32_t)((unsigned long long)total_units * (xcd + 1u) / 8u);
            u_step = ((gridDim.x - xcd + 7u) >> 3) * wpb;
            u_first = u_begin + (blockIdx.x >> 3) * wpb + wib;
        }
        for (uint32_t u = u_first; u < u_end; u += u_step) {
            const uint32_t frame = u / a.n_units;
            const uint32_t r = u - frame * a.n_units;
            const uint32_t slot = units[r].scale;
            const uint32_t first = units[r].first;
            const uint32_t count = units[r].count;
            const float step = scales[slot].step;
            const uint32_t pos_base = scales[slot].pos_base;
            const uint32_t nx = scales[slot].nx;
            const uint32_t e_lt = scales[slot].e_lt, e_dw = scales[slot].e_dw, e_dh = scales[slot].e_dh;
            const float area = scales[slot].area;
            const size_t frame_off = (size_t)frame * a.frame_elems;
            const rsrc_t sum_f = make_rsrc(a.sum + frame_off, frame_bytes4);
            const rsrc_t sq_f = make_rsrc(a.sqsum + frame_off, frame_bytes4 * 2u);
            const uint32_t frame_bytes = frame * frame_bytes4;  // < 2^32, checked on the host
            // precomputeWindows (clod.cpp:495-527): x = lrint(ix * step); a unit is either a row-major run of
            // windows or a 2-D block of them (compact footprint in the sum image: the L2 traffic of the
            // large scales is what bounds this pass)
            const uint32_t bw = units[r].bw, ny = scales[slot].ny;
            uint32_t n_q = 0;
            for (uint32_t i0 = 0; i0 < count; i0 += 64u) {
                const uint32_t i = i0 + lane;
                uint32_t ix, iy;
                if (bw != 0u) {
                    const uint32_t ty = i / bw;
                    ix = (first & 0xffffu) + (i - ty * bw);
                    iy = (first >> 16) + ty;
                } else {
                    iy = (first + i) / nx;
                    ix = first + i - iy * nx;
                }
                bool valid = i < count && ix < nx && iy < ny;
                if (a.skip_bits != nullptr && valid)   // uniform test
                    valid = window_visited(a, frame, scales[slot].skip_base, scales[slot].skip_wpr, nx, ix, iy);
                QEntry en{0u, 0.0f};
                if (valid) {
                    const uint32_t x = window_pos(a, pos_base, ix, step);
                    const uint32_t y = window_pos(a, pos_base, iy, step);
                    const uint32_t e = y * a.stride + x;
                    en.var = window_variance(sum_f, sq_f, e, e_lt, e_dw, e_dh, area, a.signed_mean != 0u);
                    en.off = frame_bytes + e * 4u;
                }
                const unsigned long long mask = __ballot(valid);
                if (valid) q[n_q + mbcnt(mask)] = en;
                n_q += (uint32_t)__popcll(mask);
            }
            __builtin_amdgcn_wave_barrier();
            if (GENERAL)
                run_stages_general<TREES, COUNT>(a, img, q, lds_tgt + wib * UNIT_WINDOWS, n_q, slot,
                                                 scales[slot].table_first, lane);
            else
                run_stages_linear<TREES, LAST, COUNT>(a, img, q, n_q, slot, scales[slot].table_first,
                                                      scales[slot].q_base, lane, a.stage_begin, frame_part(a, frame),
                                                      a.run_table != nullptr ? u : 0xffffffffu);
            __builtin_amdgcn_wave_barrier();
        }
    } else if (!GENERAL && a.q_groups != nullptr) {
        // Band-major queue pass (see CascadeArgs::run_table).  The (frame, group) list is cut by frame group — the parts of the
        // sub-queues — with one ticket counter each; an XCD's waves start on "their" part and steal from the others.
        kptr<UnitDev> groups = as_k(a.q_groups);
        kptr<uint32_t> runs = as_k(a.run_table);
        const uint32_t nG = a.n_q_groups;
        uint32_t part = a.xcd_affinity != 0u ? (blockIdx.x & (Q_PARTS - 1u)) : 0u;
        for (uint32_t tries = 0; tries < Q_PARTS;) {
            // frames of this part: frame_part(f) == part  <=>  f in [ceil(part * n / 8), ceil((part + 1) * n / 8))
            // (fewer frames than parts: part = frame)
            const uint32_t f_lo = a.n_frames >= Q_PARTS ? (uint32_t)(((unsigned long long)part * a.n_frames + Q_PARTS - 1u) / Q_PARTS) : min(part, a.n_frames);
            const uint32_t f_hi = a.n_frames >= Q_PARTS ? (uint32_t)(((unsigned long long)(part + 1u) * a.n_frames + Q_PARTS - 1u) / Q_PARTS)
                                                        : min(part + 1u, a.n_frames);
            const uint32_t n_items = (f_hi - f_lo) * nG;
            uint32_t t = n_items;
            if (n_items != 0u) {
                if (lane == 0) t = atomicAdd(a.q_ticket + part, 1u);
                t = __builtin_amdgcn_readfirstlane(t);
            }
            if (t >= n_items) {   // this part is used up: steal from the next one
                part = (part + 1u) & (Q_PARTS - 1u);
                ++tries;
                continue;
            }
            const uint32_t frame = f_lo + t / nG, gi = t - (t / nG) * nG;
            const uint32_t slot = groups[gi].scale, u0 = groups[gi].first, nu = groups[gi].count;
            const uint32_t q_base = scales[slot].q_base;
            const size_t base = (size_t)q_base + (size_t)part * scales[slot].q_cap;
            uint32_t n_q = 0;
            for (uint32_t k = 0; k < nu; ++k) {
                const size_t ri = ((size_t)frame * a.n_units + u0 + k) * 2u;
                const uint32_t g = runs[ri], n = runs[ri + 1u];
                if (n == 0u) continue;
                if (n_q + n > (uint32_t)UNIT_WINDOWS) {   // the wave's queue is full: sweep what it holds first
                    __builtin_amdgcn_wave_barrier();
                    run_stages_linear<TREES, LAST, COUNT, NW>(a, img, q, n_q, slot, scales[slot].table_first, q_base, lane, a.stage_begin, part);
                    __builtin_amdgcn_wave_barrier();
                    n_q = 0;
                }
                for (uint32_t i = lane; i < n; i += 64u) q[n_q + i] = a.q_in[base + g + i];
                n_q += n;
            }
            __builtin_amdgcn_wave_barrier();
            if (n_q != 0u) run_stages_linear<TREES, LAST, COUNT, NW>(a, img, q, n_q, slot, scales[slot].table_first, q_base, lane, a.stage_begin, part);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // Queue pass.  Every scale's segment is cut into Q_PARTS parts by frame group (frame_part); the waves that
        // share an XCD work through "their" part first — the frames whose sum images their L2 already holds from
        // the previous pass — and then steal chunks from the other parts.  Chunks are handed out by one ticket
        // counter per part; ticket t of part x is the t-th chunk of that part's scales taken in order.
        kptr<uint32_t> counts = as_k(a.q_in_count);
        // chunk size: spread this pass's windows over all waves (a wave works through its chunk serially, ~1 us
        // per stump, so late passes with few windows want small chunks), in whole 64-lane groups, at most the
        // LDS queue capacity
        // (the counters are summed lane-parallel: a scalar walk over up to 512 of them is ~0.1 ms of latency, which
        // was the floor of every queue pass launch)
        constexpr uint32_t N_CNT = (MAX_SCALES * Q_PARTS + 63u) / 64u;
        uint32_t my_cnt[N_CNT];
        uint32_t total = 0;
#pragma unroll
        for (uint32_t k = 0; k < N_CNT; ++k) {
            const uint32_t i = k * 64u + lane;
            my_cnt[k] = i < a.n_scales * Q_PARTS ? a.q_in_count[i] : 0u;
            total += my_cnt[k];
        }
        auto wave_total = [](uint32_t v) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            return (uint32_t)__builtin_amdgcn_readfirstlane(v);
        };
        total = wave_total(total);
        // With fewer windows than 64 per wave the chunks shrink further (a.min_chunk, 32 by default): the thin waves of a late
        // pass are bound by windows x blocks memory round trips each, and a cluster of detections — neighbours in the queue —
        // then spreads over two waves instead of sitting in one (16 and 8 make the populous passes of a stage tree slower).
        const uint32_t min_chunk = max(1u, min(a.min_chunk, 64u));
        const uint32_t per_wave = total / a.total_waves;
        const uint32_t chunk = per_wave >= 64u ? min((uint32_t)UNIT_WINDOWS, ((per_wave + 63u) / 64u) * 64u)
                                               : max(min_chunk, ((per_wave + min_chunk - 1u) / min_chunk) * min_chunk);
        if (a.thin_pass_spread != 0u) {
            // Fewer chunks than waves (a late pass, a single frame): whichever waves draw the tickets first get the
            // chunks, and with 24 resident waves per CU some CUs end up with twice the average — the pass then waits for
            // the CU with the most (measured, 4096 x 4096 stage tree: 6.5 ms with 8 workgroups per CU against 3.9 ms
            // with 4, same chunks).  Workgroups are dealt round-robin over XCDs and CUs
            // (tools/microbench/dispatch_order.hip: the first 552 of 2048 sit 2-3 per CU on all 256 CUs), so only the
            // first ceil(chunks / waves per workgroup) workgroups draw tickets; the others leave.  Speed only: the
            // remaining waves loop until every part is drained.
            uint32_t n_chunks = 0;
#pragma unroll
            for (uint32_t k = 0; k < N_CNT; ++k) n_chunks += (my_cnt[k] + chunk - 1u) / chunk;
            n_chunks = wave_total(n_chunks);
            if (blockIdx.x >= max((n_chunks + wpb - 1u) / wpb, 1u)) return;
        }
        // chunks per part (counter i belongs to part i % Q_PARTS, and 64 % Q_PARTS == 0: lane l sums part l % Q_PARTS):
        // an empty or used-up part is left without walking its scales
        uint32_t part_chunks_v = 0;
#pragma unroll
        for (uint32_t k = 0; k < N_CNT; ++k) part_chunks_v += (my_cnt[k] + chunk - 1u) / chunk;
#pragma unroll
        for (int d = 32; d >= (int)Q_PARTS; d >>= 1) part_chunks_v += __shfl_xor(part_chunks_v, d, 64);
        uint32_t part = a.xcd_affinity != 0u ? (blockIdx.x & (Q_PARTS - 1u)) : 0u;
        for (uint32_t tries = 0; tries < Q_PARTS;) {
            const uint32_t part_chunks = __builtin_amdgcn_readfirstlane((uint32_t)__shfl(part_chunks_v, (int)part, 64));   // (uniform: keep the walk scalar)
            uint32_t t = part_chunks;
            if (part_chunks != 0u) {
                if (lane == 0) t = atomicAdd(a.q_ticket + part, 1u);
                t = __builtin_amdgcn_readfirstlane(t);
            }
            // ticket -> (scale, chunk): walk the part's scales.  The entries of a (scale, part) sub-queue arrive roughly
            // frame by frame (the grid pass walks an XCD's eighth of the (frame, unit) list in order), so the walk goes
            // through the part in q_slices slices of every scale's chunk range — slice j of ALL scales before slice j + 1
            // of any: the waves of an XCD then work on about one frame's sum image at a time (8 MB) instead of sweeping
            // the part's whole frame group (66 MB for 64 x 1080p) once per scale.  q_slices = 1: scale by scale.
            uint32_t slot = a.n_scales, c = t;
            if (t < part_chunks && a.q_slices <= 1u) {
                // (the plain walk: no divisions — a thin pass draws thousands of tickets per wave, and a scalar integer
                // division is a long instruction sequence: with the sliced walk for J = 1 the chains of a stage tree took
                // 2.9 ms instead of 0.43)
                for (slot = 0; slot < a.n_scales; ++slot) {
                    const uint32_t n_chunks = (counts[slot * Q_PARTS + part] + chunk - 1u) / chunk;
                    if (c < n_chunks) break;
                    c -= n_chunks;
                }
            } else if (t < part_chunks) {
                const uint32_t J = a.q_slices;
                bool found = false;
                for (uint32_t j = 0; j < J && !found; ++j)
                    for (slot = 0; slot < a.n_scales; ++slot) {
                        const uint32_t n_chunks = (counts[slot * Q_PARTS + part] + chunk - 1u) / chunk;
                        const uint32_t lo = (j * n_chunks + J - 1u) / J, hi = ((j + 1u) * n_chunks + J - 1u) / J;
                        if (c < hi - lo) {
                            c += lo;
                            found = true;
                            break;
                        }
                        c -= hi - lo;
                    }
                if (!found) slot = a.n_scales;
            }
            if (slot == a.n_scales) {   // this part is used up: steal from the next one
                part = (part + 1u) & (Q_PARTS - 1u);
                ++tries;
                continue;
            }
            const uint32_t cnt = counts[slot * Q_PARTS + part];
            const uint32_t q_base = scales[slot].q_base;
            const size_t base = (size_t)q_base + (size_t)part * scales[slot].q_cap;
            const uint32_t c0 = c * chunk;
            const uint32_t n = min(cnt - c0, chunk);
            for (uint32_t i = lane; i < n; i += 64u) q[i] = a.q_in[base + c0 + i];
            __builtin_amdgcn_wave_barrier();
            const unsigned long long t_chunk = VJ_STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
            if (GENERAL)   // the rest of a stage tree, for the survivors of its linear prefix
                run_stages_general<TREES, COUNT>(a, img, q, lds_tgt + wib * UNIT_WINDOWS, n, slot, scales[slot].table_first, lane);
            else
                run_stages_linear<TREES, LAST, COUNT, NW>(a, img, q, n, slot, scales[slot].table_first, q_base, lane, a.stage_begin, part);
            __builtin_amdgcn_wave_barrier();
            if (VJ_STAMPS && lane == 0) {   // diagnostic build: time per chunk of a queue pass (sum, max, chunks)
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_chunk;
                atomicAdd(a.stage_entered + 48, dt);
                atomicMax(a.stage_entered + 49, dt);
                atomicAdd(a.stage_entered + 50, 1ull);
            }
        }
    }
}


// ------------------------------------------------------------------ P2: the windows the CPU variants visit
// The reference's CPU loops are sequential: after a window that stage 0 rejects, the next one is not evaluated at
// all (x_incr / subwindow_incr = 2: clod.cpp:1430 inside a row, :729-732 over the flattened list).  That is the
// recurrence e[i] = !(e[i-1] && f[i-1]) with f = "stage 0 rejects"; a window that follows a non-reject is always
// visited, so e[i] is the PARITY of the run of rejects that ends at i-1 — local information once f is known for every
// grid window.  skip_fail_bits computes f (one 64-bit word per 64 consecutive windows), skip_resolve turns the words of
// a recurrence domain (a row, or a scale's whole list) into visited bits; the cascade passes then drop unvisited
// windows when they enumerate the grid.
template <bool TREES>
__global__ __launch_bounds__(256) void skip_fail_bits(CascadeArgs a) {
    const uint32_t lane = lane_id();
    const uint32_t rank = blockIdx.x * 4u + (threadIdx.x >> 6);
    kptr<ScaleDev> scales = as_k(a.scales);
    kptr<StageDev> stages = as_k(a.stages);
    kptr<UnitDev> units = as_k(a.skip_units);
    const uint32_t total = a.n_skip_units * a.n_frames;
    const uint32_t frame_bytes4 = a.frame_elems * 4u;
    const rsrc_t img = make_rsrc(a.sum, a.sum_bytes);
    for (uint32_t u = rank; u < total; u += gridDim.x * 4u) {
        const uint32_t frame = u / a.n_skip_units;
        const uint32_t r = u - frame * a.n_skip_units;
        const uint32_t slot = units[r].scale, first = units[r].first, count = units[r].count, word = units[r].bw;
        const uint32_t nx = scales[slot].nx;
        const float step = scales[slot].step;
        const uint32_t pos_base = scales[slot].pos_base;
        uint32_t ix, iy;
        if (scales[slot].skip_wpr != 0u) {
            ix = (first & 0xffffu) + lane;
            iy = first >> 16;
        } else {
            const uint32_t i = first + lane;
            iy = i / nx;
            ix = i - iy * nx;
        }
        bool fail = false;
        if (lane < count) {
            const size_t frame_off = (size_t)frame * a.frame_elems;
            const rsrc_t sum_f = make_rsrc(a.sum + frame_off, frame_bytes4);
            const rsrc_t sq_f = make_rsrc(a.sqsum + frame_off, frame_bytes4 * 2u);
            const uint32_t x = window_pos(a, pos_base, ix, step), y = window_pos(a, pos_base, iy, step);
            const uint32_t e = y * a.stride + x;
            const float var = window_variance(sum_f, sq_f, e, scales[slot].e_lt, scales[slot].e_dw, scales[slot].e_dh, scales[slot].area,
                                              a.signed_mean != 0u);
            kptr<NodeRecDev> tab = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + scales[slot].table_first + stages[0].first_node;
            fail = !(stage_sum_of<TREES>(GlobalImg{img}, tab, stages[0].n_nodes, frame * frame_bytes4 + e * 4u, var) >= stages[0].threshold);
        }
        const unsigned long long F = __ballot(fail);
        if (lane == 0) a.skip_bits[(size_t)frame * a.skip_frame_words + word] = F;
    }
}

__global__ __launch_bounds__(256) void skip_resolve(CascadeArgs a) {
    const uint32_t lane = lane_id();
    const uint32_t rank = blockIdx.x * 4u + (threadIdx.x >> 6);
    kptr<UnitDev> segs = as_k(a.skip_segs);
    const uint32_t total = a.n_skip_segs * a.n_frames;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t u = rank; u < total; u += gridDim.x * 4u) {
        const uint32_t frame = u / a.n_skip_segs;
        const uint32_t r = u - frame * a.n_skip_segs;
        unsigned long long* words = a.skip_bits + (size_t)frame * a.skip_frame_words + segs[r].first;
        const uint32_t n_words = segs[r].count;
        uint32_t carry = 0;   // parity of the reject run that ends just before the chunk's first bit (the first window is visited)
        for (uint32_t w0 = 0; w0 < n_words; w0 += 64u) {
            const bool act = w0 + lane < n_words;
            const unsigned long long F = act ? words[w0 + lane] : 0ull;   // bits past the domain's end are 0 (not rejects)
            // a word of 64 rejects passes the parity through; any other word ends with a run whose parity is its own
            const bool through = F == ~0ull;
            const uint32_t tpar = (uint32_t)__clzll((long long)~F) & 1u;   // leading ones of F = the run that ends at bit 63
            const unsigned long long m = __ballot(act && !through);
            const unsigned long long lower = m & below;
            const uint32_t src = lower != 0ull ? 63u - (uint32_t)__clzll((long long)lower) : 0u;
            const uint32_t from_lane = (uint32_t)__shfl((int)tpar, (int)src, 64);
            uint32_t p = lower != 0ull ? from_lane : carry;
            unsigned long long V = 0ull;
#pragma unroll 8
            for (uint32_t b = 0; b < 64u; ++b) {
                V |= (unsigned long long)(p ^ 1u) << b;             // visited when the run before it is even
                p = ((F >> b) & 1ull) != 0ull ? p ^ 1u : 0u;        // run-length parity after this window
            }
            if (act) words[w0 + lane] = V;
            if (m != 0ull) carry = (uint32_t)__shfl((int)tpar, 63 - __clzll((long long)m), 64);
        }
    }
}

int launch_skip_resolve(const CascadeArgs& a, int n_blocks, void* stream_) {
    hipLaunchKernelGGL(skip_resolve, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

int launch_skip_bitmap(const CascadeArgs& a, bool trees, int n_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (trees) hipLaunchKernelGGL(skip_fail_bits<true>, dim3(n_blocks), dim3(256), 0, stream, a);
    else       hipLaunchKernelGGL(skip_fail_bits<false>, dim3(n_blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(skip_resolve, dim3(n_blocks), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------ regions of interest on the device
// Two cascades back to back (BASELINE config 5): the raw candidates of a first cascade become regions
// of interest without leaving the device, and a second cascade runs inside every region on the frame's OWN integral
// images.  A rectangle sum is a four-corner difference, so it does not depend on where the integral image starts: a
// region evaluated in place gives exactly what the reference would compute on the sub-image (its caller would pass
// clodDetectObjects a sub-image header).  Every region lays out its own grid — setupScale (clod.cpp:371-415) for the
// region's size, evaluated on the device with the same f32 operations as the host's plan_scales — and is cut into
// units of <= UNIT_WINDOWS windows; persistent waves then draw units from a ticket counter and run the whole second
// cascade on each with the compacting stage sweep of the global-gather passes.  No size-dependent host plan, no host
// round trip between the cascades.

__global__ __launch_bounds__(256) void dets_to_rois(RoiArgs r) {
    const uint32_t n = min(*r.det_in_count, min(r.det_in_cap, r.max_rois));
    kptr<ScaleDev> scales = as_k(r.scales_in);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const DetEntry d = r.det_in[i];
        const uint32_t frame = d.off / r.frame_bytes;
        const uint32_t el = (d.off - frame * r.frame_bytes) >> 2;
        const uint32_t y = el / r.stride, x = el - y * r.stride;
        r.rois[i] = RoiDev{(int32_t)frame, (int32_t)x, (int32_t)y, (int32_t)scales[d.scale].win_w, (int32_t)scales[d.scale].win_h};
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *r.n_rois = n;
}

__global__ __launch_bounds__(256) void roi_plan_units(RoiArgs r, CascadeArgs a) {
    const uint32_t n = min(*r.n_rois, r.max_rois);
    kptr<ScaleDev> scales = as_k(a.scales);
    const uint32_t total = n * a.n_scales;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
        const uint32_t roi = t / a.n_scales, slot = t - roi * a.n_scales;
        const RoiDev R = r.rois[roi];
        if (R.frame < 0 || (uint32_t)R.frame >= r.n_frames || R.x < 0 || R.y < 0 || R.w <= 0 || R.h <= 0 || R.x + R.w > r.frame_w ||
            R.y + R.h > r.frame_h) {
            if (slot == 0u) atomicAdd(r.n_units + 1, 1u);   // invalid regions are counted and reported by the host
            continue;
        }
        const float cs = scales[slot].scale_f, step = scales[slot].step;
        const int32_t win_w = (int32_t)scales[slot].win_w, win_h = (int32_t)scales[slot].win_h;
        // scale enumeration (clod.cpp:1198-1204): the f32 chain s_k increases, so scale k is enumerated for this region
        // exactly when its own test holds
        if (!(cs * (float)r.win_w0 < (float)(R.w - 10) && cs * (float)r.win_h0 < (float)(R.h - 10))) continue;
        // setupScale's rejections (clod.cpp:391-401); min / max limits were applied when the plan chose its scales
        if (win_w > R.w || win_h > R.h) continue;
        const int32_t nx = __float2int_rn((float)(R.w - win_w) / step);   // lrint of an int / float quotient (clod.cpp:409-412)
        const int32_t ny = __float2int_rn((float)(R.h - win_h) / step);
        if (nx <= 0 || ny <= 0) continue;
        const uint32_t nwin = (uint32_t)nx * (uint32_t)ny;
        if (r.tiles != nullptr && scales[slot].tile_rw != 0u && scales[slot].tile_class == 0u && nwin >= r.tile_min_windows && nx < 65536 &&
            ny < 65536) {
            // a grid worth staging: tiles of the scale's tile shape inside the region (cascade_tile_roi_pass)
            const uint32_t tw_s = scales[slot].tile_tw, th_s = scales[slot].tile_th;
            const uint32_t tx = ((uint32_t)nx + tw_s - 1u) / tw_s, ty = ((uint32_t)ny + th_s - 1u) / th_s;
            // equal parts: the tiles of a region have the same shape (and the last one is not a sliver)
            const uint32_t tw = ((uint32_t)nx + tx - 1u) / tx, th = ((uint32_t)ny + ty - 1u) / ty;
            const uint32_t base = atomicAdd(r.n_tiles, tx * ty);
            for (uint32_t j = 0; j < tx * ty && base + j < r.max_tiles; ++j)
                r.tiles[base + j] = RoiTile{roi, slot, (j % tx) * tw | ((j / tx) * th) << 16, (uint32_t)nx | (uint32_t)ny << 16, tw | th << 16};
            continue;
        }
        const uint32_t nun = (nwin + UNIT_WINDOWS - 1u) / UNIT_WINDOWS;
        const uint32_t base = atomicAdd(r.n_units, nun);
        for (uint32_t j = 0; j < nun && base + j < r.max_units; ++j)
            r.units[base + j] = RoiUnit{roi, slot, j * UNIT_WINDOWS, min((uint32_t)UNIT_WINDOWS, nwin - j * UNIT_WINDOWS), (uint32_t)nx, (uint32_t)ny};
    }
}

template <bool TREES, bool COUNT, bool GENERAL>
__global__ __launch_bounds__(GATHER_WAVES_MAX * 64) void cascade_roi_pass(RoiArgs r, CascadeArgs a) {
    __shared__ QEntry lds_q[(GENERAL ? WAVES_PER_BLOCK : GATHER_WAVES_MAX) * UNIT_WINDOWS];
    __shared__ int32_t lds_tgt[GENERAL ? WAVES_PER_BLOCK * UNIT_WINDOWS : 1];   // stage trees: the stage every queued window visits next
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    QEntry* q = lds_q + wib * UNIT_WINDOWS;
    kptr<ScaleDev> scales = as_k(a.scales);
    const rsrc_t img = make_rsrc(a.sum, a.sum_bytes);
    const uint32_t n_units = min(*r.n_units, r.max_units);
    const uint32_t frame_bytes4 = a.frame_elems * 4u;
    while (true) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(r.ticket, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= n_units) break;
        const RoiUnit un = r.units[t];
        const uint32_t roi = __builtin_amdgcn_readfirstlane(un.roi), slot = __builtin_amdgcn_readfirstlane(un.slot);
        const uint32_t first = __builtin_amdgcn_readfirstlane(un.first), count = __builtin_amdgcn_readfirstlane(un.count);
        const uint32_t nx = __builtin_amdgcn_readfirstlane(un.nx);
        const RoiDev R = r.rois[roi];
        const uint32_t frame = __builtin_amdgcn_readfirstlane((uint32_t)R.frame);
        const uint32_t x0 = __builtin_amdgcn_readfirstlane((uint32_t)R.x), y0 = __builtin_amdgcn_readfirstlane((uint32_t)R.y);
        const float step = scales[slot].step;
        const uint32_t e_lt = scales[slot].e_lt, e_dw = scales[slot].e_dw, e_dh = scales[slot].e_dh;
        const float area = scales[slot].area;
        const size_t frame_off = (size_t)frame * a.frame_elems;
        const rsrc_t sum_f = make_rsrc(a.sum + frame_off, frame_bytes4);
        const rsrc_t sq_f = make_rsrc(a.sqsum + frame_off, frame_bytes4 * 2u);
        const uint32_t frame_bytes = frame * frame_bytes4;
        uint32_t n_q = 0;
        for (uint32_t i0 = 0; i0 < count; i0 += 64u) {
            const bool valid = i0 + lane < count;
            const uint32_t i = first + i0 + lane;
            const uint32_t iy = i / nx, ix = i - iy * nx;
            QEntry en{0u, 0.0f};
            if (valid) {
                // precomputeWindows (clod.cpp:495-527) inside the region: x = lrint(ix * step) from the region's origin
                const uint32_t x = x0 + (uint32_t)__float2int_rn((float)ix * step);
                const uint32_t y = y0 + (uint32_t)__float2int_rn((float)iy * step);
                const uint32_t e = y * a.stride + x;
                en.var = window_variance(sum_f, sq_f, e, e_lt, e_dw, e_dh, area, a.signed_mean != 0u);
                en.off = frame_bytes + e * 4u;
            }
            const unsigned long long mask = __ballot(valid);
            if (valid) q[n_q + mbcnt(mask)] = en;
            n_q += (uint32_t)__popcll(mask);
        }
        __builtin_amdgcn_wave_barrier();
        if (GENERAL) {
            // a stage tree as the second cascade (tempcv.cpp:834-861): the per-window walk of run_stages_general, accepted
            // windows going to the region pass's own detection list
            run_stages_general_to<TREES, COUNT>(a, img, q, lds_tgt + wib * UNIT_WINDOWS, n_q, scales[slot].table_first, lane,
                                                [&](unsigned long long acc_mask, bool mine, uint32_t off) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(r.det_count, (uint32_t)__popcll(acc_mask));
                g = __builtin_amdgcn_readfirstlane(g);
                const uint32_t pos = g + mbcnt(acc_mask);
                if (mine && pos < r.det_cap) r.det[pos] = RoiDet{off, slot, roi};
            });
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + scales[slot].table_first;
        const uint32_t n = sweep_stages<TREES, COUNT>(a, GlobalImg{img}, table, q, n_q, lane, a.stage_begin, a.stage_end);
        if (n != 0u) {
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(r.det_count, n);
            g = __builtin_amdgcn_readfirstlane(g);
            for (uint32_t i = lane; i < n; i += 64u)
                if (g + i < r.det_cap) r.det[g + i] = RoiDet{q[i].off, slot, roi};
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool COUNT>
__global__ void cascade_tile_roi_pass(RoiArgs r, CascadeArgs a);   // (defined with the tile kernel below)

template <bool GENERAL>
static void launch_roi_pass(const RoiArgs& r, const CascadeArgs& a, bool trees, bool count, dim3 g, dim3 b, hipStream_t stream) {
    if (trees) {
        if (count) hipLaunchKernelGGL((cascade_roi_pass<true, true, GENERAL>), g, b, 0, stream, r, a);
        else       hipLaunchKernelGGL((cascade_roi_pass<true, false, GENERAL>), g, b, 0, stream, r, a);
    } else {
        if (count) hipLaunchKernelGGL((cascade_roi_pass<false, true, GENERAL>), g, b, 0, stream, r, a);
        else       hipLaunchKernelGGL((cascade_roi_pass<false, false, GENERAL>), g, b, 0, stream, r, a);
    }
}

// Threads per workgroup of the global-gather kernels: the host's choice (CascadeArgs::gather_waves), within what the kernels' LDS
// arrays hold; the kernels read the same field.
static uint32_t gather_block_threads(const CascadeArgs& a, bool general) {
    const uint32_t most = general ? (uint32_t)WAVES_PER_BLOCK : (uint32_t)GATHER_WAVES_MAX;
    return 64u * std::min(std::max(a.gather_waves, 1u), most);
}

int launch_roi_chain(const RoiArgs& r, const CascadeArgs& a, bool from_dets, bool trees, bool count, bool general, int n_blocks,
                     void* stream_, void* stream2_, void* fork_ev_, void* join_ev_) {
    hipStream_t stream = (hipStream_t)stream_, stream2 = (hipStream_t)stream2_;
    if (from_dets) hipLaunchKernelGGL(dets_to_rois, dim3(256), dim3(256), 0, stream, r);
    hipLaunchKernelGGL(roi_plan_units, dim3(512), dim3(256), 0, stream, r, a);
    dim3 g(n_blocks), b(gather_block_threads(a, general));
    // the regions' tiles (LDS / VALU-bound) and the thin units of the region pass (texture-address-bound) side by side on two
    // streams when the caller lends a second one: the gather pass first, the tile workgroups fill the CUs next to it
    const bool two = r.tiles != nullptr && stream2 != nullptr;
    hipStream_t sB = two ? stream2 : stream;
    if (two) {
        if (hipEventRecord((hipEvent_t)fork_ev_, stream) != hipSuccess || hipStreamWaitEvent(stream2, (hipEvent_t)fork_ev_, 0) != hipSuccess)
            return (int)hipGetLastError();
    }
    if (general) launch_roi_pass<true>(r, a, trees, count, g, b, sB);
    else launch_roi_pass<false>(r, a, trees, count, g, b, sB);
    if (r.tiles != nullptr) {
        dim3 tg(r.tile_blocks), tb(TILE_WAVES * 64);
        if (count) hipLaunchKernelGGL((cascade_tile_roi_pass<true>), tg, tb, a.tile_lds_bytes, stream, r, a);
        else       hipLaunchKernelGGL((cascade_tile_roi_pass<false>), tg, tb, a.tile_lds_bytes, stream, r, a);
    }
    if (two) {
        if (hipEventRecord((hipEvent_t)join_ev_, stream2) != hipSuccess || hipStreamWaitEvent(stream, (hipEvent_t)join_ev_, 0) != hipSuccess)
            return (int)hipGetLastError();
    }
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------ LDS-tile pass
// The whole cascade for the dense small scales (cascade_tile_pass, below).  A workgroup of TILE_WAVES waves owns
// a tile of up to TILE_WAVES * TILE_WAVE_CAP windows: it stages the tile's footprint of the sum image in LDS, every
// wave runs the compacting stage sweep over its share of the windows with the rectangle corners gathered from LDS
// instead of through the texture-address unit, the waves pool their survivors before the later stages, and once
// few windows are left the tile finishes the cascade with the two routines that follow.  What survives is a
// detection; only a tile that stays crowded hands its windows to the global queues, as {global byte offset,
// variance}.  The same kernel also runs unstaged (L2 gathers) on 2-D blocks of windows of the large scales.

// Stump-parallel finish of a tile (stump cascades).  lds_q[0..T) holds the tile's T <= TILE_SP_MAX_WINDOWS
// surviving windows.  Per stage, in blocks of <= 64 consecutive stumps: the block's node records are copied
// to LDS field-major (four blocks are prefetched into registers meanwhile); lane j owns stump j of the
// block, wave w takes windows w, w + 8, ... two at a time; a (window, block) result is 64 verdict bits — which
// stumps answered alpha[1], one __ballot — and the block's leaf sum (a DPP butterfly).  After the stage's last
// block, thread t adds window t's block sums: when that clears the stage threshold by more than sp_delta (the
// host's bound on the difference between any two summation orders) the stage is decided; otherwise the thread
// walks the window's bits IN STUMP ORDER and adds the leaf values (stage_sum += alpha[rect_sum >=
// norm_threshold], clod.cl:81) — exactly the sequence of f32 additions a single lane would have made.  The
// survivors are compacted across the waves.  Replaces the serial tail (one thin wave, ~300 cycles per stump) of
// the late stages; the wave-split finish (further down) takes the populations above tile_ws_min.
template <bool COUNT, bool STAMPS = true, typename Img>
__device__ __forceinline__ uint32_t tile_stump_parallel(const CascadeArgs& a, const Img& img_by_window,
                                                        const uint32_t* table /* the scale's tile table, global */,
                                                        QEntry* lds_q, unsigned long long* lds_mask, uint32_t* lds_sp,
                                                        uint32_t* lds_cnt, uint32_t T, uint32_t st_begin,
                                                        uint32_t n_stages, uint32_t lane, uint32_t wib,
                                                        unsigned long long& t_last) {
    unsigned long long sp_acc[5] = {0, 0, 0, 0, 0};
#define SPSTAMP(ph) do { if (VJ_STAMPS && STAMPS && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); sp_acc[ph] += t_ - t_last; t_last = t_; } } while (0)
#define SPFLUSH() do { if (VJ_STAMPS && STAMPS && threadIdx.x == 0) { for (int i_ = 0; i_ < 5; ++i_) atomicAdd(a.stage_entered + 54 + i_, sp_acc[i_]); } } while (0)
    const auto img = img_by_window.by_stump();
    kptr<StageDev> stages = as_k(a.stages);
    kptr<uint32_t> blocks = as_k(reinterpret_cast<const uint32_t*>(a.sp_blocks));   // {first_node, desc} pairs
    const uint32_t tid = wib * 64u + lane;
    constexpr uint32_t PITCH = TILE_SP_BLOCK + 1u;   // odd pitch: the field-major copy is written conflict-free
    constexpr uint32_t TABSZ = TILE_SP_FIELDS * PITCH;
    constexpr uint32_t NT = TILE_WAVES * 64u;
    constexpr uint32_t MAXB = TILE_SP_MAX_BLOCKS;
    constexpr int DEPTH = 4;                                         // record blocks in flight from global memory
    uint32_t* lds_tab = lds_sp;                                      // two buffers of TABSZ dwords
    uint32_t* lds_lx = lds_sp + 2u * TABSZ;   // {left, right} bit patterns of every stump of the stage
    float* lds_part = reinterpret_cast<float*>(lds_mask + TILE_SP_MAX_WINDOWS * MAXB);   // butterfly partial sums
    const uint32_t g_end = a.n_sp_blocks;
    uint32_t g = stages[st_begin].sp_first;   // running block number over all stages
    // records of block x: 16 dwords per node, thread t fetches dwords t and t + 512 of the block
    uint32_t pre0[DEPTH], pre1[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
        pre0[u] = 0u;
        pre1[u] = 0u;
        if (g + (uint32_t)u < g_end) {
            const uint32_t jn = blocks[2u * (g + (uint32_t)u) + 1u] & 0xffu;
            const uint32_t* src = table + (size_t)blocks[2u * (g + (uint32_t)u)] * 16u;
            if (tid < jn * 16u) pre0[u] = src[tid];
            if (tid + NT < jn * 16u) pre1[u] = src[tid + NT];
        }
    }
    if (COUNT && tid == 0) atomicAdd(a.stage_entered + st_begin, (unsigned long long)T);
    while (true) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const uint32_t desc = blocks[2u * g + 1u];
            const uint32_t jn = desc & 0xffu, jb = (desc >> 8) & 0xffu, b = (desc >> 16) & 0xfu, nb = (desc >> 20) & 0xfu;
            const uint32_t s = desc >> 24;
            uint32_t* tab = lds_tab + (g & 1u) * TABSZ;
            // 1. this block's records -> LDS field-major (lane j reads field f at tab[f * PITCH + j]); the leaf
            // values also go to lds_lr in stage order for the accumulation
            {
                const uint32_t i0 = tid, i1 = tid + NT;
                if (i0 < jn * 16u) {
                    const uint32_t f = i0 & 15u, j = i0 >> 4;
                    if (f < (uint32_t)TILE_SP_FIELDS) tab[f * PITCH + j] = pre0[u];
                    if (f == 12u) lds_lx[(jb + j) * 2u] = pre0[u];
                    if (f == 13u) lds_lx[(jb + j) * 2u + 1u] = pre0[u];
                }
                if (i1 < jn * 16u) {
                    const uint32_t f = i1 & 15u, j = i1 >> 4;
                    if (f < (uint32_t)TILE_SP_FIELDS) tab[f * PITCH + j] = pre1[u];
                    if (f == 12u) lds_lx[(jb + j) * 2u] = pre1[u];
                    if (f == 13u) lds_lx[(jb + j) * 2u + 1u] = pre1[u];
                }
            }
            lds_barrier();
            SPSTAMP(0);
            // refill this slot with the block DEPTH ahead (global-memory latency is ~2 us: with one block in
            // flight the thin late stages would run at one block per round trip)
            if (g + DEPTH < g_end) {
                const uint32_t njn = blocks[2u * (g + DEPTH) + 1u] & 0xffu;
                const uint32_t* src = table + (size_t)blocks[2u * (g + DEPTH)] * 16u;
                if (tid < njn * 16u) pre0[u] = src[tid];
                if (tid + NT < njn * 16u) pre1[u] = src[tid + NT];
            }
            // 2. verdict bits of the block
            {
                NodeRecDev r;
#pragma unroll
                for (int f = 0; f < TILE_SP_FIELDS; ++f) r[f] = tab[f * PITCH + (lane < jn ? lane : 0u)];
                r[14] = 0u;
                r[15] = 0u;
                const float thr_node = __uint_as_float(r[11]);
                const float leaf_l = lane < jn ? __uint_as_float(r[12]) : 0.0f;
                const float leaf_r = lane < jn ? __uint_as_float(r[13]) : 0.0f;
                // two windows per iteration: their gathers are independent, so the second window's LDS latency
                // hides behind the first one's arithmetic
                uint32_t w = wib;
                for (; w + TILE_WAVES < T; w += 2u * TILE_WAVES) {
                    const QEntry e0 = lds_q[w], e1 = lds_q[w + TILE_WAVES];   // broadcasts
                    const float s0 = node_rect_sum(img, r, e0.off), s1 = node_rect_sum(img, r, e1.off);
                    const bool right0 = lane < jn && s0 >= thr_node * e0.var;
                    const bool right1 = lane < jn && s1 >= thr_node * e1.var;
                    const unsigned long long m0 = __ballot(right0), m1 = __ballot(right1);
                    const float part0 = wave_sum_to_lane63(right0 ? leaf_r : leaf_l);
                    const float part1 = wave_sum_to_lane63(right1 ? leaf_r : leaf_l);
                    if (lane == 63) {
                        lds_mask[w * MAXB + b] = m0;
                        lds_part[w * MAXB + b] = part0;
                        lds_mask[(w + TILE_WAVES) * MAXB + b] = m1;
                        lds_part[(w + TILE_WAVES) * MAXB + b] = part1;
                    }
                }
                if (w < T) {
                    const QEntry e = lds_q[w];   // broadcast
                    const bool right = lane < jn && node_rect_sum(img, r, e.off) >= thr_node * e.var;
                    const unsigned long long m = __ballot(right);
                    // the block's leaf values summed across the lanes (DPP butterfly order — NOT the cascade's
                    // order; it only feeds the fast decision below)
                    const float part = wave_sum_to_lane63(right ? leaf_r : leaf_l);
                    if (lane == 63) {
                        lds_mask[w * MAXB + b] = m;
                        lds_part[w * MAXB + b] = part;
                    }
                }
            }
            ++g;
            SPSTAMP(1);
            if (b + 1u == nb) {   // last block of stage s (uniform)
                lds_barrier();   // every verdict of the stage is in lds_mask
                SPSTAMP(2);
                // 3. the stage decision: thread t owns window t.  The sequential stage sum (stage_sum += alpha in
                // stump order, clod.cl:81) and the butterfly-order sum of the same values differ by at most
                // sp_delta (a rigorous a-priori bound, computed per stage on the host), so when the butterfly sum
                // clears the stage threshold by more than sp_delta either way the reference's comparison is
                // decided; only the rare windows inside the band walk their verdict bits in stump order.
                float sum = 0.0f;
                bool decided_pass = false;
                if (tid < T) {
                    float approx = 0.0f;
                    for (uint32_t bb = 0; bb < nb; ++bb) approx += lds_part[tid * MAXB + bb];
                    const float thr_s = stages[s].threshold, delta = stages[s].sp_delta;
                    const float d = approx - thr_s;
                    const bool clear = d > delta || d < -delta;
                    decided_pass = d > delta;
                    if (!clear) {
                        // exact: sign-extend each verdict bit, pick left or right with and/xor, add in order
                        const uint2* lx = reinterpret_cast<const uint2*>(lds_lx);
                        uint32_t k0 = 0;
                        for (uint32_t bb = 0; bb < nb; ++bb) {
                            const uint32_t bjn = blocks[2u * (g - nb + bb) + 1u] & 0xffu;
                            const unsigned long long m = lds_mask[tid * MAXB + bb];
                            for (uint32_t k = 0; k < bjn; ++k) {
                                const uint2 v = lx[k0 + k];
                                sum += (m >> k) & 1ull ? __uint_as_float(v.y) : __uint_as_float(v.x);
                            }
                            k0 += bjn;
                        }
                        decided_pass = sum >= thr_s;
                    }
                }
                SPSTAMP(3);
                // 4. survivors: compact lds_q across the waves
                const bool pass = tid < T && decided_pass;
                const QEntry e = lds_q[tid < T ? tid : 0u];
                const unsigned long long mask = __ballot(pass);
                if (lane == 0) lds_cnt[1u + wib] = (uint32_t)__popcll(mask);
                lds_barrier();   // every entry is in registers, every wave's count is published
                uint32_t before = 0, total = 0;
#pragma unroll
                for (uint32_t w = 0; w < TILE_WAVES; ++w) {
                    const uint32_t c = lds_cnt[1u + w];
                    before += w < wib ? c : 0u;
                    total += c;
                }
                if (pass) lds_q[before + mbcnt(mask)] = e;
                T = __builtin_amdgcn_readfirstlane(total);
                lds_barrier();   // lds_q is repacked; lds_cnt / lds_mask / lds_lr may be rewritten
                SPSTAMP(4);
                if (T == 0u || s + 1u >= n_stages) { SPFLUSH(); return T; }
                if (COUNT && tid == 0) atomicAdd(a.stage_entered + s + 1u, (unsigned long long)T);
            }
        }
    }
}

// Two consecutive stumps on one window with every gather of both in flight together.  The third
// rectangles are read for both when either stump has one (an absent rectangle has lt = dh = dw = 0:
// four reads of the window's own origin, harmless) but only added where the weight is non-zero, as
// in node_rect_sum.
template <typename Img>
__device__ __forceinline__ void node_rect_sum_pair(const Img& img, const NodeRecDev& ra, const NodeRecDev& rb, uint32_t off,
                                                   float& sum_a, float& sum_b) {
    uint32_t v[2][3][4];
    const NodeRecDev* rr[2] = {&ra, &rb};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const NodeRecDev& r = *rr[p];
        const uint32_t dw0 = (uint32_t)(int32_t)(int16_t)(r[6] & 0xffffu), dw1 = (uint32_t)((int32_t)r[6] >> 16);
        v[p][0][0] = img.ld(off, r[0]);
        v[p][0][1] = img.ld(off, r[0] + dw0);
        v[p][0][2] = img.ld(off, r[0] + r[3]);
        v[p][0][3] = img.ld(off, r[0] + r[3] + dw0);
        v[p][1][0] = img.ld(off, r[1]);
        v[p][1][1] = img.ld(off, r[1] + dw1);
        v[p][1][2] = img.ld(off, r[1] + r[4]);
        v[p][1][3] = img.ld(off, r[1] + r[4] + dw1);
    }
    float out[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const NodeRecDev& r = *rr[p];
        const uint32_t r0 = v[p][0][0] - v[p][0][1] - v[p][0][2] + v[p][0][3];
        const uint32_t r1 = v[p][1][0] - v[p][1][1] - v[p][1][2] + v[p][1][3];
        out[p] = (float)r0 * __uint_as_float(r[8]);
        out[p] += (float)r1 * __uint_as_float(r[9]);
    }
    const float wa2 = __uint_as_float(ra[10]), wb2 = __uint_as_float(rb[10]);
    if (wa2 != 0.0f || wb2 != 0.0f) {   // uniform
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const NodeRecDev& r = *rr[p];
            const uint32_t dw2 = (uint32_t)(int32_t)(int16_t)(r[7] & 0xffffu);
            v[p][2][0] = img.ld(off, r[2]);
            v[p][2][1] = img.ld(off, r[2] + dw2);
            v[p][2][2] = img.ld(off, r[2] + r[5]);
            v[p][2][3] = img.ld(off, r[2] + r[5] + dw2);
        }
        if (wa2 != 0.0f) out[0] += (float)(v[0][2][0] - v[0][2][1] - v[0][2][2] + v[0][2][3]) * wa2;
        if (wb2 != 0.0f) out[1] += (float)(v[1][2][0] - v[1][2][1] - v[1][2][2] + v[1][2][3]) * wb2;
    }
    sum_a = out[0];
    sum_b = out[1];
}

// A two-node tree (root r0, its only node child r1, leaves elsewhere — every tree of frontalface_alt2) on
// one window: icvEvalHidHaarClassifier's walk (tempcv.cpp:771-792: idx = sum < t ? left : right) with both
// nodes' gathers in flight together.  Returns the leaf value; `code` names the leaf (bit 1: reached through
// the child, bit 0: right side) so that an ordered replay can pick the same value again.
template <typename Img>
__device__ __forceinline__ float tree2_value(const Img& img, const NodeRecDev& r0, const NodeRecDev& r1, uint32_t off,
                                             float var, uint32_t& code) {
    float s0, s1;
    node_rect_sum_pair(img, r0, r1, off, s0, s1);
    const uint32_t flags0 = r0[7] >> 16;
    const bool left0 = s0 < __uint_as_float(r0[11]) * var;
    const bool left1 = s1 < __uint_as_float(r1[11]) * var;
    const bool to_child = left0 ? (flags0 & 1u) != 0u : (flags0 & 2u) != 0u;
    const float leaf0 = left0 ? __uint_as_float(r0[12]) : __uint_as_float(r0[13]);
    const float leaf1 = left1 ? __uint_as_float(r1[12]) : __uint_as_float(r1[13]);
    code = to_child ? (left1 ? 2u : 3u) : (left0 ? 0u : 1u);
    return to_child ? leaf1 : leaf0;
}

// One stage of two-node trees on NC chunks of windows: the two records of a tree are fetched once for all
// chunks (next tree prefetched); per window the values are added in tree order, as stage_sum_trees does.
template <int NC, bool COUNT, typename Img>
__device__ __forceinline__ void stage_sum_tree2_multi(const Img& img, kptr<NodeRecDev> tab, uint32_t n_trees,
                                                      const uint32_t (&off)[NC], const float (&var)[NC],
                                                      float (&stage_sum)[NC], uint32_t (*cnt)[2]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) stage_sum[c] = 0.0f;
    NodeRecDev r0 = tab[0], r1 = tab[1];
    for (uint32_t t = 0; t < n_trees; ++t) {
        const uint32_t tn = t + 1u < n_trees ? t + 1u : t;
        const NodeRecDev n0 = tab[2u * tn], n1 = tab[2u * tn + 1u];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint32_t code;
            stage_sum[c] += tree2_value(img, r0, r1, off[c], var[c], code);
            if (COUNT && (code & 2u)) {   // the walk went through the child
                cnt[c][0] += 1u;
                cnt[c][1] += __uint_as_float(r1[10]) != 0.0f ? 3u : 2u;
            }
        }
        r0 = n0;
        r1 = n1;
    }
}

// Wave-split finish of a tile (stump cascades).  T <= TILE_WS_MAX_WINDOWS packed survivors sit in
// lds_q[0, T): c = ceil(T / 64) chunks.  Thin tiles are latency-bound when one wave walks a whole
// stage for its chunk while the others idle, so the stage's stumps are split into K = 8 / c
// contiguous ranges and wave w evaluates range w / c on chunk w % c, one window per lane, records
// through the scalar cache as in the dense sweep.  Each wave leaves, per window, the f32 sum of its
// range (added in stump order from 0) and one verdict bit per stump.  The stage decision adds the
// K range sums: that is the stage's leaf values in a different association than the reference's
// single running sum, so it is only trusted when it clears the threshold by more than sp_delta (the
// host's a-priori bound on the difference between ANY two summation orders of the stage); windows
// inside the band replay their verdict bits in stump order — the reference's exact sequence of f32
// additions (clod.cl:81).  K = 1 is the plain sequential sum.  Results are bit-identical either way.
template <bool COUNT, bool TREE2, typename Img>
__device__ __forceinline__ uint32_t tile_wave_split(const CascadeArgs& a, const Img& img, kptr<NodeRecDev> table,
                                                    QEntry* lds_q, uint32_t* lds_cnt, uint32_t T, uint32_t& st_io,
                                                    uint32_t n_stages, uint32_t lane, uint32_t wib,
                                                    unsigned long long& t_last, QEntry* fail_list = nullptr,
                                                    uint32_t* fail_n = nullptr) {
    unsigned long long sp_acc[5] = {0, 0, 0, 0, 0};
    constexpr bool STAMPS = true;
    kptr<StageDev> stages = as_k(a.stages);
    // scratch behind the packed entries: per producing wave 64 range sums + 4 x 64 verdict words
    uint32_t* lds_x = reinterpret_cast<uint32_t*>(lds_q + TILE_WS_MAX_WINDOWS);
    uint32_t pos = st_io;   // position in the sweep order (StageDev::order); a linear cascade's order is 0, 1, 2, ...
    // below tile_ws_min windows a chunk's lanes are mostly empty: the caller continues stump-parallel
    for (; pos < n_stages && T != 0u && T >= a.tile_ws_min; ++pos) {
        const uint32_t s = a.identity_order != 0u ? pos : stages[pos].order;   // (no dependent load for linear cascades)
        if (COUNT && threadIdx.x == 0) atomicAdd(a.stage_entered + s, (unsigned long long)T);
        // items of a stage: stumps, or two-node trees (TREE2: records 2t and 2t+1, a 2-bit leaf code per tree)
        constexpr uint32_t PER_WORD = TREE2 ? 16u : 32u;
        const uint32_t n = TREE2 ? stages[s].n_nodes >> 1 : stages[s].n_nodes;
        const float thr_s = stages[s].threshold, delta = stages[s].sp_delta;
        kptr<NodeRecDev> tab = table + stages[s].first_node;
        const uint32_t c = (T + 63u) >> 6;
        uint32_t K = (uint32_t)TILE_WAVES / c;
        uint32_t rs = (n + K - 1u) / K;
        rs = (rs + 1u) & ~1u;                      // even: pairs never straddle two ranges
        if (rs > 4u * PER_WORD || K == 1u) { K = 1u; rs = n; }
        const uint32_t chunk = wib % c, range = wib / c;   // uniform
        const uint32_t i = chunk * 64u + lane;
        const bool valid = i < T;
        const QEntry e = lds_q[valid ? i : 0u];
        bool pass = false;
        if (range < K) {
            if (K == 1u) {
                if (TREE2) {
                    const uint32_t off1[1] = {e.off};
                    const float var1[1] = {e.var};
                    float sum1[1];
                    uint32_t cnt1[1][2] = {{0u, 0u}};
                    stage_sum_tree2_multi<1, COUNT>(img, tab, n, off1, var1, sum1, cnt1);
                    if (COUNT) tree_count_flush(a, valid ? cnt1[0][0] : 0u, valid ? cnt1[0][1] : 0u, lane);
                    pass = valid && sum1[0] >= thr_s;
                } else {
                    pass = valid && stage_sum_stumps(img, tab, n, e.off, e.var) >= thr_s;
                }
            } else {
                const uint32_t j0 = min(range * rs, n), j1 = min(j0 + rs, n);
                float psum = 0.0f;
                uint32_t* xw = lds_x + wib * 320u;   // [0,64) sums, [64 + 64 w, ...) verdict word w
                if (TREE2) {
                    uint32_t c_nodes = 0u, c_rects = 0u;
                    for (uint32_t w0 = j0, wd = 0; w0 < j1; w0 += PER_WORD, ++wd) {
                        const uint32_t m = min(PER_WORD, j1 - w0);
                        uint32_t bw = 0u;
                        NodeRecDev r0 = tab[2u * w0], r1 = tab[2u * w0 + 1u];
                        for (uint32_t k = 0; k < m; ++k) {
                            const uint32_t tn = min(w0 + k + 1u, j1 - 1u);
                            const NodeRecDev n0 = tab[2u * tn], n1 = tab[2u * tn + 1u];
                            uint32_t code;
                            psum += tree2_value(img, r0, r1, e.off, e.var, code);
                            if (COUNT && (code & 2u)) {
                                c_nodes += 1u;
                                c_rects += __uint_as_float(r1[10]) != 0.0f ? 3u : 2u;
                            }
                            bw |= code << (2u * k);
                            r0 = n0;
                            r1 = n1;
                        }
                        xw[64u + wd * 64u + lane] = bw;
                    }
                    if (COUNT) tree_count_flush(a, valid ? c_nodes : 0u, valid ? c_rects : 0u, lane);
                } else {
                    for (uint32_t w0 = j0, wd = 0; w0 < j1; w0 += 32u, ++wd) {
                        const uint32_t m = min(32u, j1 - w0);
                        uint32_t bw = 0u;
                        uint32_t k = 0;
                        NodeRecDev ra = tab[w0], rb = tab[min(w0 + 1u, j1 - 1u)];
                        for (; k + 1u < m; k += 2u) {
                            // the next pair's records travel while this pair is evaluated
#if VJ_SCHED_PREFETCH
                            NodeRecDev na = rec_fetch(tab + min(w0 + k + 2u, j1 - 1u)), nb = rec_fetch(tab + min(w0 + k + 3u, j1 - 1u));
#else
                            const NodeRecDev na = tab[min(w0 + k + 2u, j1 - 1u)], nb = tab[min(w0 + k + 3u, j1 - 1u)];
#endif
                            float sa, sb;
                            node_rect_sum_pair(img, ra, rb, e.off, sa, sb);
                            const bool pa = sa >= __uint_as_float(ra[11]) * e.var, pb = sb >= __uint_as_float(rb[11]) * e.var;
                            psum += pa ? __uint_as_float(ra[13]) : __uint_as_float(ra[12]);
                            psum += pb ? __uint_as_float(rb[13]) : __uint_as_float(rb[12]);
                            bw |= (pa ? 1u : 0u) << k;
                            bw |= (pb ? 2u : 0u) << k;
#if VJ_SCHED_PREFETCH
                            rec_arrived(na, nb);
#endif
                            ra = na;
                            rb = nb;
                        }
                        if (k < m) {   // odd tail (only the last word of the stage's last range)
                            const bool pa = node_rect_sum(img, ra, e.off) >= __uint_as_float(ra[11]) * e.var;
                            psum += pa ? __uint_as_float(ra[13]) : __uint_as_float(ra[12]);
                            bw |= (pa ? 1u : 0u) << k;
                        }
                        xw[64u + wd * 64u + lane] = bw;
                    }
                }
                xw[lane] = __float_as_uint(psum);
            }
        }
        SPSTAMP(0);
        if (K > 1u) {
            lds_barrier();   // every range sum and verdict word of the stage is in LDS
            SPSTAMP(1);
            if (wib < c) {   // range 0's wave decides its chunk
                float approx = 0.0f;
                for (uint32_t r = 0; r < K; ++r) approx += __uint_as_float(lds_x[(r * c + wib) * 320u + lane]);
                const float d = approx - thr_s;
                const bool clear = d > delta || d < -delta;
                pass = valid && d > delta;
                if (__ballot(valid && !clear) != 0ull) {
                    // replay in stump order; the leaf values come through the scalar cache
                    float sum = 0.0f;
                    for (uint32_t r = 0; r < K; ++r) {
                        const uint32_t j0 = min(r * rs, n), j1 = min(j0 + rs, n);
                        const uint32_t* xw = lds_x + (r * c + wib) * 320u + 64u;
                        for (uint32_t w0 = j0, wd = 0; w0 < j1; w0 += PER_WORD, ++wd) {
                            const uint32_t m = min(PER_WORD, j1 - w0);
                            const uint32_t bw = xw[wd * 64u + lane];
                            if (TREE2) {
                                kptr<uint32_t> lr = reinterpret_cast<kptr<uint32_t>>(tab + 2u * w0);
                                for (uint32_t k = 0; k < m; ++k) {
                                    const uint32_t code = (bw >> (2u * k)) & 3u;
                                    const float l0 = __uint_as_float(lr[k * 32u + 12u]), r0 = __uint_as_float(lr[k * 32u + 13u]);
                                    const float l1 = __uint_as_float(lr[k * 32u + 28u]), r1 = __uint_as_float(lr[k * 32u + 29u]);
                                    sum += code & 2u ? (code & 1u ? r1 : l1) : (code & 1u ? r0 : l0);
                                }
                            } else {
                                kptr<uint32_t> lr = reinterpret_cast<kptr<uint32_t>>(tab + w0);
                                for (uint32_t k = 0; k < m; ++k)
                                    sum += (bw >> k) & 1u ? __uint_as_float(lr[k * 16u + 13u]) : __uint_as_float(lr[k * 16u + 12u]);
                            }
                        }
                    }
                    if (!clear) pass = valid && sum >= thr_s;
                }
            }
        }
        SPSTAMP(2);
        // survivors: compact lds_q across the deciding waves; a stage-tree segment also keeps its rejects, which
        // continue in another chain (fail_list)
        const unsigned long long mask = __ballot(pass);
        const bool rejected = fail_list != nullptr && wib < c && valid && !pass;
        const unsigned long long fmask = __ballot(rejected);
        if (lane == 0) {
            lds_cnt[1u + wib] = (uint32_t)__popcll(mask);
            lds_cnt[20u + wib] = (uint32_t)__popcll(fmask);
        }
        lds_barrier();   // every entry is in registers, every wave's count is published
        SPSTAMP(3);
        uint32_t before = 0, total = 0, fbefore = 0, ftotal = 0;
#pragma unroll
        for (uint32_t w = 0; w < TILE_WAVES; ++w) {
            const uint32_t cw = lds_cnt[1u + w], fw = lds_cnt[20u + w];
            before += w < wib ? cw : 0u;
            total += cw;
            fbefore += w < wib ? fw : 0u;
            ftotal += fw;
        }
        if (fail_list != nullptr) {
            if (rejected) fail_list[*fail_n + fbefore + mbcnt(fmask)] = e;
            *fail_n += __builtin_amdgcn_readfirstlane(ftotal);
        }
        if (pass) lds_q[before + mbcnt(mask)] = e;
        T = __builtin_amdgcn_readfirstlane(total);
        lds_barrier();   // lds_q is repacked; lds_cnt and the scratch may be rewritten
        SPSTAMP(4);
    }
    SPFLUSH();
    st_io = pos;
    return T;
}

// ROI: the tiles come from a device-built list of (region, scale, tile) entries (roi_plan_units) instead of a frame's own
// tile list: a tile then lies inside a region of interest — its window grid is the region's, its origin the region's
// corner — and the survivors of the last stage go to the region pass's detection list.  Everything else is the same code.
template <bool TREES, bool COUNT, bool STAGED, bool ROI>
__device__ __forceinline__ void tile_pass_body(const CascadeArgs& a, const RoiArgs* r_) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
    QEntry* lds_q = reinterpret_cast<QEntry*>(lds_dyn);                   // TILE_WAVES * TILE_WAVE_CAP entries
    uint32_t* lds_cnt = lds_dyn + TILE_WAVES * TILE_WAVE_CAP * 2;         // survivors per wave (re-packing)
    uint32_t* lds_tab = lds_dyn + TILE_LDS_HEADER / 4;                    // stump-parallel: one stage's table, field-major
    uint32_t* lds_img = lds_tab + a.tile_sp_pad;                          // the image tile (tile_sp_pad: dwords of lds_tab)
    const uint32_t lane = lane_id();
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    QEntry* q = lds_q + wib * TILE_WAVE_CAP;
    kptr<ScaleDev> scales = as_k(a.scales);
    kptr<UnitDev> units = as_k(a.tile_units);
    const uint32_t total_units = ROI ? min(*r_->n_tiles, r_->max_tiles) : a.n_tile_units * a.n_frames;
    const uint32_t frame_bytes4 = a.frame_elems * 4u;

#define STAMP(ph) do { if (VJ_STAMPS && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(a.stage_entered + 40 + (ph), t_ - t_last); t_last = t_; } } while (0)
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
    if (VJ_STAMPS && threadIdx.x == 0) {   // diagnostic: how many workgroups of this launch are resident at once?
        const unsigned long long v = atomicAdd(a.stage_entered + 38, 1ull) + 1ull;
        atomicMax(a.stage_entered + 39, v);
    }
    // Tiles are handed out dynamically: a workgroup that becomes resident late — e.g. because another kernel
    // holds part of the CU — simply finds fewer tickets left.  The (frame, tile) list is cut into eight
    // contiguous parts, one per XCD (blocks b and b + 8 share an XCD under the observed round-robin placement;
    // speed only): the workgroups of an XCD stage neighbouring tiles of the same frame at the same time, so the
    // halo rows and the squared-sum corners they share are L2 hits.  Each part has its own ticket counter (one
    // counter would also be a hot spot for 512 pullers); a workgroup whose part is used up steals from the
    // others.  The next ticket is drawn while the current tile is processed.
    const uint32_t my_xcd = blockIdx.x & 7u;
    auto part_begin = [&](uint32_t x) { return (uint32_t)((unsigned long long)total_units * x / 8u); };
    auto seeds = [&](uint32_t x) { return gridDim.x > x ? (gridDim.x - x + 7u) >> 3 : 0u; };   // blocks on "XCD" x
    uint32_t cur_part = my_xcd;
    auto draw = [&]() -> uint32_t {   // thread 0 only
        for (uint32_t tries = 0; tries < 8u; ++tries) {
            const uint32_t x = (cur_part + tries) & 7u;
            const uint32_t t = atomicAdd(a.tile_ticket + x, 1u);
            const uint32_t cand = part_begin(x) + seeds(x) + t;
            if (cand < part_begin(x + 1u)) {
                cur_part = x;
                return cand;
            }
        }
        return total_units;
    };
    uint32_t u = part_begin(my_xcd) + (blockIdx.x >> 3);
    if (u >= part_begin(my_xcd + 1u)) {   // more workgroups than tiles in this part: start by stealing
        if (threadIdx.x == 0) lds_cnt[TILE_WAVES + 8] = draw();
        __syncthreads();
        u = __builtin_amdgcn_readfirstlane(lds_cnt[TILE_WAVES + 8]);
    }
    while (u < total_units) {
        uint32_t next_u = 0;
        if (threadIdx.x == 0) next_u = draw();
        uint32_t frame, slot, ix0, iy0, nx, ny, ox = 0, oy = 0, roi = 0;
        if constexpr (ROI) {
            kptr<RoiTile> rt = as_k(r_->tiles);
            roi = rt[u].roi;
            slot = rt[u].slot;
            ix0 = rt[u].first & 0xffffu;
            iy0 = rt[u].first >> 16;
            nx = rt[u].nxy & 0xffffu;
            ny = rt[u].nxy >> 16;
            const RoiDev R = r_->rois[roi];
            frame = __builtin_amdgcn_readfirstlane((uint32_t)R.frame);
            ox = __builtin_amdgcn_readfirstlane((uint32_t)R.x);
            oy = __builtin_amdgcn_readfirstlane((uint32_t)R.y);
        } else {
            frame = u / a.n_tile_units;
            const uint32_t r = u - frame * a.n_tile_units;
            slot = units[r].scale;
            ix0 = units[r].first & 0xffffu;
            iy0 = units[r].first >> 16;
            nx = scales[slot].nx;
            ny = scales[slot].ny;
        }
        const float step = scales[slot].step;
        const uint32_t pos_base = scales[slot].pos_base;
        uint32_t tw = scales[slot].tile_tw, th = scales[slot].tile_th;
        const uint32_t pitch = STAGED ? scales[slot].tile_pitch : 0u;
        uint32_t rows = STAGED ? scales[slot].tile_rows : 0u, cols = pitch;   // rows / columns of the image tile that are staged
        if constexpr (ROI) {
            // a region's tile may be smaller than the scale's: fewer rows and columns to stage.  The scale's tile spans
            // ceil((t - 1) * step) + margin pixels (host, double); floor(...) + 2 of the smaller shape against floor(...) of the
            // full one never trims a pixel the smaller tile reads (one spare row / column at most)
            const uint32_t tw_s = tw, th_s = th;
            tw = as_k(r_->tiles)[u].twh & 0xffffu;
            th = as_k(r_->tiles)[u].twh >> 16;
            const uint32_t full_y = (uint32_t)((float)(th_s - 1u) * step), need_y = (uint32_t)((float)(th - 1u) * step) + 2u;
            const uint32_t full_x = (uint32_t)((float)(tw_s - 1u) * step), need_x = (uint32_t)((float)(tw - 1u) * step) + 2u;
            if (need_y < full_y) rows -= full_y - need_y;
            if (need_x < full_x) cols -= full_x - need_x;
        }
        const uint32_t frame_bytes = frame * frame_bytes4;   // < 2^32, checked on the host
        const size_t frame_off = (size_t)frame * a.frame_elems;
        const rsrc_t sum_f = make_rsrc(a.sum + frame_off, frame_bytes4);
        const rsrc_t sq_f = make_rsrc(a.sqsum + frame_off, frame_bytes4 * 2u);
        // tile origin in the image: the first window's origin (same expression as below)
        const uint32_t x0 = ox + __builtin_amdgcn_readfirstlane(window_pos(a, pos_base, ix0, step));
        const uint32_t y0 = oy + __builtin_amdgcn_readfirstlane(window_pos(a, pos_base, iy0, step));

        __syncthreads();  // the previous tile's gathers are finished
        STAMP(0);
        // stage the tile: rows round-robin over the waves, 64 consecutive dwords per instruction,
        // straight into LDS (buffer_load ... lds: no VGPR round trip, so every load of the tile is in
        // flight at once instead of one load-wait-store per 256 bytes); the barrier drains them
        const uint32_t half = STAGED ? scales[slot].tile_half : 0u;
        const uint32_t x4 = STAGED ? scales[slot].tile_x4 : 0u;
        for (uint32_t rr = wib; STAGED && rr < rows; rr += TILE_WAVES) {
            const uint32_t g_row = ((y0 + rr) * a.stride + x0) * 4u;   // uniform
            if (half == 0u && x4 != 0u) {
                // 16 bytes per lane, 1 KiB per instruction: a quarter of the texture-address work of the dword form
                // (which the global-gather chain on the same CU is competing for)
                for (uint32_t c0 = 0; c0 < cols; c0 += 256u) {
                    const uint32_t soff = __builtin_amdgcn_readfirstlane(g_row + c0 * 4u);
                    if (c0 + lane * 4u < cols)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            sum_f, (__attribute__((address_space(3))) uint32_t*)(lds_img + rr * pitch + c0), 16,
                            lane * 16u, soff, 0, 0);
                }
            } else if (half == 0u) {
                for (uint32_t c0 = 0; c0 < cols; c0 += 64u) {
                    const uint32_t soff = __builtin_amdgcn_readfirstlane(g_row + c0 * 4u);   // keep it scalar
                    if (c0 + lane < cols)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            sum_f, (__attribute__((address_space(3))) uint32_t*)(lds_img + rr * pitch + c0), 4,
                            lane * 4u, soff, 0, 0);
                }
            } else {
                // de-interleave while staging: plane 0 takes image columns 0, 2, 4, ..., plane 1 the odd ones
                // (LDS destinations stay lane-contiguous, the sources are 8 bytes apart)
                for (uint32_t plane = 0; plane < 2u; ++plane) {
                    const uint32_t n_cols = min(plane == 0u ? half : pitch - half, (cols + 1u) / 2u + 1u);
                    for (uint32_t c0 = 0; c0 < n_cols; c0 += 64u) {
                        const uint32_t soff = __builtin_amdgcn_readfirstlane(g_row + (c0 * 2u + plane) * 4u);
                        if (c0 + lane < n_cols)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                sum_f,
                                (__attribute__((address_space(3))) uint32_t*)(lds_img + rr * pitch + plane * half + c0), 4,
                                lane * 8u, soff, 0, 0);
                    }
                }
            }
        }
        // while the tile is in flight: this wave's share of the tile's tw*th windows (a run of consecutive
        // tile-local indices, at least one full wave per wave), their positions, and the four squared-sum
        // corners of each (HBM, 8 bytes per lane) — all issued before the barrier that drains the staging
        q = lds_q + wib * TILE_WAVE_CAP;   // (re-packing below moves the wave's queue base)
        const uint32_t n_tile = tw * th;
        const uint32_t per_wave = max(64u, (n_tile + TILE_WAVES - 1u) / TILE_WAVES);   // <= TILE_WAVE_CAP (host)
        const uint32_t t_begin = min(wib * per_wave, n_tile), t_end = min(t_begin + per_wave, n_tile);
        const uint32_t te_lt = scales[slot].te_lt * 4u, te_dh = scales[slot].te_dh * 4u, e_dw = scales[slot].e_dw;
        const uint32_t te_dw = (uint32_t)scales[slot].te_dw * 4u;
        const uint32_t e_lt = scales[slot].e_lt, e_dh = scales[slot].e_dh;
        const float area = scales[slot].area;
        constexpr int NCH = TILE_WAVE_CAP / 64;
        uint32_t w_lo4[NCH];
        uint64_t w_q[NCH];
        bool w_valid[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const uint32_t t = t_begin + (uint32_t)k * 64u + lane;
            const uint32_t ty = t / tw, tx = t - ty * tw;
            const uint32_t iy = iy0 + ty, ix = ix0 + tx;
            w_valid[k] = t < t_end && iy < ny && ix < nx;
            if (!ROI && a.skip_bits != nullptr && w_valid[k])   // uniform test
                w_valid[k] = window_visited(a, frame, scales[slot].skip_base, scales[slot].skip_wpr, nx, ix, iy);
            w_lo4[k] = 0u;
            w_q[k] = 0ull;
            if (w_valid[k]) {
                const uint32_t x = ox + window_pos(a, pos_base, ix, step);
                const uint32_t y = oy + window_pos(a, pos_base, iy, step);
                // byte offset inside the tile (de-interleaved rows: window origins are even columns)
                const uint32_t e = y * a.stride + x;
                // unstaged blocks: byte offset in the batch sum image, as in cascade_pass
                w_lo4[k] = STAGED ? ((y - y0) * pitch + (half ? (x - x0) >> 1 : x - x0)) * 4u : frame_bytes + e * 4u;
                const uint32_t c0 = e_lt, c1 = e_lt + e_dw, c2 = e_lt + e_dh, c3 = e_lt + e_dh + e_dw;
                w_q[k] = ld_u64(sq_f, e * 8u, c0 * 8u) - ld_u64(sq_f, e * 8u, c1 * 8u) - ld_u64(sq_f, e * 8u, c2 * 8u) +
                         ld_u64(sq_f, e * 8u, c3 * 8u);
            }
        }
        __syncthreads();
        STAMP(1);

        // computeVariance (clod.cpp:418-446): pixel sum from the LDS tile, squared sum as loaded above
        uint32_t n = 0;
        typedef typename std::conditional<STAGED, LdsImg, GlobalImg>::type ImgT;
        ImgT img;
        if constexpr (STAGED) img = LdsImg{reinterpret_cast<const char*>(lds_img)};
        else img = GlobalImg{make_rsrc(a.sum, a.sum_bytes)};
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            QEntry en{0u, 0.0f};
            if (w_valid[k]) {
                const uint32_t lo4 = w_lo4[k];
                uint32_t s4;
                if constexpr (STAGED) {
                    s4 = img.ld(lo4, te_lt) - img.ld(lo4, te_lt + te_dw) - img.ld(lo4, te_lt + te_dh) +
                         img.ld(lo4, te_lt + te_dh + te_dw);
                } else {
                    const uint32_t g_lt = e_lt * 4u, g_dw = e_dw * 4u, g_dh = e_dh * 4u;
                    s4 = img.ld(lo4, g_lt) - img.ld(lo4, g_lt + g_dw) - img.ld(lo4, g_lt + g_dh) + img.ld(lo4, g_lt + g_dh + g_dw);
                }
                const float mean = (a.signed_mean ? (float)(int32_t)s4 : (float)s4) / area;
                float variance = (float)w_q[k];
                variance = (variance / area) - (mean * mean);
                en.var = variance >= 0.0f ? sqrtf(variance) : 1.0f;
                en.off = lo4;
            }
            const unsigned long long mask = __ballot(w_valid[k]);
            if (w_valid[k]) q[n + mbcnt(mask)] = en;
            n += (uint32_t)__popcll(mask);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t table_first = STAGED ? scales[slot].tile_table_first : scales[slot].table_first;
        kptr<NodeRecDev> table = as_k(reinterpret_cast<const NodeRecDev*>(a.table)) + table_first;
        // Sweep the cascade stage by stage.  At the stages named by tile_repack_mask and at every
        // pass boundary the tile's survivors are re-packed across its waves into runs of full
        // 64-lane chunks (gathers cost the same for 1 lane as for 64, so 8 thin waves would pay 8x).
        // At a pass boundary the whole workgroup leaves — handing its survivors to that pass's
        // queue — when the boundary lies at or beyond tile_end or fewer than tile_min_lanes
        // windows are left in the tile (the queue passes re-pack windows of the whole batch).
        STAMP(2);
        // hand a wave's survivors on: detections (dest == n_pass) or the global queue of pass boundary `dest`;
        // tile-local offsets become byte offsets in the batch sum image
        auto flush_wave = [&](const QEntry* qq, uint32_t nn, uint32_t dest_) {
            if constexpr (ROI) {   // (one pass: whatever survives is a detection of this region)
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(r_->det_count, nn);
                g = __builtin_amdgcn_readfirstlane(g);
                for (uint32_t i = lane; i < nn; i += 64u) {
                    const uint32_t lo = qq[i].off >> 2;
                    const uint32_t ly = lo / pitch, lc = lo - ly * pitch;
                    const uint32_t lx = half ? lc * 2u : lc;
                    if (g + i < r_->det_cap) r_->det[g + i] = RoiDet{frame_bytes + ((y0 + ly) * a.stride + (x0 + lx)) * 4u, slot, roi};
                }
                return;
            }
            const bool is_det = dest_ == a.n_pass;
            const uint32_t part = frame_part(a, frame);
            uint32_t g = 0;
            if (lane == 0)
                g = is_det ? atomicAdd(a.det_count, nn) : atomicAdd(a.q_pass_count[dest_] + slot * Q_PARTS + part, nn);
            g = __builtin_amdgcn_readfirstlane(g);
            const size_t q_base = (size_t)scales[slot].q_base + (size_t)part * scales[slot].q_cap;
            QEntry* qd = is_det ? nullptr : a.q_pass[dest_];
            for (uint32_t i = lane; i < nn; i += 64u) {
                const QEntry e = qq[i];
                uint32_t off = e.off;
                if (STAGED) {
                    const uint32_t lo = e.off >> 2;
                    const uint32_t ly = lo / pitch, lc = lo - ly * pitch;
                    const uint32_t lx = half ? lc * 2u : lc;   // window origins sit in the even plane
                    off = frame_bytes + ((y0 + ly) * a.stride + (x0 + lx)) * 4u;
                }
                if (is_det) {
                    if (g + i < a.det_cap) a.det[g + i] = DetEntry{off, slot};
                } else {
                    qd[q_base + g + i] = QEntry{off, e.var};
                }
            }
        };
        uint32_t dest = a.n_pass;   // n_pass = ran the whole cascade: survivors are detections
        uint32_t next_p = 1;        // next pass boundary index
        const uint32_t n_stages_total = a.pass_begin[a.n_pass];
        for (uint32_t st = 0; st < n_stages_total; ++st) {
            const bool at_boundary = next_p < a.n_pass && st == a.pass_begin[next_p];
            if (st != 0u && (at_boundary || ((a.tile_repack_mask >> st) & 1ull))) {
                // ---- workgroup-uniform: count, then move every survivor to its packed position
                if (lane == 0) lds_cnt[wib] = n;
                __syncthreads();
                uint32_t before = 0, total = 0;
#pragma unroll
                for (uint32_t w = 0; w < TILE_WAVES; ++w) {
                    const uint32_t c = lds_cnt[w];
                    before += w < wib ? c : 0u;
                    total += c;
                }
                before = __builtin_amdgcn_readfirstlane(before);
                total = __builtin_amdgcn_readfirstlane(total);
                QEntry hold[TILE_WAVE_CAP / 64];
#pragma unroll
                for (uint32_t k = 0; k < TILE_WAVE_CAP / 64; ++k)
                    if (k * 64u + lane < n) hold[k] = q[k * 64u + lane];
                __syncthreads();   // every wave holds its survivors in registers
#pragma unroll
                for (uint32_t k = 0; k < TILE_WAVE_CAP / 64; ++k)
                    if (k * 64u + lane < n) lds_q[before + k * 64u + lane] = hold[k];
                __syncthreads();
                // contiguous shares of whole chunks: wave w owns [w*share, (w+1)*share)
                const uint32_t share = 64u * (((total + 63u) / 64u + TILE_WAVES - 1u) / TILE_WAVES);
                const uint32_t first = min(wib * share, total);
                q = lds_q + first;
                n = min(share, total - first);
                STAMP(3 + min(st, 8u));   // time of stage st-1 (incl. waiting for the slowest wave) + this re-pack
                if (!TREES && a.n_seg != 0u && st == a.tile_end && total != 0u && total <= (uint32_t)TILE_SEG_MAX_WINDOWS) {
                    // Stage tree (frontalface_alt_tree): the linear prefix ends here; the chains that follow run on the
                    // packed survivors with the wave-split machinery — a chain's rejects are collected in LDS and
                    // become the population of the next chain — so the whole tree finishes inside the tile.
                    QEntry* flist = lds_q + (TILE_WS_MAX_WINDOWS + TILE_WAVES * 160);   // behind the wave-split scratch
                    uint32_t T = total, posn = st;
                    for (uint32_t k = 0; k < a.n_seg; ++k) {
                        const bool chained = ((a.seg_chain >> k) & 1u) != 0u;
                        uint32_t F_n = 0;
                        const uint32_t left = tile_wave_split<COUNT, false>(a, img, table, lds_q, lds_cnt, T, posn, a.seg_end[k],
                                                                            lane, wib, t_last, chained ? flist : nullptr, &F_n);
                        if (wib == 0u && left != 0u) flush_wave(lds_q, left, a.n_pass);   // the chain's end accepts
                        if (!chained || F_n == 0u) break;
                        __syncthreads();
                        for (uint32_t i = threadIdx.x; i < F_n; i += TILE_WAVES * 64u) lds_q[i] = flist[i];
                        __syncthreads();
                        T = F_n;
                        posn = a.seg_end[k];
                    }
                    q = lds_q;
                    n = 0u;
                    dest = a.n_pass;
                    STAMP(12);
                    break;
                }
                if ((!TREES || a.tree2) && a.tile_finish == 1u && st >= a.tile_sp_begin && total != 0u && total <= a.tile_ws_max) {
                    // few windows left: finish the whole cascade with the stage's stumps (or two-node trees) split
                    // over the waves
                    uint32_t s_next = st;
                    uint32_t left = tile_wave_split<COUNT, TREES>(a, img, table, lds_q, lds_cnt, total, s_next, n_stages_total,
                                                                  lane, wib, t_last);
                    if (!TREES && left != 0u && s_next < n_stages_total)
                        left = tile_stump_parallel<COUNT, false>(
                            a, img, a.table + (size_t)table_first * 16u, lds_q,
                            reinterpret_cast<unsigned long long*>(lds_q + TILE_SP_MAX_WINDOWS), lds_tab, lds_cnt, left, s_next,
                            n_stages_total, lane, wib, t_last);
                    q = lds_q;
                    n = wib == 0u ? left : 0u;
                    dest = a.n_pass;
                    STAMP(12);
                    break;
                }
                if (!TREES && a.tile_finish == 0u && st >= a.tile_sp_begin && total != 0u && total <= a.tile_sp_max) {
                    // few windows left: finish the whole cascade stump-parallel; survivors are detections
                    const uint32_t left = tile_stump_parallel<COUNT, true>(
                        a, img, a.table + (size_t)table_first * 16u, lds_q,
                        reinterpret_cast<unsigned long long*>(lds_q + TILE_SP_MAX_WINDOWS), lds_tab, lds_cnt, total, st,
                        n_stages_total, lane, wib, t_last);
                    q = lds_q;
                    n = wib == 0u ? left : 0u;
                    dest = a.n_pass;
                    STAMP(12);
                    break;
                }
                if (at_boundary) {
                    if (total < max(a.tile_min_lanes, 1u) || st >= a.tile_end) {
                        dest = next_p;
                        break;
                    }
                    ++next_p;
                }
            }
            if (n != 0u) n = sweep_stages<TREES, COUNT, true>(a, img, table, q, n, lane, st, st + 1u);
        }
        if (n != 0u) flush_wave(q, n, dest);
        STAMP(13);
        __syncthreads();   // the tile is finished: lds_cnt may carry the next ticket
        if (threadIdx.x == 0) lds_cnt[TILE_WAVES + 8] = next_u;
        __syncthreads();
        u = __builtin_amdgcn_readfirstlane(lds_cnt[TILE_WAVES + 8]);
    }
    if (VJ_STAMPS && threadIdx.x == 0) atomicAdd(a.stage_entered + 38, ~0ull);
}

template <bool TREES, bool COUNT, bool STAGED>
__global__ __launch_bounds__(TILE_WAVES * 64) void cascade_tile_pass(CascadeArgs a) {
    tile_pass_body<TREES, COUNT, STAGED, false>(a, nullptr);
}

// The tile kernel inside regions of interest (vj_detect_chain / vj_detect_rois with many or large regions; stump cascades).
template <bool COUNT>
__global__ __launch_bounds__(TILE_WAVES * 64) void cascade_tile_roi_pass(RoiArgs r, CascadeArgs a) {
    tile_pass_body<false, COUNT, true, true>(a, &r);
}

// The tile kernel uses up to the CU's whole 160 KiB of dynamic LDS; HIP caps a kernel at 64 KiB until the attribute
// is raised, and the attribute belongs to the (function, device) pair: vj_env_create calls this once per environment,
// with that environment's device current.
int prepare_tile_kernels() {
    const int max_lds = 160 * 1024;
    const void* fns[] = {(const void*)cascade_tile_roi_pass<false>, (const void*)cascade_tile_roi_pass<true>,
                         (const void*)cascade_tile_pass<false, false, true>, (const void*)cascade_tile_pass<false, true, true>,
                         (const void*)cascade_tile_pass<true, false, true>,  (const void*)cascade_tile_pass<true, true, true>,
                         (const void*)cascade_tile_pass<false, false, false>, (const void*)cascade_tile_pass<false, true, false>};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int launch_cascade_tile_pass(const CascadeArgs& a, bool trees, bool count, bool staged, int n_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    dim3 g(n_blocks), b(TILE_WAVES * 64);
    const size_t lds = a.tile_lds_bytes;
    if (!staged) {   // unstaged blocks: stump cascades only (the host keeps tree cascades on cascade_pass)
        if (count) hipLaunchKernelGGL((cascade_tile_pass<false, true, false>), g, b, lds, stream, a);
        else       hipLaunchKernelGGL((cascade_tile_pass<false, false, false>), g, b, lds, stream, a);
    } else if (trees) {
        if (count) hipLaunchKernelGGL((cascade_tile_pass<true, true, true>), g, b, lds, stream, a);
        else       hipLaunchKernelGGL((cascade_tile_pass<true, false, true>), g, b, lds, stream, a);
    } else {
        if (count) hipLaunchKernelGGL((cascade_tile_pass<false, true, true>), g, b, lds, stream, a);
        else       hipLaunchKernelGGL((cascade_tile_pass<false, false, true>), g, b, lds, stream, a);
    }
    return (int)hipGetLastError();
}

template <bool FROM_GRID, bool TREES>
static void launch_variant(const CascadeArgs& a, bool last, bool count, int n_blocks, hipStream_t stream) {
    dim3 g(n_blocks), b(gather_block_threads(a, false));
    if constexpr (!FROM_GRID && !TREES) {
        if (a.wide_tail != 0u) {   // queue passes of a small batch: TAIL_NW windows per step of the stump-parallel tail
            if (last) {
                if (count) hipLaunchKernelGGL((cascade_pass<false, false, true, true, false, TAIL_NW>), g, b, 0, stream, a);
                else       hipLaunchKernelGGL((cascade_pass<false, false, true, false, false, TAIL_NW>), g, b, 0, stream, a);
            } else {
                if (count) hipLaunchKernelGGL((cascade_pass<false, false, false, true, false, TAIL_NW>), g, b, 0, stream, a);
                else       hipLaunchKernelGGL((cascade_pass<false, false, false, false, false, TAIL_NW>), g, b, 0, stream, a);
            }
            return;
        }
    }
    if (last) {
        if (count) hipLaunchKernelGGL((cascade_pass<FROM_GRID, TREES, true, true, false>), g, b, 0, stream, a);
        else       hipLaunchKernelGGL((cascade_pass<FROM_GRID, TREES, true, false, false>), g, b, 0, stream, a);
    } else {
        if (count) hipLaunchKernelGGL((cascade_pass<FROM_GRID, TREES, false, true, false>), g, b, 0, stream, a);
        else       hipLaunchKernelGGL((cascade_pass<FROM_GRID, TREES, false, false, false>), g, b, 0, stream, a);
    }
}

template <bool FROM_GRID, bool TREES>
static void launch_general(const CascadeArgs& a, bool count, int n_blocks, hipStream_t stream) {
    dim3 g(n_blocks), b(gather_block_threads(a, true));
    if (count) hipLaunchKernelGGL((cascade_pass<FROM_GRID, TREES, true, true, true>), g, b, 0, stream, a);
    else       hipLaunchKernelGGL((cascade_pass<FROM_GRID, TREES, true, false, true>), g, b, 0, stream, a);
}

int launch_cascade_pass(const CascadeArgs& a, bool from_grid, bool trees, bool last, bool count, bool general,
                        int n_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (general) {  // stage-tree cascade: everything (from the grid) or everything after the linear prefix (from its queue)
        if (!last) return (int)hipErrorInvalidValue;
        if (from_grid) {
            if (trees) launch_general<true, true>(a, count, n_blocks, stream);
            else       launch_general<true, false>(a, count, n_blocks, stream);
        } else {
            if (trees) launch_general<false, true>(a, count, n_blocks, stream);
            else       launch_general<false, false>(a, count, n_blocks, stream);
        }
    } else if (from_grid) {
        if (trees) launch_variant<true, true>(a, last, count, n_blocks, stream);
        else       launch_variant<true, false>(a, last, count, n_blocks, stream);
    } else {
        if (trees) launch_variant<false, true>(a, last, count, n_blocks, stream);
        else       launch_variant<false, false>(a, last, count, n_blocks, stream);
    }
    return (int)hipGetLastError();
}

}  // namespace vj
