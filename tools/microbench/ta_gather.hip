// Micro-benchmark: cost of wave-wide dword gathers through buffer loads on gfx950 as a function of
// the address pattern (what bounds the global-gather chain).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/ta_gather.hip -o /tmp/ta_gather && /tmp/ta_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

using rsrc_t = __amdgpu_buffer_rsrc_t;

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint32_t* base, uint32_t bytes, uint32_t stride, uint32_t iters,
                                              uint32_t span_mask, uint32_t* out) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, bytes, 0x00020000);
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint32_t h = wave * 2654435761u;
    for (uint32_t i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        uint32_t off;
        if (MODE == 0) off = ((h & span_mask) + lane * stride) & ~3u;                              // strided lanes, random wave base
        else off = (((h ^ (lane * 2246822519u)) * 2654435761u) & span_mask) & ~3u;               // every lane random in the span
        // 8 independent loads per iteration at small uniform displacements (like the corners of rectangles)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, (uint32_t)k * 7700u, 0);
    }
    if (acc == 0x12345u) out[0] = acc;
}

// The four squared-sum corners of 64 consecutive step-2 windows (cascade_tile_pass, variance): 8 bytes per lane at a lane stride of
// 16 bytes, the right corners 18 elements to the right of the left ones, the bottom ones 18 rows below.  FORM 0: four strided 8-byte
// loads (what the kernel does).  FORM 1: per row one contiguous 16-byte load per lane (elements 2l, 2l+1) and one more for the few
// lanes beyond 128 elements; the right corners are then lane shifts of the left ones (two ds_bpermute per 64-bit value).
template <int FORM>
__global__ __launch_bounds__(256) void corners(const uint32_t* base, uint32_t bytes, uint32_t iters, uint32_t span_mask, uint32_t* out) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, bytes, 0x00020000);
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t row = 1921u * 8u, e_dw = 18u, e_dh = 18u;
    uint64_t acc = 0;
    uint32_t h = wave * 2654435761u;
    for (uint32_t i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t b0 = __builtin_amdgcn_readfirstlane((h & span_mask) & ~15u);
        if (FORM == 0) {
            const uint32_t off = lane * 16u;
            auto ld = [&](uint32_t u) { const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, off, b0 + u, 0); return (uint64_t)v[0] | (uint64_t)v[1] << 32; };
            acc += ld(0) - ld(e_dw * 8u) - ld(e_dh * row) + ld(e_dh * row + e_dw * 8u);
        } else {
            const uint32_t extra = e_dw / 2u + 1u;
            uint64_t tl = 0, tr = 0;
            uint64_t corner[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t rb = b0 + (q ? e_dh * row : 0u);
                const auto A = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u, rb, 0);
                auto B = A;
                if (lane < extra) B = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u, rb + 1024u, 0);
                const uint32_t j = 2u * lane + e_dw, src = j >> 1, c = (j & 1u) * 2u;
                const uint32_t a_lo = __builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)A[c]), a_hi = __builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)A[c + 1]);
                const uint32_t b_lo = __builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)B[c]), b_hi = __builtin_amdgcn_ds_bpermute((int)((src & 63u) * 4u), (int)B[c + 1]);
                corner[q][0] = (uint64_t)A[0] | (uint64_t)A[1] << 32;
                corner[q][1] = src < 64u ? ((uint64_t)a_lo | (uint64_t)a_hi << 32) : ((uint64_t)b_lo | (uint64_t)b_hi << 32);
            }
            (void)tl; (void)tr;
            acc += corner[0][0] - corner[0][1] - corner[1][0] + corner[1][1];
        }
    }
    if (acc == 0x12345u) out[0] = (uint32_t)acc;
    if (out[1] == 77u) out[2 + (threadIdx.x & 1)] = (uint32_t)(acc >> 3);   // (keeps both forms' results observable)
}

int main() {
    const size_t bytes = 512u << 20;
    uint32_t* d;
    uint32_t* o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 64);
    hipMemset(d, 1, bytes);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Case { const char* name; int mode; uint32_t stride; uint32_t span; };
    const Case cases[] = {
        {"stride 4 B (coalesced), base random in 8 MB", 0, 4, (8u << 20) - 1},
        {"stride 8 B", 0, 8, (8u << 20) - 1},
        {"stride 20 B (s=5 windows)", 0, 20, (8u << 20) - 1},
        {"stride 40 B", 0, 40, (8u << 20) - 1},
        {"stride 80 B", 0, 80, (8u << 20) - 1},
        {"stride 160 B", 0, 160, (8u << 20) - 1},
        {"lanes random in 16 KB", 1, 0, (16u << 10) - 1},
        {"lanes random in 256 KB", 1, 0, (256u << 10) - 1},
        {"lanes random in 8 MB", 1, 0, (8u << 20) - 1},
        {"lanes random in 256 MB", 1, 0, (256u << 20) - 1},
    };
    for (int wpc : {4, 16}) {
        for (const Case& c : cases) {
            const uint32_t iters = 2000;
            const int blocks = cus * wpc / 4;
            auto run = [&]() {
                if (c.mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)bytes, c.stride, iters, c.span, o);
                else hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)bytes, c.stride, iters, c.span, o);
            };
            run();
            hipEventRecord(e0);
            run();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double wave_loads_per_cu = (double)wpc * iters * 8;
            const double cyc = ms * 1e-3 * 2.1e9 / wave_loads_per_cu;   // assuming ~2.1 GHz
            printf("%2d waves/CU  %-46s %8.3f ms  %6.1f cycles per wave-load per CU (@2.1 GHz)\n", wpc, c.name, ms, cyc);
        }
    }
    for (int wpc : {4, 16}) {
        for (int form = 0; form < 2; ++form) {
            const uint32_t iters = 4000;
            const int blocks = cus * wpc / 4;
            auto run = [&]() {
                if (form == 0) hipLaunchKernelGGL(corners<0>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)bytes, iters, (64u << 20) - 1, o);
                else hipLaunchKernelGGL(corners<1>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)bytes, iters, (64u << 20) - 1, o);
            };
            run();
            hipEventRecord(e0);
            run();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%2d waves/CU  squared-sum corners of 64 step-2 windows, %-44s %8.3f ms  %6.1f cycles per chunk of 64 windows per CU\n", wpc,
                   form == 0 ? "four strided 8-byte loads" : "two contiguous 16-byte loads + lane shifts", ms, ms * 1e-3 * 2.1e9 / ((double)wpc * iters));
        }
    }
    return 0;
}
