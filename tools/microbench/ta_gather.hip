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
    return 0;
}
