// Micro-benchmark: can two neighbouring windows share ONE LDS read?  In a de-interleaved step-2 tile the same corner of
// windows x and x + 2 sits in consecutive dwords, so a lane that owns both could fetch the pair with ds_read_b64 — 2 LDS
// cycles per wave-instruction like ds_read_b32 (MI355X_MICROARCH.md, LDS table), i.e. twice the corners per cycle — but only
// every other corner offset leaves the pair 8-byte aligned.  Measured here, 2 workgroups of 8 waves per CU like the
// tile kernel's two-per-CU class, 16 reads in flight per wave:
//   b32      ds_read_b32, lane l reads dword l + k               (the kernel today: one corner per lane)
//   b64a     ds_read_b64, lane l reads dwords 2l, 2l + 1          (aligned pairs)
//   b64u     ds_read_b64, lane l reads dwords 2l + 1, 2l + 2      (pairs at 4 mod 8: what half of the corners would be)
//   b64s     ds_read_b64, compacted pairs (random pair slots)     (what a queue of surviving pairs holds)
// and whether the misaligned form returns the right dwords at all.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_b64.hip -o /tmp/lds_b64 && /tmp/lds_b64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int LDS_DW = 12 * 1024;   // 48 KiB per workgroup: two per CU

template <int MODE>
__global__ __launch_bounds__(512) void lds_loop(uint32_t iters, uint32_t* out, unsigned long long* cycles) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (uint32_t)LDS_DW; i += 512u) lds[i] = i;
    __syncthreads();
    uint32_t base;   // byte address of the lane's (pair) slot
    if (MODE == 0) base = (wib * 1024u + lane) * 4u;
    else if (MODE == 1) base = (wib * 1024u + 2u * lane) * 4u;
    else if (MODE == 2) base = (wib * 1024u + 2u * lane + 1u) * 4u;
    else {
        uint32_t h = (blockIdx.x * 8u + wib) * 747796405u + lane * 2891336453u + 1u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        base = (wib * 1024u + (h % 480u) * 2u + (lane & 1u)) * 4u;   // pair slots anywhere in the wave's 4 KiB, either alignment
    }
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t v[16][2];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t a = base + (uint32_t)k * 8u * (MODE == 0 ? 1u : 1u) + ((it & 7u) << 6);
            if (MODE == 0) {
                asm volatile("ds_read_b32 %0, %1" : "=v"(v[k][0]) : "v"(a));
                v[k][1] = 0;
            } else {
                uint64_t w;
                asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(a));
                v[k][0] = (uint32_t)w;
                v[k][1] = (uint32_t)(w >> 32);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k][0] ^ v[k][1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512u + threadIdx.x] = acc;
    if (threadIdx.x == 0) atomicAdd(cycles, t1 - t0);
}

__global__ void check_unaligned(uint32_t* out) {
    __shared__ uint32_t lds[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += 64u) lds[i] = 1000u + i;
    __syncthreads();
    const uint32_t a = (2u * threadIdx.x + 1u) * 4u;
    uint64_t w;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a) : "memory");
    out[2 * threadIdx.x] = (uint32_t)w;
    out[2 * threadIdx.x + 1] = (uint32_t)(w >> 32);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t* d_out;
    unsigned long long* d_cyc;
    hipMalloc(&d_out, (size_t)cus * 2 * 512 * 4);
    hipMalloc(&d_cyc, 8);
    {
        uint32_t h[128];
        hipLaunchKernelGGL(check_unaligned, dim3(1), dim3(64), 0, 0, d_out);
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        bool ok = true;
        for (int l = 0; l < 64; ++l) ok = ok && h[2 * l] == 1000u + 2 * l + 1 && h[2 * l + 1] == 1000u + 2 * l + 2;
        printf("ds_read_b64 at 4 mod 8: %s (lane 0 read %u %u, expected 1001 1002)\n", ok ? "correct dwords" : "WRONG DWORDS", h[0], h[1]);
    }
    const uint32_t iters = 20000;
    const char* names[4] = {"b32  (dword per lane)", "b64a (aligned pairs)", "b64u (pairs at 4 mod 8)", "b64s (scattered pair slots)"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_cyc, 0, 8);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipFuncSetAttribute((const void*)lds_loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DW * 4);
            hipFuncSetAttribute((const void*)lds_loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DW * 4);
            hipFuncSetAttribute((const void*)lds_loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DW * 4);
            hipFuncSetAttribute((const void*)lds_loop<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DW * 4);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(lds_loop<0>, dim3(cus * 2), dim3(512), LDS_DW * 4, 0, iters, d_out, d_cyc);
            if (mode == 1) hipLaunchKernelGGL(lds_loop<1>, dim3(cus * 2), dim3(512), LDS_DW * 4, 0, iters, d_out, d_cyc);
            if (mode == 2) hipLaunchKernelGGL(lds_loop<2>, dim3(cus * 2), dim3(512), LDS_DW * 4, 0, iters, d_out, d_cyc);
            if (mode == 3) hipLaunchKernelGGL(lds_loop<3>, dim3(cus * 2), dim3(512), LDS_DW * 4, 0, iters, d_out, d_cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc = 0;
            hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
            if (rep == 1) {
                // per CU: 16 waves x iters x 16 wave-instructions in cyc / (2 * cus) shader cycles per workgroup
                const double per_wg = (double)cyc / (2.0 * cus);
                const double cu_cycles_per_inst = per_wg / ((double)iters * 16.0 * 16.0);
                printf("%-30s %.3f ms  %.2f CU-cycles per wave-instruction (s_memtime), %.1f B/clk/CU\n", names[mode], ms,
                       cu_cycles_per_inst, (mode == 0 ? 256.0 : 512.0) / cu_cycles_per_inst);
            }
        }
    }
    return 0;
}
