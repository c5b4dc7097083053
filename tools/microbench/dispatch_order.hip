// Where do the first workgroups of a grid land?  A persistent kernel whose waves draw work from a ticket counter
// hands the first tickets to the workgroups that start first; if those sit on few CUs, a pass with fewer chunks than
// waves runs on a fraction of the chip.  2048 workgroups of 192 threads with 12 KiB of LDS (the queue passes' shape):
// every workgroup records its XCC / SE / SH / CU ids and start time, then stays resident for a while.
// Build: hipcc --offload-arch=gfx950 -O2 dispatch_order.hip -o dispatch_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(192) void probe(unsigned* out, unsigned long long* t, int spin) {
    __shared__ unsigned lds[3072];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
        t[blockIdx.x] = t0;
    }
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 7) % 192] == 12345u) out[0] = 0;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048, first = argc > 2 ? atoi(argv[2]) : 552;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, blocks * 8); hipMalloc(&dt, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(192), 0, 0, d, dt, 200000);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(blocks * 2); std::vector<unsigned long long> ht(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
    auto key = [&](int b) { const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 15u; return (xcc << 8) | (((hw >> 13) & 3u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u); };
    std::map<unsigned, int> all, head;
    for (int b = 0; b < blocks; ++b) { all[key(b)]++; if (b < first) head[key(b)]++; }
    std::vector<int> cnt; for (auto& kv : head) cnt.push_back(kv.second);
    std::sort(cnt.begin(), cnt.end());
    printf("%d workgroups on %zu CUs; the first %d sit on %zu CUs, per-CU count min %d median %d max %d\n", blocks, all.size(), first, head.size(),
           cnt.front(), cnt[cnt.size() / 2], cnt.back());
    printf("first 24 workgroups (xcc se sh cu): ");
    for (int b = 0; b < 24; ++b) { const unsigned hw = h[b * 2]; printf("%u/%u/%u/%u ", h[b * 2 + 1] & 15u, (hw >> 13) & 3u, (hw >> 12) & 1u, (hw >> 8) & 15u); }
    printf("\nworkgroups 8, 16, ..., 64 (same XCD as 0?): ");
    for (int b = 8; b <= 64; b += 8) { const unsigned hw = h[b * 2]; printf("%u/%u/%u/%u ", h[b * 2 + 1] & 15u, (hw >> 13) & 3u, (hw >> 12) & 1u, (hw >> 8) & 15u); }
    printf("\n");
    return 0;
}
