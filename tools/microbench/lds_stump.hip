// Micro-benchmark: what one stump costs in the dense loop of cascade_tile_pass (4 chunks of 64 windows per stump,
// node record through the scalar cache, 8 LDS corner gathers per window) as a function of
//   * the window offsets the lanes hold (consecutive = a dense row; "thinned" = survivors of a compaction at rate p in
//     row-major order, what a compacted queue holds; random = unrelated windows)
//   * what is left out: the per-corner address adds (NOADD), the LDS reads (NOLDS), the arithmetic (NOALU)
// so that the LDS bank-conflict share and the VALU share of the loop can be read off directly.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/microbench/lds_stump.hip -o /tmp/lds_stump && /tmp/lds_stump
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>

typedef uint32_t Rec __attribute__((ext_vector_type(16)));
template <typename T>
using kptr = const T __attribute__((address_space(4)))*;

constexpr int PITCH = 149;          // dwords per tile row (s = 1 tile of 64 x 32 windows, de-interleaved)
constexpr int ROWS = 85;
constexpr int TILE_DW = PITCH * ROWS;

enum { FULL = 0, NOADD = 1, NOLDS = 2, NOALU = 3, ADDTID = 4 };   // ADDTID: dense rows only — ds_read_addtid_b32 (address = M0 + lane * 4), no VALU address adds

template <int VAR>
__global__ __launch_bounds__(512) void stump_loop(const Rec* table, uint32_t n_recs, uint32_t iters, uint32_t pattern, uint32_t keep_pm,
                                                  float* out, const uint32_t* binned) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (uint32_t)TILE_DW; i += 512u) lds[i] = i * 2654435761u >> 8;
    __syncthreads();
    // window offsets of this wave's 4 chunks (bytes)
    uint32_t off[4];
    float var[4];
    uint32_t next = wib * 256u;   // the wave's 256 windows: tile-local indices [wib * 256, ...)
    uint32_t h = (blockIdx.x * 8u + wib) * 747796405u + 1u;
    for (int c = 0; c < 4; ++c) {
        uint32_t t;
        if (pattern == 0) t = wib * 256u + c * 64u + lane;                         // dense rows
        else if (pattern == 1) {                                                    // survivors at rate keep_pm / 1000, in order
            // every lane walks the same pseudo-random keep sequence; lane l takes the (c * 64 + l)-th kept index
            uint32_t want = c * 64u + lane, idx = wib * 256u, kept = 0, hh = h;
            while (true) {
                hh = hh * 1664525u + 1013904223u;
                if ((hh >> 8) % 1000u < keep_pm) {
                    if (kept == want) break;
                    ++kept;
                }
                ++idx;
            }
            t = idx % 2048u;
            (void)next;
        } else if (pattern == 3) {
            // binned: the wave's 256 slots hold survivors (rate keep_pm / 1000 of the wave's share of the tile, wrapped) laid out
            // residue-major and dealt round-robin over the 8 groups of 32 lanes, so that a group holds at most
            // ceil(c_r / 8) windows of bank residue r (what a residue-aware re-pack would produce)
            t = binned[(wib * 256u + c * 64u + lane)];
        } else {
            uint32_t x = (h ^ (lane * 2246822519u) ^ (c * 3266489917u)) * 2654435761u;
            t = (x >> 7) % 2048u;                                                   // unrelated windows
        }
        const uint32_t ty = t / 64u, tx = t % 64u;
        off[c] = (2u * ty * PITCH + tx) * 4u;   // step-2 scale, de-interleaved: window (tx, ty) starts at row 2 ty, dword tx
        var[c] = 1.0f + (float)(t & 15u);
    }
    kptr<Rec> tab = (kptr<Rec>)(uintptr_t)table;
    float acc[4] = {0, 0, 0, 0};
    const char* base = reinterpret_cast<const char*>(lds);
    for (uint32_t it = 0; it < iters; ++it) {
        Rec r = tab[0];
        for (uint32_t j = 0; j < n_recs; ++j) {
            const Rec rn = tab[j + 1 < n_recs ? j + 1 : j];
            const uint32_t lt0 = r[0], lt1 = r[1], dh0 = r[3], dh1 = r[4];
            const uint32_t dw0 = r[6] & 0xffffu, dw1 = r[6] >> 16;
            const float w0 = __uint_as_float(r[8]), w1 = __uint_as_float(r[9]);
            const float thr = __uint_as_float(r[11]), left = __uint_as_float(r[12]), right = __uint_as_float(r[13]);
            uint32_t v[4][8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t a[8] = {lt0, lt0 + dw0, lt0 + dh0, lt0 + dh0 + dw0, lt1, lt1 + dw1, lt1 + dh1, lt1 + dh1 + dw1};
                if (VAR == ADDTID) {
                    // the chunk is one tile row of 64 consecutive windows: lane l reads dword (row base + corner) / 4 + l
                    const uint32_t rb = __builtin_amdgcn_readfirstlane(2u * (wib * 4u + (uint32_t)c) * PITCH * 4u);
                    asm volatile(
                        "s_add_u32 m0, %8, %9\n s_nop 0\n ds_read_addtid_b32 %0\n"
                        "s_add_u32 m0, %8, %10\n s_nop 0\n ds_read_addtid_b32 %1\n"
                        "s_add_u32 m0, %8, %11\n s_nop 0\n ds_read_addtid_b32 %2\n"
                        "s_add_u32 m0, %8, %12\n s_nop 0\n ds_read_addtid_b32 %3\n"
                        "s_add_u32 m0, %8, %13\n s_nop 0\n ds_read_addtid_b32 %4\n"
                        "s_add_u32 m0, %8, %14\n s_nop 0\n ds_read_addtid_b32 %5\n"
                        "s_add_u32 m0, %8, %15\n s_nop 0\n ds_read_addtid_b32 %6\n"
                        "s_add_u32 m0, %8, %16\n s_nop 0\n ds_read_addtid_b32 %7\n"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(v[c][0]), "=&v"(v[c][1]), "=&v"(v[c][2]), "=&v"(v[c][3]), "=&v"(v[c][4]), "=&v"(v[c][5]), "=&v"(v[c][6]), "=&v"(v[c][7])
                        : "s"(rb), "s"(a[0]), "s"(a[1]), "s"(a[2]), "s"(a[3]), "s"(a[4]), "s"(a[5]), "s"(a[6]), "s"(a[7])
                        : "m0", "memory");
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (VAR == NOLDS) v[c][k] = off[c] ^ a[k];
                    else if (VAR == NOADD) v[c][k] = *reinterpret_cast<const uint32_t*>(base + off[c] + 64 * k + (lt0 & 0u));   // immediate offsets only
                    else v[c][k] = *reinterpret_cast<const uint32_t*>(base + (off[c] + a[k]));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (VAR == NOALU) {
                    acc[c] += __uint_as_float((v[c][0] ^ v[c][1] ^ v[c][2] ^ v[c][3] ^ v[c][4] ^ v[c][5] ^ v[c][6] ^ v[c][7]) & 0x3fffffffu);
                } else {
                    const uint32_t r0 = v[c][0] - v[c][1] - v[c][2] + v[c][3];
                    const uint32_t r1 = v[c][4] - v[c][5] - v[c][6] + v[c][7];
                    float s = (float)r0 * w0;
                    s += (float)r1 * w1;
                    acc[c] += (s >= thr * var[c]) ? right : left;
                }
            }
            r = rn;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const uint32_t n_recs = 64;
    std::vector<uint32_t> recs(n_recs * 16);
    uint32_t h = 12345;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return h >> 8; };
    for (uint32_t j = 0; j < n_recs; ++j) {
        uint32_t* r = &recs[j * 16];
        for (int q = 0; q < 2; ++q) {
            const uint32_t rx = rnd() % 12, ry = rnd() % 12, rw = 2 + rnd() % 8, rh = 2 + rnd() % 8;
            r[q] = (ry * PITCH + (rx >> 1)) * 4u;          // lt (de-interleaved column)
            r[3 + q] = rh * PITCH * 4u;                    // dh
            if (q == 0) r[6] = ((rw >> 1) + 1) * 4u; else r[6] |= (((rw >> 1) + 1) * 4u) << 16;
        }
        const float w0 = -1.0f / 324, w1 = 2.0f / 324, thr = 0.01f, l = 0.3f, rr = -0.2f;
        memcpy(&r[8], &w0, 4); memcpy(&r[9], &w1, 4); memcpy(&r[11], &thr, 4); memcpy(&r[12], &l, 4); memcpy(&r[13], &rr, 4);
    }
    // pattern 3: host-side binning of survivors (one table per keep rate, reused by every workgroup)
    auto make_binned = [&](uint32_t keep_pm) {
        std::vector<uint32_t> out(2048, 0);
        for (uint32_t w = 0; w < 8; ++w) {
            // survivors of a sparse tile, taken in order until the wave's 256 slots are full
            std::vector<uint32_t> surv;
            uint32_t idx = w * 977u, hh = 99u + w;
            while (surv.size() < 256) {
                hh = hh * 1664525u + 1013904223u;
                if ((hh >> 8) % 1000u < keep_pm) surv.push_back(idx % 2048u);
                ++idx;
            }
            std::vector<std::vector<uint32_t>> by_res(32);
            for (uint32_t t : surv) by_res[(2u * (t / 64u) * PITCH + (t % 64u)) % 32u].push_back(t);
            uint32_t q = 0;
            std::vector<std::vector<uint32_t>> groups(8);
            for (auto& v : by_res) for (uint32_t t : v) groups[(q++) % 8].push_back(t);
            for (uint32_t g = 0; g < 8; ++g)
                for (uint32_t i = 0; i < 32; ++i) out[w * 256 + g * 32 + i] = groups[g][i % groups[g].size()];
        }
        return out;
    };
    uint32_t* d_binned;
    hipMalloc(&d_binned, 2048 * 4);
    Rec* d_tab;
    float* d_out;
    hipMalloc(&d_tab, recs.size() * 4);
    hipMalloc(&d_out, 64);
    hipMemcpy(d_tab, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds_bytes = (size_t)TILE_DW * 4 + 16384;   // image + what the queues take in the real kernel
    const void* fns[] = {(const void*)stump_loop<FULL>, (const void*)stump_loop<NOADD>, (const void*)stump_loop<NOLDS>, (const void*)stump_loop<NOALU>,
                         (const void*)stump_loop<ADDTID>};
    for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Pat { const char* name; uint32_t pattern, keep; };
    const Pat pats[] = {{"dense rows (stage 0)", 0, 0}, {"survivors p=0.67 in order", 1, 670}, {"survivors p=0.30 in order", 1, 300},
                        {"survivors p=0.10 in order", 1, 100}, {"unrelated windows", 2, 0}, {"survivors p=0.30 binned", 3, 300},
                        {"survivors p=0.10 binned", 3, 100}, {"survivors p=0.03 binned", 3, 30}};
    const char* vnames[] = {"full", "no address adds", "no LDS reads", "no arithmetic", "addtid reads"};
    for (int wg_per_cu : {1, 2}) {
        for (const Pat& pt : pats) {
            for (int var = 0; var < 5; ++var) {
                if (var == 4 && pt.pattern != 0) continue;   // addtid reads need consecutive windows in the lanes
                const uint32_t iters = 40;
                const int blocks = cus * wg_per_cu;
                if (pt.pattern == 3) {
                    const std::vector<uint32_t> b = make_binned(pt.keep);
                    hipMemcpy(d_binned, b.data(), b.size() * 4, hipMemcpyHostToDevice);
                }
                auto run = [&]() {
                    switch (var) {
                        case 0: hipLaunchKernelGGL(stump_loop<FULL>, dim3(blocks), dim3(512), lds_bytes, 0, d_tab, n_recs, iters, pt.pattern, pt.keep, d_out, d_binned); break;
                        case 1: hipLaunchKernelGGL(stump_loop<NOADD>, dim3(blocks), dim3(512), lds_bytes, 0, d_tab, n_recs, iters, pt.pattern, pt.keep, d_out, d_binned); break;
                        case 2: hipLaunchKernelGGL(stump_loop<NOLDS>, dim3(blocks), dim3(512), lds_bytes, 0, d_tab, n_recs, iters, pt.pattern, pt.keep, d_out, d_binned); break;
                        case 3: hipLaunchKernelGGL(stump_loop<NOALU>, dim3(blocks), dim3(512), lds_bytes, 0, d_tab, n_recs, iters, pt.pattern, pt.keep, d_out, d_binned); break;
                        default: hipLaunchKernelGGL(stump_loop<ADDTID>, dim3(blocks), dim3(512), lds_bytes, 0, d_tab, n_recs, iters, pt.pattern, pt.keep, d_out, d_binned); break;
                    }
                };
                run();
                hipEventRecord(e0);
                run();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                // wave-stumps (64 windows x 1 stump) per CU
                const double ws = (double)wg_per_cu * 8 * 4 * iters * n_recs;
                const double gev = (double)blocks * 512 * 4 * iters * n_recs / (ms * 1e-3) / 1e9;
                printf("%d WG/CU  %-28s %-18s %7.3f ms  %6.1f CU-cycles per wave-stump (@2.1 GHz)  %7.1f G stump-evals/s\n", wg_per_cu, pt.name,
                       vnames[var], ms, ms * 1e-3 * 2.1e9 / ws, gev);
            }
        }
    }
    return 0;
}
