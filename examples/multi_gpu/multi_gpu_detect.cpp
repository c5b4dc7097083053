// A C++ host driving every GPU of the node (BASELINE config 3's shape: a batch of frames sharded over the devices, one
// RCCL all-gather of the detection rectangles): one thread and one vj_env per device — the reference's own pattern is
// clodInitEnvironment(device_index) (clod.cpp:72-100) — with vj_shard_frames for the split and include/vj_rccl.h for the
// gather.  The result is checked against a single-device run of the whole batch.
//   hipcc -O2 -std=c++17 -Iinclude examples/multi_gpu/multi_gpu_detect.cpp -Lclfacedetection_amd -lvjhip -lrccl \
//         -Wl,-rpath,$PWD/clfacedetection_amd -o multi_gpu_detect && ./multi_gpu_detect [n_devices] [n_frames]
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "vj.h"
#include "vj_rccl.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #x, vj_last_error()); exit(1); } } while (0)

int main(int argc, char** argv) {
    int n_dev = 0;
    CHECK(hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0);
    if (argc > 1) n_dev = std::min(n_dev, atoi(argv[1]));
    const int n_frames = argc > 2 ? atoi(argv[2]) : 6, W = 640, H = 360;
    vj_cascade* casc = nullptr;
    CHECK(vj_cascade_load("clfacedetection_amd/data/haarcascade_frontalface_alt.vjc", &casc) == VJ_OK);
    // frames: xorshift noise, one seed per frame
    std::vector<uint8_t> px((size_t)n_frames * W * H);
    for (int f = 0; f < n_frames; ++f) {
        uint32_t s = 1000u + (uint32_t)f;
        for (size_t i = 0; i < (size_t)W * H; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; px[(size_t)f * W * H + i] = (uint8_t)s; }
    }
    std::vector<vj_image> frames((size_t)n_frames);
    for (int f = 0; f < n_frames; ++f) frames[(size_t)f] = vj_image{px.data() + (size_t)f * W * H, W, H, W, 0, 1};
    vj_params p;
    vj_params_default(&p);

    std::vector<int> devs((size_t)n_dev);
    for (int d = 0; d < n_dev; ++d) devs[(size_t)d] = d;
    std::vector<ncclComm_t> comms((size_t)n_dev);
    CHECK(ncclCommInitAll(comms.data(), n_dev, devs.data()) == ncclSuccess);

    std::vector<std::vector<vj_rect>> gathered((size_t)n_dev);
    int collectives[3] = {0, 0, 0};
    float gather_ms[3] = {0, 0, 0};
    std::vector<std::thread> th;
    for (int rank = 0; rank < n_dev; ++rank)
        th.emplace_back([&, rank]() {
            CHECK(hipSetDevice(rank) == hipSuccess);
            vj_env* env = nullptr;
            CHECK(vj_env_create(rank, &env) == VJ_OK);
            int first = 0, count = 0;
            CHECK(vj_shard_frames(n_frames, n_dev, rank, &first, &count) == VJ_OK);
            vj_result r;
            memset(&r, 0, sizeof(r));
            if (count) CHECK(vj_detect(env, casc, frames.data() + first, count, &p, &r) == VJ_OK);
            for (uint32_t i = 0; i < r.count; ++i) r.rects[i].frame += first;   // global frame numbers
            hipStream_t st;
            CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
            // the gatherer lives across steps (a service would keep it next to its vj_env): ONE collective per step.  It starts
            // with room for 4 rectangles per rank here so that the first step also shows the regrow path (every rank reads the
            // same counts in the gathered headers and repeats the collective with the same larger blocks).
            vj_rccl_gatherer g;
            CHECK(vj_rccl_gatherer_init(&g, comms[(size_t)rank], st, 4) == VJ_OK);
            CHECK(g.n_ranks == n_dev);
            for (int step = 0; step < 3; ++step) {
                vj_rect* all = nullptr;
                uint32_t n_all = 0;
                const uint64_t before = g.n_collectives;
                CHECK(vj_rccl_gatherer_run(&g, r.rects, r.count, &all, &n_all) == VJ_OK);
                if (step > 0) CHECK(g.n_collectives == before + 1);      // steady state: one ncclAllGather per step
                if (rank == 0) collectives[step] = (int)(g.n_collectives - before), gather_ms[step] = g.last_ms;
                gathered[(size_t)rank].assign(all, all + n_all);
                free(all);
            }
            vj_rccl_gatherer_destroy(&g);
            vj_result_free(&r);
            (void)hipStreamDestroy(st);
            vj_env_destroy(env);
        });
    for (auto& t : th) t.join();
    int rccl_ranks = 0;   // what the communicator itself says (1 on a one-GPU box: RCCL does not take two ranks on one device)
    CHECK(ncclCommCount(comms[0], &rccl_ranks) == ncclSuccess);
    for (auto& c : comms) ncclCommDestroy(c);

    // reference: the whole batch on device 0
    CHECK(hipSetDevice(0) == hipSuccess);
    vj_env* env0 = nullptr;
    CHECK(vj_env_create(0, &env0) == VJ_OK);
    vj_result whole;
    CHECK(vj_detect(env0, casc, frames.data(), n_frames, &p, &whole) == VJ_OK);
    bool ok = true;
    for (int rank = 0; rank < n_dev; ++rank) {
        const auto& g = gathered[(size_t)rank];
        ok = ok && g.size() == whole.count;
        for (size_t i = 0; ok && i < g.size(); ++i)
            ok = g[i].x == whole.rects[i].x && g[i].y == whole.rects[i].y && g[i].w == whole.rects[i].w && g[i].frame == whole.rects[i].frame &&
                 g[i].scale_idx == whole.rects[i].scale_idx;
    }
    printf("multi_gpu_detect: %d device(s), rccl_ranks %d, %d frames, %u rectangles gathered on every rank: %s; collectives per step %d %d %d, "
           "gather %.3f %.3f %.3f ms\n", n_dev, rccl_ranks, n_frames, whole.count, ok ? "OK" : "MISMATCH", collectives[0], collectives[1],
           collectives[2], gather_ms[0], gather_ms[1], gather_ms[2]);
    vj_result_free(&whole);
    vj_env_destroy(env0);
    vj_cascade_free(casc);
    return ok ? 0 : 1;
}
