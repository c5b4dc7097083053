/* The C ABI from a plain C translation unit (gcc -std=c99): load a shipped cascade, run vj_detect and vj_integral on a
 * generated frame, check the known answers.  Built and run by tests/test_gpu_native.py on the GPU box:
 *   gcc -std=c99 -Wall -Wextra -Iinclude tests/capi_smoke.c -Lclfacedetection_amd -lvjhip -Wl,-rpath,$PWD/clfacedetection_amd
 * The frame is the survey's pin: 640x480 xorshift32(13,17,5) noise, seed 12345, on which the
 * reference's own kernel finds 2 raw detections with haarcascade_frontalface_alt (tests/test_oracle_pins.py).        */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vj.h"

static int fail(const char* what, int rc) {
    fprintf(stderr, "capi_smoke: %s failed: %s (%s)\n", what, vj_strerror(rc), vj_last_error());
    return 1;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "clfacedetection_amd/data/haarcascade_frontalface_alt.vjc";
    const int W = 640, H = 480;
    vj_cascade* casc = NULL;
    vj_env* env = NULL;
    int rc;
    if ((rc = vj_cascade_load(path, &casc))) return fail("vj_cascade_load", rc);
    vj_cascade_info info;
    if ((rc = vj_cascade_get_info(casc, &info))) return fail("vj_cascade_get_info", rc);
    if (info.win_w != 20 || info.n_stages != 22 || info.n_nodes != 2135) { fprintf(stderr, "capi_smoke: unexpected cascade\n"); return 1; }
    if ((rc = vj_env_create(0, &env))) return fail("vj_env_create", rc);
    if ((rc = vj_env_reserve(env, W, H, 1))) return fail("vj_env_reserve", rc);

    uint8_t* img = (uint8_t*)malloc((size_t)W * H);
    uint32_t s = 12345u;
    for (int i = 0; i < W * H; ++i) {   /* xorshift32 (13, 17, 5); the frame is the low byte of the state */
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        img[i] = (uint8_t)(s & 0xffu);
    }
    vj_image frame;
    memset(&frame, 0, sizeof(frame));
    frame.data = img; frame.width = W; frame.height = H; frame.stride = W; frame.on_device = 0; frame.channels = 1;
    vj_params p;
    vj_params_default(&p);
    p.flags = VJ_FLAG_COUNTERS;
    vj_result r;
    if ((rc = vj_detect(env, casc, &frame, 1, &p, &r))) return fail("vj_detect", rc);
    printf("capi_smoke: %u raw detections, %llu windows, %llu stump evaluations\n", r.count,
           (unsigned long long)r.counters.windows, (unsigned long long)r.counters.stump_evals);
    for (uint32_t i = 0; i < r.count; ++i)
        printf("  rect %u: x=%d y=%d w=%d h=%d scale=%d\n", i, r.rects[i].x, r.rects[i].y, r.rects[i].w, r.rects[i].h, r.rects[i].scale_idx);
    int ok = r.count == 2 && r.counters.windows == 839321ull;   /* 839,321 windows at 640x480 / alt */
    vj_result_free(&r);

    /* clifIntegral's contract on the same frame: last element = sum of all pixels, first row / column zero */
    uint32_t* sum = (uint32_t*)malloc((size_t)(W + 1) * (H + 1) * sizeof(uint32_t));
    uint64_t* sq = (uint64_t*)malloc((size_t)(W + 1) * (H + 1) * sizeof(uint64_t));
    if ((rc = vj_integral(env, img, W, H, W, sum, sq))) return fail("vj_integral", rc);
    uint64_t tot = 0, tot2 = 0;
    for (int i = 0; i < W * H; ++i) { tot += img[i]; tot2 += (uint64_t)img[i] * img[i]; }
    ok = ok && sum[(size_t)H * (W + 1) + W] == (uint32_t)tot && sq[(size_t)H * (W + 1) + W] == tot2 && sum[5] == 0 && sum[(size_t)7 * (W + 1)] == 0;

    /* errors are return codes, never exit() */
    ok = ok && vj_detect(env, casc, NULL, 1, &p, &r) == VJ_ERR_ARG && vj_cascade_load("/nonexistent.vjc", &casc) == VJ_ERR_IO;
    free(sum); free(sq); free(img);
    vj_env_destroy(env);
    printf(ok ? "capi_smoke: OK\n" : "capi_smoke: MISMATCH\n");
    return ok ? 0 : 1;
}
