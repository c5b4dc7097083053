// The reference demo's call pattern (main.cpp:19-136, 159-184) against the shim: a CvHaarClassifierCascade in memory —
// built here from a shipped .vjc because cvLoad needs OpenCV — a 640x480 frame, clodInitEnvironment / clodInitBuffers /
// clodDetectObjects(..., CL_TRUE) and the two CPU-variant window sets, free().  Prints the match counts; exits 1 when the
// known answer of the survey's pin frame (2 raw detections) is missed.
#include "clod_hip.h"
#include "vj.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// what cvLoad(xml) would hand the caller (main.cpp:36)
struct OwnedCascade {
    CvHaarClassifierCascade c{};
    std::vector<CvHaarStageClassifier> stages;
    std::vector<CvHaarClassifier> classifiers;
    std::vector<CvHaarFeature> features;
    std::vector<float> thresholds, alpha;
    std::vector<int> left, right;
};

static bool load(const char* path, OwnedCascade* o) {
    vj_cascade* vc = nullptr;
    if (vj_cascade_load(path, &vc) != VJ_OK) return false;
    vj_cascade_info info;
    vj_cascade_get_info(vc, &info);
    const vj_stage_desc* st = vj_cascade_stages(vc);
    const vj_tree_desc* tr = vj_cascade_trees(vc);
    const vj_node_desc* nd = vj_cascade_nodes(vc);
    const float* al = vj_cascade_alpha(vc);
    o->stages.resize(info.n_stages);
    o->classifiers.resize(info.n_trees);
    o->features.resize(info.n_nodes);
    o->thresholds.resize(info.n_nodes);
    o->left.resize(info.n_nodes);
    o->right.resize(info.n_nodes);
    o->alpha.assign(al, al + info.n_alpha);
    for (int n = 0; n < info.n_nodes; ++n) {
        o->features[n].tilted = nd[n].tilted;
        for (int k = 0; k < 3; ++k) {
            o->features[n].rect[k].r = cvRect(nd[n].rect[k].x, nd[n].rect[k].y, nd[n].rect[k].w, nd[n].rect[k].h);
            o->features[n].rect[k].weight = nd[n].rect[k].weight;
        }
        o->thresholds[n] = nd[n].threshold;
        o->left[n] = nd[n].left;
        o->right[n] = nd[n].right;
    }
    for (int t = 0; t < info.n_trees; ++t)
        o->classifiers[t] = CvHaarClassifier{tr[t].n_nodes, &o->features[tr[t].first_node], &o->thresholds[tr[t].first_node],
                                             &o->left[tr[t].first_node], &o->right[tr[t].first_node], &o->alpha[tr[t].first_alpha]};
    for (int s = 0; s < info.n_stages; ++s)
        o->stages[s] = CvHaarStageClassifier{st[s].n_trees, st[s].threshold, &o->classifiers[st[s].first_tree], st[s].next, st[s].child,
                                             st[s].parent};
    o->c.count = info.n_stages;
    o->c.orig_window_size = cvSize(info.win_w, info.win_h);
    o->c.stage_classifier = o->stages.data();
    vj_cascade_free(vc);
    return true;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "clfacedetection_amd/data/haarcascade_frontalface_alt.vjc";
    OwnedCascade casc;
    if (!load(path, &casc)) { fprintf(stderr, "cannot load %s\n", path); return 1; }
    const int W = 640, H = 480;
    std::vector<unsigned char> px((size_t)W * H);
    unsigned s = 12345u;
    for (auto& v : px) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; v = (unsigned char)(s & 0xffu); }
    IplImage img{1, IPL_DEPTH_8U, W, H, W, (char*)px.data()};

    CLODEnvironmentData* env = clodInitEnvironment(0);          // main.cpp:53
    CvSize size = cvSize(W, H);
    clodInitBuffers(env, &size);                                 // main.cpp:55
    int ok = 1;
    for (int variant = 0; variant < 3; ++variant) {              // main.cpp:72-97 runs the variants one after the other
        const clod_flags flags = CLOD_PRECOMPUTE_FEATURES | (variant == 2 ? CLOD_PER_STAGE_ITERATIONS : 0);
        CLODDetectObjectsResult r = clodDetectObjects(&img, &casc.c, env, cvSize(0, 0), cvSize(0, 0), 0, flags, variant == 0 ? CL_TRUE : CL_FALSE);
        printf("variant %d: %u matches", variant, r.match_count);
        for (cl_uint i = 0; i < r.match_count; ++i)
            printf(" [%d %d %d %d]", r.matches[i].rect.x, r.matches[i].rect.y, r.matches[i].rect.width, r.matches[i].rect.height);
        printf("\n");
        ok = ok && r.match_count == 2;   // the survey's probe: the OpenCL route and every CPU variant return the same 2 rectangles
        free(r.matches);                 // main.cpp:183
    }
    clodReleaseBuffers(env);
    clodReleaseEnvironment(env);
    printf(ok ? "clod shim demo: OK\n" : "clod shim demo: MISMATCH\n");
    return ok ? 0 : 1;
}
