// clod.h's five functions on top of libvjhip.so — what a maintainer of the reference links instead of clod.cpp +
// OpenCL + CLUtil.  Call sites (main.cpp:53-57, 159-184) stay as they are: clodDetectObjects still takes the
// CvHaarClassifierCascade* that cvLoad returned (main.cpp:36); it is converted once to the library's flat arrays
// (vj_cascade_from_arrays) and cached by pointer.
//   g++ -std=c++17 -I../../include clod_hip.cpp clif_hip.cpp demo_main.cpp -L../../clfacedetection_amd -lvjhip
#include "clod_hip.h"
#include "vj.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct CLODHipState {
    vj_env* env = nullptr;                                             // the clif environment's: one device context for both
    std::map<const CvHaarClassifierCascade*, vj_cascade*> cascades;   // converted on first use
};

[[noreturn]] static void die(const char* what, int rc) {   // the reference exits inside clCheckOrExit (clod.cpp:114...)
    fprintf(stderr, "%s: %s (%s)\n", what, vj_strerror(rc), vj_last_error());
    exit(1);
}

CLODEnvironmentData* clodInitEnvironment(const cl_uint device_index) {                       // clod.h:61-62, clod.cpp:72-100
    // the reference creates its clif environment on device 0 whatever device_index says (clod.cpp:76) and a second
    // OpenCL context for the detector; here both halves share one environment on the device asked for
    auto* d = new CLODEnvironmentData();
    d->clif = clifInitEnvironment(device_index);
    d->hip = new CLODHipState();
    d->hip->env = clifHipEnv(d->clif);
    return d;
}

void clodReleaseEnvironment(CLODFEnvironmentData* d) {                                         // clod.h:64-65
    if (!d) return;
    for (auto& kv : d->hip->cascades) vj_cascade_free(kv.second);
    delete d->hip;
    clifReleaseEnvironment(d->clif);
    delete d;
}

void clodInitBuffers(CLODEnvironmentData* d, const CvSize* s) {                                // clod.h:67-69
    const int rc = vj_env_reserve(d->hip->env, s->width, s->height, 1);
    if (rc) die("clodInitBuffers", rc);
}

void clodReleaseBuffers(CLODEnvironmentData*) {}                                               // clod.h:70-71: owned by the environment

// CvHaarClassifierCascade (tempcv.hpp:70-112) -> the flat arrays of vj.h.  left / right keep OpenCV's convention
// (> 0: node index inside the tree, <= 0: -index into the tree's alpha[], tempcv.cpp:1994-1995).
static vj_cascade* convert(const CvHaarClassifierCascade* c) {
    std::vector<vj_stage_desc> stages;
    std::vector<vj_tree_desc> trees;
    std::vector<vj_node_desc> nodes;
    std::vector<float> alpha;
    for (int i = 0; i < c->count; ++i) {
        const CvHaarStageClassifier& st = c->stage_classifier[i];
        stages.push_back(vj_stage_desc{(int32_t)trees.size(), st.count, st.threshold, st.parent, st.next, st.child});
        for (int j = 0; j < st.count; ++j) {
            const CvHaarClassifier& cl = st.classifier[j];
            trees.push_back(vj_tree_desc{(int32_t)nodes.size(), cl.count, (int32_t)alpha.size()});
            for (int l = 0; l < cl.count; ++l) {
                vj_node_desc n{};
                const CvHaarFeature& f = cl.haar_feature[l];
                n.tilted = f.tilted;
                n.threshold = cl.threshold[l];
                n.left = cl.left[l];
                n.right = cl.right[l];
                for (int k = 0; k < CV_HAAR_FEATURE_MAX; ++k) {
                    n.rect[k] = vj_rect_desc{f.rect[k].r.x, f.rect[k].r.y, f.rect[k].r.width, f.rect[k].r.height, f.rect[k].weight};
                    if (f.rect[k].weight != 0.0f) n.n_rects = k + 1;
                }
                nodes.push_back(n);
            }
            alpha.insert(alpha.end(), cl.alpha, cl.alpha + cl.count + 1);
        }
    }
    vj_cascade* out = nullptr;
    const int rc = vj_cascade_from_arrays(c->orig_window_size.width, c->orig_window_size.height, stages.data(), (int)stages.size(),
                                          trees.data(), (int)trees.size(), nodes.data(), (int)nodes.size(), alpha.data(),
                                          (int)alpha.size(), &out);
    if (rc) die("clodDetectObjects: cascade conversion", rc);
    return out;
}

CLODDetectObjectsResult clodDetectObjects(const IplImage* image, const CvHaarClassifierCascade* cascade,
                                          const CLODEnvironmentData* data, const CvSize min_window_size,
                                          const CvSize max_window_size, const cl_uint min_neighbors, const clod_flags flags,
                                          const cl_bool use_opencl) {                            // clod.h:72-81, clod.cpp:1339-1356
    CLODHipState* d = data->hip;                        // the reference mutates its "const" environment too (clod.cpp:800)
    vj_cascade*& vc = d->cascades[cascade];
    if (!vc) vc = convert(cascade);
    // the reference converts BGR -> gray on the host (cvCvtColor, clif.cpp:328); here the interleaved image goes in as it
    // is and the integral kernels convert each pixel with the same 8-bit fixed-point formula
    vj_image f = {(const uint8_t*)image->imageData, image->width, image->height, image->widthStep, 0, image->nChannels};
    vj_params p;
    vj_params_default(&p);
    p.min_w = min_window_size.width;
    p.min_h = min_window_size.height;
    p.max_w = max_window_size.width;
    p.max_h = max_window_size.height;
    p.min_neighbors = min_neighbors;   // 0 in the demo (main.cpp:165); != 0: grouped as cv::groupRectangles does
    // at the reference's own signature a cascade with tilted features behaves as in the reference: precomputeFeatures never reads
    // the flag (clod.cpp:448-492), the rectangles count as upright ones (vj_detect itself refuses such a cascade without this)
    p.flags |= VJ_FLAG_TILTED_AS_UPRIGHT;
    if (!use_opencl) {                 // the CPU variants' window sets (clod.cpp:1358-1499), still computed on the device
        p.flags |= (flags & CLOD_PER_STAGE_ITERATIONS) ? VJ_FLAG_SKIP_LIST : VJ_FLAG_SKIP_ROW;
        if (flags & CLOD_BLOCK_IMPLEMENTATION) p.flags |= VJ_FLAG_GRID_F64;   // clodDetectObjectsBlock: `step` is a double (clod.cpp:862)
    }
    vj_result r;
    const int rc = vj_detect(d->env, vc, &f, 1, &p, &r);
    if (rc) die("clodDetectObjects", rc);
    CLODDetectObjectsResult out;
    out.match_count = r.count;
    out.matches = (CLODWeightedRect*)malloc((r.count ? r.count : 1) * sizeof(CLODWeightedRect));   // caller free()s (main.cpp:183)
    for (uint32_t i = 0; i < r.count; ++i) {
        out.matches[i].rect = cvRect(r.rects[i].x, r.rects[i].y, r.rects[i].w, r.rects[i].h);
        out.matches[i].weight = r.rects[i].weight;
    }
    vj_result_free(&r);
    return out;
}
