// clif.h's seven functions on top of libvjhip.so — what a maintainer of the reference links instead of clif.cpp +
// OpenCL + CLUtil.  Call sites (main.cpp:54, 59-69; clod.cpp:360-369) stay as they are.
//
// use_opencl: the reference's CL_FALSE branches are the OpenCV host calls (cvCvtColor, cvIntegral: clif.cpp:247-251,
// 281-285, 326-335) and its CL_TRUE branches the clif.cl kernels.  Here both run on the device — there is no CPU path —
// and return the values the reference's CL_FALSE branch defines (the one its detector uses, clod.cpp:366): BGR2GRAY in
// OpenCV's 8-bit fixed point, cvIntegral's exact integers.  The one thing the flag still selects is the element type
// of the squared image, which differs between the reference's branches: doubles from cvIntegral (CL_FALSE; read as
// ->data.db at clod.cpp:837), 64-bit unsigned integers from integralImageSumCols (CL_TRUE; main.cpp:69 reads them
// through (unsigned long*)).  Results are cvCreateMat / cvCreateImage allocations: the caller releases them.
#include "clif_hip.h"
#include "vj.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct CLIFEnvironmentData {
    vj_env* env = nullptr;
};

[[noreturn]] static void die(const char* what, int rc) {   // the reference exits inside clCheckOrExit (clif.cpp:135...)
    fprintf(stderr, "%s: %s (%s)\n", what, vj_strerror(rc), vj_last_error());
    exit(1);
}

vj_env* clifHipEnv(CLIFEnvironmentData* d) { return d ? d->env : nullptr; }

CLIFEnvironmentData* clifInitEnvironment(const cl_uint device_index) {                          // clif.cpp:79-104
    auto* d = new CLIFEnvironmentData();
    const int rc = vj_env_create((int)device_index, &d->env);
    if (rc) die("clifInitEnvironment", rc);
    return d;
}

void clifReleaseEnvironment(CLIFEnvironmentData* d) {                                           // clif.cpp:230-238
    if (!d) return;
    vj_env_destroy(d->env);
    delete d;
}

void clifInitBuffers(CLIFEnvironmentData* d, const cl_uint image_width, const cl_uint image_height, const cl_uint /*image_stride*/,
                     const cl_uint /*image_channels*/) {                                        // clif.cpp:106-190
    const int rc = vj_env_reserve(d->env, (int)image_width, (int)image_height, 1);
    if (rc) die("clifInitBuffers", rc);
}

void clifReleaseBuffers(CLIFEnvironmentData*) {}                                                // clif.cpp:192-228: owned by the environment

static vj_image as_image(const IplImage* s) {
    vj_image f = {(const uint8_t*)s->imageData, s->width, s->height, s->widthStep, 0, s->nChannels};
    return f;
}

CLIFGrayscaleResult clifGrayscale(const IplImage* source, CLIFEnvironmentData* d, const cl_bool /*use_opencl*/) {   // clif.cpp:241-271
    CLIFGrayscaleResult ret;
    ret.image = cvCreateImage(cvSize(source->width, source->height), IPL_DEPTH_8U, 1);
    const vj_image f = as_image(source);
    const int rc = vj_grayscale(d->env, &f, (uint8_t*)ret.image->imageData, ret.image->widthStep);
    if (rc) die("clifGrayscale", rc);
    return ret;
}

static CLIFIntegralResult integral_of(const char* who, const IplImage* source, CLIFEnvironmentData* d, const cl_bool use_opencl) {
    const int rows = source->height + 1, cols = source->width + 1;
    CLIFIntegralResult ret;
    ret.image = cvCreateMat(rows, cols, CV_32SC1);                 // clif.cpp:282-283
    ret.square_image = cvCreateMat(rows, cols, CV_64FC1);
    // the library writes tight rows; OpenCV may pad a matrix row (step), so go through a tight buffer when it does
    const bool tight = ret.image->step == cols * 4 && ret.square_image->step == cols * 8;
    std::vector<uint32_t> s_tmp;
    std::vector<uint64_t> q_tmp;
    uint32_t* s = (uint32_t*)ret.image->data.i;
    uint64_t* q = (uint64_t*)ret.square_image->data.db;
    if (!tight) {
        s_tmp.resize((size_t)rows * cols);
        q_tmp.resize((size_t)rows * cols);
        s = s_tmp.data();
        q = q_tmp.data();
    }
    const vj_image f = as_image(source);
    const int rc = vj_integral_image(d->env, &f, s, q);            // 1 channel: cvIntegral; 3 / 4: BGR2GRAY first
    if (rc) die(who, rc);
    for (int y = 0; y < rows; ++y) {
        uint64_t* qrow = (uint64_t*)(ret.square_image->data.ptr + (size_t)y * ret.square_image->step);
        if (!tight) {
            memcpy(ret.image->data.ptr + (size_t)y * ret.image->step, s + (size_t)y * cols, (size_t)cols * 4);
            memcpy(qrow, q + (size_t)y * cols, (size_t)cols * 8);
        }
        if (!use_opencl)                                           // cvIntegral's CV_64F: exact below 2^53
            for (int x = 0; x < cols; ++x) {
                const double v = (double)qrow[x];
                memcpy(&qrow[x], &v, 8);
            }
    }
    return ret;
}

// clif.cpp:273-316.  The reference hands `source` to cvIntegral as it is (single channel); a colour image is converted
// first, like clifGrayscaleIntegral (main.cpp:68 passes the BGR frame).
CLIFIntegralResult clifIntegral(const IplImage* source, CLIFEnvironmentData* d, const cl_bool use_opencl) {
    return integral_of("clifIntegral", source, d, use_opencl);
}

// clif.cpp:319-374: cvCvtColor(BGR2GRAY) + cvIntegral — fused here: the integral kernels convert every pixel as they read it
CLIFIntegralResult clifGrayscaleIntegral(const IplImage* source, CLIFEnvironmentData* d, const cl_bool use_opencl) {
    return integral_of("clifGrayscaleIntegral", source, d, use_opencl);
}
