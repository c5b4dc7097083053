/* The handful of OpenCV 2.4 C types that clod.h's interface mentions (clod.h:61-81, main.cpp:27-57), as plain PODs —
 * ONLY so that the shim in this directory can be compiled and exercised in an image without OpenCV.  A maintainer of
 * the reference includes the real <opencv2/...> headers instead; field names and order follow the public OpenCV C API
 * (the reference keeps a copy of the Haar structs in tempcv.hpp:70-112).                                            */
#ifndef CV_COMPAT_MIN_H_
#define CV_COMPAT_MIN_H_
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct CvSize { int width, height; } CvSize;
typedef struct CvRect { int x, y, width, height; } CvRect;
static inline CvSize cvSize(int w, int h) { CvSize s = {w, h}; return s; }
static inline CvRect cvRect(int x, int y, int w, int h) { CvRect r = {x, y, w, h}; return r; }

typedef struct IplImage {      /* the fields the detect path reads (clif.cpp:326-335, clod.cpp:360-369) */
    int nChannels, depth, width, height, widthStep;
    char* imageData;
} IplImage;
#define IPL_DEPTH_8U 8

/* CvMat as clif.h's results use it (CLIFIntegralResult, clif.h:33-36; read through ->data.i / ->data.db / ->width at
 * clod.cpp:426-433, 836-837): type, row step in bytes, the data union, rows / cols.  cvCreateMat / cvReleaseMat /
 * cvCreateImage / cvReleaseImage are here only so that the shim and its demo run without OpenCV (tight rows).      */
typedef struct CvMat {
    int type, step;
    int* refcount;
    int hdr_refcount;
    union { unsigned char* ptr; short* s; int* i; float* fl; double* db; } data;
    union { int rows; int height; };
    union { int cols; int width; };
} CvMat;
#define CV_32SC1 4
#define CV_64FC1 6
static inline CvMat* cvCreateMat(int rows, int cols, int type) {
    const int elem = type == CV_64FC1 ? 8 : 4;
    CvMat* m = (CvMat*)calloc(1, sizeof(CvMat));
    m->type = type; m->rows = rows; m->cols = cols; m->step = cols * elem;
    m->data.ptr = (unsigned char*)calloc((size_t)rows * (size_t)cols, (size_t)elem);
    return m;
}
static inline void cvReleaseMat(CvMat** m) { if (m && *m) { free((*m)->data.ptr); free(*m); *m = NULL; } }
static inline IplImage* cvCreateImage(CvSize size, int depth, int channels) {
    IplImage* im = (IplImage*)calloc(1, sizeof(IplImage));
    im->nChannels = channels; im->depth = depth; im->width = size.width; im->height = size.height;
    im->widthStep = (size.width * channels + 3) & ~3;        /* OpenCV aligns rows to 4 bytes */
    im->imageData = (char*)calloc((size_t)im->widthStep, (size_t)size.height);
    return im;
}
static inline void cvReleaseImage(IplImage** im) { if (im && *im) { free((*im)->imageData); free(*im); *im = NULL; } }

#define CV_HAAR_FEATURE_MAX 3
typedef struct CvHaarFeature {
    int tilted;
    struct { CvRect r; float weight; } rect[CV_HAAR_FEATURE_MAX];
} CvHaarFeature;
typedef struct CvHaarClassifier {
    int count;
    CvHaarFeature* haar_feature;
    float* threshold;
    int* left;
    int* right;
    float* alpha;
} CvHaarClassifier;
typedef struct CvHaarStageClassifier {
    int count;
    float threshold;
    CvHaarClassifier* classifier;
    int next, child, parent;
} CvHaarStageClassifier;
typedef struct CvHaarClassifierCascade {
    int flags, count;
    CvSize orig_window_size, real_window_size;
    double scale;
    CvHaarStageClassifier* stage_classifier;
    void* hid_cascade;
} CvHaarClassifierCascade;

typedef unsigned int cl_uint;
typedef unsigned int cl_bool;
#define CL_TRUE 1
#define CL_FALSE 0
#endif
