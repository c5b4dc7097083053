/* The handful of OpenCV 2.4 C types that clod.h's interface mentions (clod.h:61-81, main.cpp:27-57), as plain PODs —
 * ONLY so that the shim in this directory can be compiled and exercised in an image without OpenCV.  A maintainer of
 * the reference includes the real <opencv2/...> headers instead; field names and order follow the public OpenCV C API
 * (the reference keeps a copy of the Haar structs in tempcv.hpp:70-112).                                            */
#ifndef CV_COMPAT_MIN_H_
#define CV_COMPAT_MIN_H_
#include <stdint.h>

typedef struct CvSize { int width, height; } CvSize;
typedef struct CvRect { int x, y, width, height; } CvRect;
static inline CvSize cvSize(int w, int h) { CvSize s = {w, h}; return s; }
static inline CvRect cvRect(int x, int y, int w, int h) { CvRect r = {x, y, w, h}; return r; }

typedef struct IplImage {      /* the fields the detect path reads (clif.cpp:326-335, clod.cpp:360-369) */
    int nChannels, depth, width, height, widthStep;
    char* imageData;
} IplImage;
#define IPL_DEPTH_8U 8

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
