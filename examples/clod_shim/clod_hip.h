/* clod.h's interface (clod.h:17-21, 39-47, 61-81) on top of libvjhip.so: same names, same argument lists, same
 * ownership rules (the caller free()s result.matches, main.cpp:183).  See clod_hip.cpp.                              */
#ifndef CLOD_HIP_H_
#define CLOD_HIP_H_
#ifdef CLOD_HIP_WITH_OPENCV
#include <opencv2/imgproc/imgproc.hpp>
#include <opencv2/objdetect/objdetect.hpp>
#else
#include "cv_compat_min.h"
#endif
#include "clif_hip.h"

typedef unsigned int clod_flags;
#define CLOD_PRECOMPUTE_FEATURES (2 << 0)
#define CLOD_BLOCK_IMPLEMENTATION (2 << 1)
#define CLOD_PER_STAGE_ITERATIONS (2 << 2)

typedef struct CLODWeightedRect { CvRect rect; float weight; } CLODWeightedRect;
typedef struct CLODDetectObjectsResult { CLODWeightedRect* matches; cl_uint match_count; } CLODDetectObjectsResult;
typedef struct CLODFEnvironmentData {      /* clod.h:55-59: callers reach the clif environment through ->clif (main.cpp:54) */
    CLIFEnvironmentData* clif;
    struct CLODHipState* hip;              /* in place of the reference's CLDeviceEnvironment + cl_mem buffers */
} CLODEnvironmentData;

CLODEnvironmentData* clodInitEnvironment(const cl_uint device_index);
void clodReleaseEnvironment(CLODFEnvironmentData* data);
void clodInitBuffers(CLODEnvironmentData* data, const CvSize* image_size);
void clodReleaseBuffers(CLODEnvironmentData* data);
CLODDetectObjectsResult clodDetectObjects(const IplImage* image, const CvHaarClassifierCascade* cascade,
                                          const CLODEnvironmentData* data, const CvSize min_window_size,
                                          const CvSize max_window_size, const cl_uint min_neighbors, const clod_flags flags,
                                          const cl_bool use_opencl);
#endif
