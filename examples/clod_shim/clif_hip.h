/* clif.h's interface (clif.h:33-73) on top of libvjhip.so: same names, same argument lists.  See clif_hip.cpp. */
#ifndef CLIF_HIP_H_
#define CLIF_HIP_H_
#ifdef CLOD_HIP_WITH_OPENCV
#include <opencv2/imgproc/imgproc.hpp>
typedef unsigned int cl_uint;
typedef unsigned int cl_bool;
#define CL_TRUE 1
#define CL_FALSE 0
#else
#include "cv_compat_min.h"
#endif

typedef struct CLIFEnvironmentData CLIFEnvironmentData;   /* the reference's holds cl_mem buffers (clif.h:12-31); opaque here */
typedef struct CLIFIntegralResult { CvMat* image; CvMat* square_image; } CLIFIntegralResult;      /* clif.h:33-36 */
typedef struct CLIFGrayscaleResult { IplImage* image; } CLIFGrayscaleResult;                      /* clif.h:38-40 */

CLIFEnvironmentData* clifInitEnvironment(const cl_uint device_index);                             /* clif.h:43-44 */
void clifReleaseEnvironment(CLIFEnvironmentData* data);                                           /* clif.h:46-47 */
void clifInitBuffers(CLIFEnvironmentData* data, const cl_uint image_width, const cl_uint image_height,
                     const cl_uint image_stride, const cl_uint image_channels);                   /* clif.h:49-54 */
void clifReleaseBuffers(CLIFEnvironmentData* data);                                               /* clif.h:56-57 */
CLIFGrayscaleResult clifGrayscale(const IplImage* source, CLIFEnvironmentData* data, const cl_bool use_opencl);          /* clif.h:60-63 */
CLIFIntegralResult clifIntegral(const IplImage* source, CLIFEnvironmentData* data, const cl_bool use_opencl);            /* clif.h:65-68 */
CLIFIntegralResult clifGrayscaleIntegral(const IplImage* source, CLIFEnvironmentData* data, const cl_bool use_opencl);   /* clif.h:70-73 */

/* not in clif.h: the library environment behind a clif environment (clod_hip.cpp runs its detector on the same one) */
struct vj_env* clifHipEnv(CLIFEnvironmentData* data);
#endif
