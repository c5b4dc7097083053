"""MI355X-native Viola–Jones detect path: a drop-in for the clif/clod detect path of
GabrieleCocco/CLFaceDetection behind the C ABI in include/vj.h (libvjhip.so).

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI + host planning),
api.py (ctypes mirror of the reference's clod*/clif* functions), data/ (stock cascades
in the compact .vjc form), synth.py (synthetic frames for tests and bench).
"""
from .api import (  # noqa: F401
    CLOD_BLOCK_IMPLEMENTATION, CLOD_PER_STAGE_ITERATIONS, CLOD_PRECOMPUTE_FEATURES,
    VJ_FLAG_COUNTERS, VJ_FLAG_SIGNED_MEAN, VJ_FLAG_GRID_F64, VJ_FLAG_TILTED_AS_UPRIGHT, VJ_FLAG_SKIP_LIST, VJ_FLAG_SKIP_ROW, Cascade, DetectResult, DeviceFrames, Environment,
    FrameStream, Params, VjError,
    clifIntegral, clodDetectObjects, clodInitBuffers, clodInitEnvironment, clodReleaseBuffers,
    clodReleaseEnvironment, cvHaarDetectObjects, default_params, group_rectangles, load_library,
)
