// d3d_common.h -- launcher helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dynam3d_hip.h"

extern "C" void d3d_set_error_(const char* msg);

#define D3D_HIP(call)                                   \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) {                        \
            d3d_set_error_(hipGetErrorString(e__));     \
            return D3D_EHIP;                            \
        }                                               \
    } while (0)

// Launch errors only (configuration / resources); never synchronises the stream.
#define D3D_LAUNCH_CHECK()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) {                        \
            d3d_set_error_(hipGetErrorString(e__));     \
            return D3D_EHIP;                            \
        }                                               \
        return D3D_OK;                                  \
    } while (0)
