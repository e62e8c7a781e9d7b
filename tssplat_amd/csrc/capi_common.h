// Shared by the translation units that implement the C ABI (capi.cpp, surface_capi.cpp).
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

#include "../../include/tssplat_amd.h"

namespace tsamd {

// Records `msg` as the calling thread's tsamd_last_error() text and returns `code`.
int capi_fail(int code, const std::string &msg);

// Makes `dev` current for the scope of a C-ABI call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool active = false;
    hipError_t enter(int dev)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev != dev) {
            e = hipSetDevice(dev);
            if (e != hipSuccess) return e;
            active = true;
        }
        return hipSuccess;
    }
    ~DeviceGuard()
    {
        if (active) (void)hipSetDevice(prev);
    }
};

}  // namespace tsamd

#define TSAMD_HIP(call)                                                                                         \
    do {                                                                                                        \
        hipError_t e__ = (call);                                                                                \
        if (e__ != hipSuccess)                                                                                  \
            return tsamd::capi_fail(TSAMD_ERR_HIP, std::string(#call) + " failed: " + hipGetErrorString(e__)); \
    } while (0)
