// Host-side helpers shared by the C-ABI translation units: thread-local error text and launch checking.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "egoprior.h"

namespace ep_host {

char* error_buffer();  // thread-local, 512 bytes (defined in egoprior_capi.hip)

inline int fail(int code, const char* msg) {
    std::snprintf(error_buffer(), 512, "%s", msg);
    return code;
}

inline int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(error_buffer(), 512, "%s: %s", what, hipGetErrorString(e));
        return EGOPRIOR_E_HIP;
    }
    return EGOPRIOR_OK;
}

}  // namespace ep_host
