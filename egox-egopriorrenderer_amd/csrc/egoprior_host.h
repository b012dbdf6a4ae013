// Host-side helpers shared by the C-ABI translation units: thread-local error text and launch checking.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

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

// ---- tuning / test options (egoprior_set_option); unknown names read as -1 -----------------------------------
int option(const char* name);

// ---- optional per-kernel timing (egoprior_profile_begin / _end): HIP events recorded on the launch stream ----
struct ProfRec {
    const char* name;
    hipEvent_t start, stop;
};
struct Prof {
    bool on = false;
    std::vector<ProfRec> recs;
};
Prof& prof();  // thread-local (defined in egoprior_capi.hip)

struct ScopedKernel {
    hipStream_t s;
    bool on;
    ProfRec r;
    ScopedKernel(const char* name, hipStream_t stream) : s(stream), on(prof().on && prof().recs.size() < 65536) {
        if (on) {
            r.name = name;
            on = hipEventCreate(&r.start) == hipSuccess && hipEventCreate(&r.stop) == hipSuccess;
            if (on) (void)hipEventRecord(r.start, s);
        }
    }
    ~ScopedKernel() {
        if (on) {
            (void)hipEventRecord(r.stop, s);
            prof().recs.push_back(r);
        }
    }
};

}  // namespace ep_host

// launch + (when profiling is on) a HIP-event pair around this one kernel
#define EP_LAUNCH(name, kernel, grid, block, shmem, stream, ...)               \
    do {                                                                       \
        ep_host::ScopedKernel _sk(name, stream);                               \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);   \
    } while (0)
