// Helpers shared by the two halves of the C ABI (capi.cpp: decoder, capi_encoder.cpp: encoder).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace pamd {

// records msg as the calling thread's pcd_last_error() text and returns code
int set_error(int code, const std::string& msg);

// Every entry point that touches the GPU runs on its handle's device and leaves the calling thread's current HIP
// device as it found it (a caller such as torch keeps its own notion of the current device).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == device) return hipSuccess;
        e = hipSetDevice(device);
        switched = e == hipSuccess;
        return e;
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace pamd
