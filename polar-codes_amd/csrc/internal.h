// Helpers shared by the two halves of the C ABI (capi.cpp: decoder, capi_encoder.cpp: encoder).
#pragma once
#include <string>

namespace pamd {

// records msg as the calling thread's pcd_last_error() text and returns code
int set_error(int code, const std::string& msg);

}  // namespace pamd
