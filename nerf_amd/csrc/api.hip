// Version and error-string entry points of libnerf_hip.
#include "common.hpp"

extern "C" int nerf_version(void) { return NERF_HIP_VERSION; }

extern "C" const char* nerf_error_string(int code) {
    switch (code) {
        case NERF_OK: return "ok";
        case NERF_E_NULL: return "a required pointer is NULL";
        case NERF_E_SHAPE: return "a size is outside the supported range";
        case NERF_E_ALIGN: return "a pointer misses its required alignment";
        case NERF_E_ARCH: return "unsupported network architecture";
        case NERF_E_WORKSPACE: return "workspace too small";
        case NERF_E_DTYPE: return "unknown precision selector";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}
