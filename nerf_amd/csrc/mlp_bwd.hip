// placeholder until the backward kernels land
#include "mlp_common.hpp"
extern "C" int64_t nerf_mlp_bwd_workspace_bytes(int64_t n_samples) { (void)n_samples; return 0; }
extern "C" int nerf_mlp_bwd(int64_t M, const float* d_raw, const void* packed, int precision, const void* saved,
                            void* workspace, int64_t workspace_bytes, float* d_params, void* stream) {
    return NERF_E_ARCH;
}
