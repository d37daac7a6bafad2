// Instantiation of the matrix-core flow program for D = 256 (see tfk_flow_mfma.h): its own
// translation unit so that the 32-element-per-lane kernels compile beside the others.
#include "tfk_flow_mfma.h"

namespace tfk {

int flow_mfma_launch_32(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                        float *logprob, int64_t N, const float *params, int n_params, const MProgram &prog,
                        int accumulate, hipStream_t s, const char *fn, const float *context, int C)
{
    return launch_m<32>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, accumulate, s, fn, context, C);
}

}  // namespace tfk
