// Instantiation of the straight-line coupling-chain kernel for D = 16 (event sizes <= 16: two row elements per lane and plane,
// 8-byte accesses; affine / shift couplings only) (see tfk_flow_chain.h).
#include "tfk_flow_chain.h"

namespace tfk {

int flow_chain_launch_2(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                         float *logprob, int64_t N, const float *params, int n_params, const ChainProg &prog,
                         int kind, int steps2, int flags, int xw, hipStream_t s, const char *fn)
{
    return launch_chain<2>(x, z, logdet, loc, log_scale, logprob, N, params, n_params, prog, kind, steps2, flags, xw, s, fn);
}

}  // namespace tfk
