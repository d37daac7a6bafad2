// Instantiation of the single-launch spline-coupling chain kernel for D = 32 (event sizes <= 32 padded to 32 instead of 64) (see tfk_flow_rqs_chain.h).
#include "tfk_flow_rqs_chain.h"

namespace tfk {

int flow_rqs_chain_launch_4(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                             float *logprob, int64_t N, const float *params, const RqsChainProg &prog, int inverse,
                             int steps2, int flags, int xw, hipStream_t s, const char *fn,
                             const float *context, int Cn)
{
    return launch_rqs_chain<4>(x, z, logdet, loc, log_scale, logprob, N, params, prog, inverse, steps2, flags, xw, s, fn, context, Cn);
}

}  // namespace tfk
