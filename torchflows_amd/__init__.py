"""torchflows_amd -- MI355X-native coupling-flow hot path behind the torchflows plugin surface.

``Flow(bijection).log_prob / .sample`` with affine and rational-quadratic-spline coupling
bijections, running on hand-written gfx950 HIP kernels (``libtfk.so``, C-ABI in
``include/tfk.h``) for fp32 tensors on a HIP device, and on an ATen composite path for
autograd / fp64 / host tensors.  Module paths mirror the reference
(``torchflows.flows``, ``torchflows.bijections.finite.autoregressive.architectures`` ...).
"""
from torchflows_amd.flows import Flow, BaseFlow  # noqa: F401
from torchflows_amd.bijections.finite.autoregressive.architectures import (  # noqa: F401
    RealNVP, CouplingRQNSF, CouplingLRS, NICE, MAF, IAF, MaskedAutoregressiveRQNSF,
    InverseAutoregressiveRQNSF, MaskedAutoregressiveLRS, InverseAutoregressiveLRS)

__version__ = "0.1.0"
