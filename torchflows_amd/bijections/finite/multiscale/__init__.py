from torchflows_amd.bijections.finite.multiscale.architectures import (  # noqa: F401
    AffineGlow, MultiscaleNICE, MultiscaleRealNVP, ShiftGlow)
from torchflows_amd.bijections.finite.multiscale.base import (  # noqa: F401
    CheckerboardCoupling, ChannelWiseCoupling, GlowChannelWiseCoupling, GlowCheckerboardCoupling,
    Invertible1x1ConvolutionalCoupling, MultiscaleBijection, NormalizedChannelWiseCoupling,
    NormalizedCheckerboardCoupling, Squeeze)
from torchflows_amd.bijections.finite.multiscale.coupling import (  # noqa: F401
    ChannelWiseHalfSplit, Checkerboard, make_image_coupling)
