"""Architecture presets in one place: ``from torchflows.architectures import RealNVP`` is the import the reference's
README shows (its v1.2.0 tree keeps the presets under ``bijections.finite.autoregressive.architectures`` and
``bijections.finite.multiscale.architectures`` only); both spellings resolve to the same classes here."""
from torchflows_amd.bijections.finite.autoregressive.architectures import *  # noqa: F401,F403
from torchflows_amd.bijections.finite.autoregressive.architectures import (  # noqa: F401
    RealNVP, CouplingRQNSF, CouplingLRS, NICE, MAF, IAF, MaskedAutoregressiveRQNSF,
    InverseAutoregressiveRQNSF, MaskedAutoregressiveLRS, InverseAutoregressiveLRS)
from torchflows_amd.bijections.finite.multiscale.architectures import *  # noqa: F401,F403
