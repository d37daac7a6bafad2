"""The presets this build implements, importable by the reference's spelling
(``from torchflows.bijections.finite.autoregressive import RealNVP``; reference
bijections/finite/autoregressive/__init__.py:1-22).  The sigmoidal / UMNN families of that list are outside the hot
path and are not defined here: importing them raises ImportError."""
from torchflows_amd.bijections.finite.autoregressive.architectures import (  # noqa: F401
    NICE, RealNVP, MAF, IAF, CouplingRQNSF, MaskedAutoregressiveRQNSF, InverseAutoregressiveRQNSF, CouplingLRS,
    MaskedAutoregressiveLRS, InverseAutoregressiveLRS)
