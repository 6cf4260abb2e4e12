"""MI355X-native fast-convolution operator and GMRES for the Lippmann-Schwinger
equation: the FastM / FastM3D hot path of tanderson92/Fast_solver_Lippmann_Schwinger
behind the reference's own operator surface.  See DESIGN.md and include/lsfc.h."""
from .operators import (FastM, FastM3D, ConvergenceHistory, FFTconvolution, buildFastConvolution,
                        buildFastConvolution3D, eltype, fastconvolution, gmres_, gmres_batch_, apply_batch, mul_, profile_apply,
                        referenceValsTrapRule, sampleG3D, sampleGConv, size, time_apply)
from .preconditioner import SparsifyingPreconditioner
from ._lib import LsfcError, device_count, load, host_register, host_unregister, host_empty

__all__ = ["FastM", "FastM3D", "ConvergenceHistory", "FFTconvolution", "buildFastConvolution",
           "buildFastConvolution3D", "eltype", "fastconvolution", "gmres_", "gmres_batch_", "apply_batch", "mul_", "profile_apply",
           "referenceValsTrapRule", "sampleG3D", "sampleGConv", "size", "time_apply", "LsfcError",
           "device_count", "load", "SparsifyingPreconditioner", "host_register", "host_unregister", "host_empty"]
