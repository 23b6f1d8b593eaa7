"""cellscreen -- host-side mirror of the reference's screening/training interface over
libcellscreen.so (hand-written gfx950 HIP).  Importing this package loads no native code;
Engine / ProductionMutantScreening do, and fail loudly when the library or a GPU is missing."""
from . import spec  # noqa: F401
from .spec import CAEWeights, DetectorParams, OCSVMParams  # noqa: F401

__all__ = ["spec", "CAEWeights", "DetectorParams", "OCSVMParams"]
