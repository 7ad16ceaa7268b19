"""gmrm_amd -- MI355X (gfx950) implementation of gmrm's per-marker Gibbs update hot path.

The compute lives in libgmrm_hip.so (hand-written HIP kernels behind the C ABI declared in
include/gmrm_hip.h); this package is the thin host-side mirror of the reference's
Bayes / Phenotype call surface used by tests, bench.py and the Python launcher.
There is no CPU fallback: loading fails loudly when the library is missing, and every
compute call fails when no HIP device is visible.
"""
from ._lib import GmrmError, load_library, library_path  # noqa: F401
from .api import Context, Sampler, Hyper, block_of_markers, im4_of, prepare_phenotype  # noqa: F401

__all__ = ["GmrmError", "load_library", "library_path", "Context", "Sampler", "Hyper",
           "block_of_markers", "im4_of", "prepare_phenotype"]
