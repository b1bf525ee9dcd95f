"""ngmlr_amd -- MI355X-native convex-gap banded Smith-Waterman (ngmlr's ConvexAlignFast hot path).

Only what the hot path needs lives here:

* ``csrc/``      hand-written gfx950 HIP kernels + the C-ABI library (``libcvxalign.so``)
* ``capi.py``    ctypes binding of ``include/cvx_align.h`` (fails loudly if the library is missing)
* ``aligner.py`` host-side mirror of the reference's ``IAlignment`` surface for this path
* ``synth.py``   seeded synthetic tile generator (corridor formulas of the reference's caller)
"""
__version__ = "0.1.0"
