"""slip_lu_amd -- MI355X-native REF sparse LU hot path behind the SLIP_LU C API.

The product is the HIP library (csrc/libslip_hip.so, C ABI in include/slip_hip.h)
and the GMP-typed drop-in shim (csrc/libslip_lu_hip.so); this package is the thin
ctypes layer the tests and bench.py use.
"""
from . import api  # noqa: F401
from .api import Factorization, SlipError, factorize, ints_to_slab, matgen, read_triplet, write_triplet  # noqa: F401
