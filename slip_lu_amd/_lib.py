"""ctypes loader for libslip_hip.so (the HIP product library).

The product path has no CPU fallback: if the in-tree HIP library is missing or
cannot be loaded this module raises.  The environment variable
SLIP_HIP_LIBRARY may point to another build of the SAME C ABI (the test-suite
uses it to load the CPU emulation build of the kernel source, tests/emu).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(_HERE, "csrc", "libslip_hip.so")


class Options(C.Structure):
    _fields_ = [("pivot", C.c_int32), ("tol", C.c_double), ("limb_cap", C.c_int32),
                ("waves", C.c_int32), ("lnz_hint", C.c_int64), ("unz_hint", C.c_int64),
                ("workers", C.c_int32), ("reserved", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("n", C.c_int32), ("K", C.c_int32), ("status", C.c_int32), ("window_end", C.c_int32),
                ("lnz", C.c_int64), ("unz", C.c_int64), ("l_limbs", C.c_int64), ("u_limbs", C.c_int64),
                ("n_upd", C.c_int64), ("b_read", C.c_int64), ("b_write", C.c_int64), ("n_src", C.c_int64),
                ("l_streamed", C.c_int64), ("max_limbs", C.c_int64),
                ("kernel_ms", C.c_double), ("launches", C.c_int32), ("xcap_digits", C.c_int32),
                ("limb_macs", C.c_int64), ("workers", C.c_int32), ("waves", C.c_int32),
                ("lds_bytes", C.c_int32), ("short_commits", C.c_int32), ("committer_commits", C.c_int32), ("farm_jobs", C.c_int32), ("farm_items", C.c_int32), ("pad2", C.c_int32),
                ("engine_commits", C.c_int32), ("engine_sources", C.c_int32), ("retractions", C.c_int32), ("reexports", C.c_int32)]


EXPORTS = ("slip_hip_default_options", "slip_hip_device_count", "slip_hip_factor_create",
           "slip_hip_factor_reset", "slip_hip_factor_run", "slip_hip_factor_info",
           "slip_hip_factor_download", "slip_hip_factor_destroy", "slip_hip_matgen",
           "slip_hip_free", "slip_hip_wave_op_test", "slip_hip_version",
           "slip_hip_factor_phase_cycles", "slip_hip_factor_solve", "slip_hip_factor_solve_ms",
           "slip_hip_factor_from_factors", "slip_hip_factor_rescale", "slip_hip_factor_set_prefix", "slip_hip_pool_release", "slip_hip_read_triplet", "slip_hip_write_triplet")

_libs = {}


def library_path():
    return os.environ.get("SLIP_HIP_LIBRARY", DEFAULT_SO)


def load(path=None):
    """Load (once) and return the C library; raises OSError if it is missing."""
    path = path or library_path()
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build the HIP extension first "
                      "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.slip_hip_default_options.argtypes = [C.POINTER(Options)]
    lib.slip_hip_default_options.restype = None
    lib.slip_hip_device_count.restype = C.c_int
    lib.slip_hip_factor_create.argtypes = [C.POINTER(vp), C.c_int32, vp, vp, vp, vp, vp, C.POINTER(Options)]
    lib.slip_hip_factor_create.restype = C.c_int
    lib.slip_hip_factor_reset.argtypes = [vp]
    lib.slip_hip_factor_run.argtypes = [vp, C.c_int32, vp]
    lib.slip_hip_factor_info.argtypes = [vp, C.POINTER(Info)]
    lib.slip_hip_factor_download.argtypes = ([vp] + [vp] * 4 + [C.POINTER(C.c_int64)] + [vp] * 4 + [C.POINTER(C.c_int64)]
                                             + [vp] * 2 + [C.POINTER(C.c_int64), vp])
    lib.slip_hip_factor_destroy.argtypes = [vp]
    lib.slip_hip_factor_destroy.restype = None
    lib.slip_hip_matgen.argtypes = [C.c_int32, C.c_double, C.c_int32, C.c_uint64,
                                    C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.slip_hip_read_triplet.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.slip_hip_write_triplet.argtypes = [C.c_char_p, C.c_int32, vp, vp, vp, vp]
    lib.slip_hip_free.argtypes = [vp]
    lib.slip_hip_free.restype = None
    lib.slip_hip_wave_op_test.argtypes = [C.c_int32] * 5 + [vp, vp, vp]
    lib.slip_hip_version.restype = C.c_char_p
    lib.slip_hip_factor_solve.argtypes = [vp, C.c_int32, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64), vp]
    lib.slip_hip_factor_solve.restype = C.c_int
    lib.slip_hip_factor_from_factors.argtypes = [C.POINTER(vp), C.c_int32] + [vp] * 9 + [C.POINTER(Options)]
    lib.slip_hip_factor_from_factors.restype = C.c_int
    lib.slip_hip_factor_rescale.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.slip_hip_factor_rescale.restype = C.c_int
    lib.slip_hip_factor_set_prefix.argtypes = [vp, C.c_int32] + [vp] * 9
    lib.slip_hip_factor_set_prefix.restype = C.c_int
    lib.slip_hip_factor_solve_ms.argtypes = [vp]
    lib.slip_hip_factor_solve_ms.restype = C.c_double
    _libs[path] = lib
    return lib
