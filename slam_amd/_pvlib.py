"""ctypes binding of libslampv.so (declared in include/slampv.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

from ._lib import SlamhipError, check_build

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libslampv.so')

NPARAM = 6
NTAB = 5

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class SlampvLevel(C.Structure):
    """Mirror of `slampv_level`."""
    _fields_ = [
        ('nx', C.c_int32), ('ny', C.c_int32), ('nz', C.c_int32),
        ('bx0', C.c_int32), ('bx1', C.c_int32), ('by0', C.c_int32),
        ('by1', C.c_int32), ('nsub', C.c_int32), ('pad_', C.c_int32),
        ('dz', C.c_double),
        ('idx_t', _ip * 2), ('j_org', _dp * 2), ('elos_r', _dp * 2),
        ('elos_t', _dp * 2), ('elos_p', _dp * 2),
    ]


class SlampvDesc(C.Structure):
    """Mirror of `slampv_desc`."""
    _fields_ = [
        ('nlevels', C.c_int32), ('nv', C.c_int32), ('nx_out', C.c_int32),
        ('nsubgrid', C.c_int32), ('ntheta', C.c_int32), ('nr', C.c_int32),
        ('signmajor', C.c_int32), ('signminor', C.c_int32),
        ('dlnr', C.c_double), ('g_si', C.c_double), ('m_sun', C.c_double),
        ('au_m', C.c_double), ('sigma', C.c_double), ('rarea', C.c_double),
        ('table', _dp), ('vedge', _dp), ('gauss_v', _dp), ('gauss_xy', _dp),
        ('obs_major', _dp), ('obs_minor', _dp),
        ('levels', C.POINTER(SlampvLevel)),
        ('max_scratch_bytes', C.c_int64),
    ]


SYMBOLS = {
    'slampv_create': (C.c_int, [C.POINTER(SlampvDesc), C.POINTER(C.c_void_p)]),
    'slampv_lnprob_batch': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    'slampv_lnprob_batch_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    'slampv_stats': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    'slampv_mockpvd': (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    'slampv_own_tau': (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp]),
    'slampv_tau': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    'slampv_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64),
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'slampv_kernel_count': (C.c_int, []),
    'slampv_kernel_name': (C.c_char_p, [C.c_int]),
    'slampv_profile_begin': (C.c_int, [C.c_void_p]),
    'slampv_profile_end': (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64)]),
    'slampv_last_error': (C.c_char_p, [C.c_void_p]),
    'slampv_destroy': (None, [C.c_void_p]),
    'slampv_device_count': (C.c_int, []),
    'slampv_version': (C.c_char_p, []),
}

_lib = None


def load():
    """Load libslampv.so (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlamhipError(
            f'{LIB_PATH} not found: build it with `python -m slam_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    check_build(lib.slampv_version().decode(), 'slampv', LIB_PATH)
    _lib = lib
    return lib
