"""ctypes binding of libslamhip.so (declared in include/slamhip.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SLAMHIP_VARIANT=<name> selects a non-product build (its own file, libslamhip_<name>.so:
# `stamps` = the diagnostic phase-stamp build, `python -m slam_amd.build --stamps`; others are
# A/B builds of kernel experiments).  Nothing else can make one the loaded library.
VARIANT = os.environ.get('SLAMHIP_VARIANT', '') or 'product'
STAMPS = VARIANT == 'stamps'
LIB_PATH = os.path.join(HERE, 'libslamhip.so' if VARIANT == 'product' else f'libslamhip_{VARIANT}.so')

NPARAM = 13
PROF_LEN = 45

_dp = C.POINTER(C.c_double)


class SlamhipDesc(C.Structure):
    """Mirror of `slamhip_desc` (include/slamhip.h)."""
    _fields_ = [
        ('nv', C.c_int32), ('ny', C.c_int32), ('nx', C.c_int32),
        ('nlayer', C.c_int32), ('npix', C.c_int32),
        ('nq1', C.c_int32), ('nq3', C.c_int32),
        ('ineed0', C.c_int32), ('ineed1', C.c_int32),
        ('kbeam', C.c_int32), ('disk', C.c_int32), ('envelope', C.c_int32),
        ('dv', C.c_double), ('sigma', C.c_double), ('pixperbeam', C.c_double),
        ('incl0', C.c_double), ('signmajor', C.c_double),
        ('signminor', C.c_double), ('vunit', C.c_double),
        ('Xnest0', _dp), ('Ynest0', _dp), ('xneed', _dp), ('yneed', _dp),
        ('xobs', _dp), ('yobs', _dp), ('v_valid', _dp), ('data_valid', _dp),
        ('gaussbeam', _dp),
        ('max_scratch_bytes', C.c_int64),
        ('mom0decon', _dp),
        ('precision', C.c_int32),
        ('beam_tol', C.c_double),
    ]


# name -> (restype, argtypes); every symbol include/slamhip.h declares
SYMBOLS = {
    'slamhip_create': (C.c_int, [C.POINTER(SlamhipDesc), C.POINTER(C.c_void_p)]),
    'slamhip_chi2_batch': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    'slamhip_chi2_batch_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    'slamhip_chi2_batch_static': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p]),
    'slamhip_static_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'slamhip_chi2_stats': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    'slamhip_cubemodel': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    'slamhip_moments': (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_double, _dp, _dp]),
    'slamhip_iunif': (C.c_int, [C.c_void_p, _dp, _dp]),
    'slamhip_iconv': (C.c_int, [C.c_void_p, _dp, _dp]),
    'slamhip_surface': (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    'slamhip_prof': (C.c_int, [C.c_void_p, C.c_double, _dp, _dp]),
    'slamhip_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64),
                               C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'slamhip_window_shape': (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 4),
    'slamhip_beam_rank': (C.c_int, [C.c_void_p]),
    'slamhip_set_conv_mode': (C.c_int, [C.c_void_p, C.c_int]),
    'slamhip_kernel_count': (C.c_int, []),
    'slamhip_kernel_name': (C.c_char_p, [C.c_int]),
    'slamhip_profile_begin': (C.c_int, [C.c_void_p]),
    'slamhip_profile_end': (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64)]),
    'slamhip_group_counts': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    'slamhip_last_error': (C.c_char_p, [C.c_void_p]),
    'slamhip_destroy': (None, [C.c_void_p]),
    'slamhip_device_count': (C.c_int, []),
    'slamhip_version': (C.c_char_p, []),
}

_lib = None


class SlamhipError(RuntimeError):
    pass


def load():
    """Load libslamhip.so (once) and declare every prototype."""
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
    check_build(lib.slamhip_version().decode(), 'slamhip', LIB_PATH, VARIANT)
    _lib = lib
    return lib


def build_words(version):
    return dict(w.split('=', 1) for w in version.split() if '=' in w)


def build_info(version):
    """`src` hash and stamps state out of a version string
    ('... src=<hash> stamps=<0|1> variant=<name>'; only src for libslampv)."""
    words = build_words(version)
    return words.get('src'), words.get('stamps', '0') == '1'


def check_build(version, name, path, variant='product'):
    """Refuse a binary that was not built from the sources next to it, or a
    diagnostic / experiment build standing in for the product (or the reverse)."""
    from . import build as _build
    words = build_words(version)
    sha, is_stamps = build_info(version)
    want = _build.src_sha(name)
    if want is not None and sha != want:
        raise SlamhipError(
            f'{path} was built from other sources (library src={sha}, tree src={want}): '
            'rebuild with `python -m slam_amd.build`')
    got = words.get('variant', 'product')
    if got != variant or is_stamps != (variant == 'stamps'):
        raise SlamhipError(
            f'{path} reports variant={got} stamps={int(is_stamps)} but SLAMHIP_VARIANT asks for '
            f'{variant}: diagnostic and experiment builds have their own files and are never the product')


def as_dp(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
