"""Inert stand-ins for third-party packages the reference imports at module
load time but that contribute no arithmetic to the channelfit hot path
(SURVEY.md Appendix A).  Used ONLY by make_golden.py, in the build container,
to import /root/reference and record golden vectors.  Never shipped or
imported by the product or by the GPU tests.
"""
import math
import sys
import types


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Val:
    def __init__(self, v):
        self.value = v


class _Const:
    def __init__(self, v):
        self.si = _Val(v)


class _Unit:
    def __init__(self, table):
        self._t = table

    def to(self, other):
        return self._t[other]


def install():
    G = 6.6743e-11                       # CODATA 2018
    GM_sun = 1.3271244e20                # IAU 2015 nominal solar mass parameter
    consts = _mod('astropy.constants', G=_Const(G), M_sun=_Const(GM_sun / G),
                  c=_Const(299792458.0))
    units = _mod('astropy.units', au=_Unit({'m': 149597870700.0}),
                 deg=_Unit({'radian': math.pi / 180}), degree=1.0)
    fits = _mod('astropy.io.fits')
    io = _mod('astropy.io', fits=fits)
    wcs = _mod('astropy.wcs')
    coords = _mod('astropy.coordinates', SkyCoord=object)
    _mod('astropy', constants=consts, units=units, io=io, wcs=wcs,
         coordinates=coords)
    moves = _mod('emcee.moves', StretchMove=lambda *a, **k: None)
    _mod('emcee', moves=moves, EnsembleSampler=None)
    _mod('corner')
    _mod('dynesty', DynamicNestedSampler=None, NestedSampler=None,
         utils=None, plotting=None)
    _mod('numba', jit=lambda *a, **k: (lambda f: f), prange=range)
