"""HipLikelihood: a fit's constants resident on one MI355X + the batched
chi-square / model-cube entry points of libslamhip.so.

This is the thin host layer between `ChannelFit` (slam_amd/channelfit.py) and
the C ABI (include/slamhip.h).  It owns one `slamhip_ctx`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SlamhipError, as_dp, f64

PARAMKEYS = ('Mstar', 'Rc', 'cs', 'h1', 'h2', 'pI', 'Rin', 'Ienv',
             'xoff', 'yoff', 'voff', 'incloff', 'paoff')


class HipLikelihood:
    """GPU-resident channelfit likelihood for one `makegrid` result.

    `setup` is a mapping with the makegrid constants (see
    `slam_amd.channelfit.ChannelFit.setup_dict`): nlayer, npix, nq1, nq3,
    ineed0, ineed1, dv, sigma, pixperbeam, incl0, signmajor, signminor, vunit,
    disk, envelope, Xnest0, Ynest0, xneed, yneed, x, y, v_valid, data_valid,
    gaussbeam, and optionally mom0decon (the 'mom0*' scalings; gaussbeam is then
    the x-flipped kernel).
    """

    def __init__(self, setup, max_scratch_bytes=0, precision='f64', beam_tol=0.0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        if self._lib.slamhip_device_count() < 1:
            raise SlamhipError('no HIP device visible: the channelfit '
                               'likelihood runs on the GPU only')
        s = setup
        keep = {k: f64(s[k]) for k in ('Xnest0', 'Ynest0', 'xneed', 'yneed',
                                       'x', 'y', 'v_valid', 'data_valid',
                                       'gaussbeam')}
        nv, ny, nx = keep['data_valid'].shape
        if keep['v_valid'].shape != (nv,) or keep['x'].shape != (nx,) \
                or keep['y'].shape != (ny,):
            raise ValueError('data_valid / v_valid / x / y shapes disagree')
        nlayer, npix = int(s['nlayer']), int(s['npix'])
        if keep['Xnest0'].shape != (nlayer, npix, npix) \
                or keep['Ynest0'].shape != (nlayer, npix, npix):
            raise ValueError('Xnest0 / Ynest0 must be (nlayer, npix, npix)')
        nneed = int(s['ineed1']) - int(s['ineed0'])
        if keep['xneed'].shape != (nneed,) or keep['yneed'].shape != (nneed,):
            raise ValueError('xneed / yneed must have ineed1-ineed0 entries')
        K = keep['gaussbeam'].shape[0]
        if keep['gaussbeam'].shape != (K, K):
            raise ValueError('gaussbeam must be square')
        self.mom0 = s.get('mom0decon') is not None
        if self.mom0:
            keep['mom0decon'] = f64(s['mom0decon'])
            if keep['mom0decon'].shape != (ny, nx):
                raise ValueError('mom0decon must be (ny, nx)')
        d = _lib.SlamhipDesc(
            nv=nv, ny=ny, nx=nx, nlayer=nlayer, npix=npix,
            nq1=int(s['nq1']), nq3=int(s['nq3']), ineed0=int(s['ineed0']),
            ineed1=int(s['ineed1']), kbeam=K, disk=int(bool(s['disk'])),
            envelope=int(bool(s['envelope'])), dv=float(s['dv']),
            sigma=float(s['sigma']), pixperbeam=float(s['pixperbeam']),
            incl0=float(s['incl0']), signmajor=float(s['signmajor']),
            signminor=float(s['signminor']), vunit=float(s['vunit']),
            Xnest0=as_dp(keep['Xnest0']), Ynest0=as_dp(keep['Ynest0']),
            xneed=as_dp(keep['xneed']), yneed=as_dp(keep['yneed']),
            xobs=as_dp(keep['x']), yobs=as_dp(keep['y']),
            v_valid=as_dp(keep['v_valid']), data_valid=as_dp(keep['data_valid']),
            gaussbeam=as_dp(keep['gaussbeam']),
            max_scratch_bytes=int(max_scratch_bytes))
        if self.mom0:
            d.mom0decon = as_dp(keep['mom0decon'])
        levels = {'f64': 0, 'f32': 1, 'f32g': 2}
        if precision not in levels:
            raise ValueError(f'precision must be one of {sorted(levels)}')
        self.precision = precision
        d.precision = levels[precision]
        d.beam_tol = float(beam_tol)
        rc = self._lib.slamhip_create(C.byref(d), C.byref(self._ctx))
        if rc != 0:
            msg = self._errmsg()
            self.close()
            raise SlamhipError(f'slamhip_create failed ({rc}): {msg}')
        self.nv, self.ny, self.nx = nv, ny, nx
        self.nlayer, self.npix, self.nneed = nlayer, npix, nneed
        nleaf, wg, sep = C.c_int64(), C.c_int32(), C.c_int32()
        self._lib.slamhip_info(self._ctx, C.byref(nleaf), C.byref(wg),
                               C.byref(sep))
        self.nleaf, self.walkers_per_group = nleaf.value, wg.value
        self.separable_beam = bool(sep.value)
        self.beam_rank = int(self._lib.slamhip_beam_rank(self._ctx))
        self.kbeam = K

    def window_shape(self):
        """dict(rows, cols, kpad, fold) of the batch path's beam window (slamhip_window_shape)."""
        v = [C.c_int32() for _ in range(4)]
        self._check(self._lib.slamhip_window_shape(self._ctx, *[C.byref(x) for x in v]), 'window_shape')
        return dict(zip(('rows', 'cols', 'kpad', 'fold'), (x.value for x in v)))

    # -- plumbing ----------------------------------------------------------
    def _errmsg(self):
        if not self._ctx:
            return 'no context'
        m = self._lib.slamhip_last_error(self._ctx)
        return m.decode() if m else ''

    def _check(self, rc, what):
        if rc < 0:
            raise SlamhipError(f'{what} failed ({rc}): {self._errmsg()}')
        return rc

    def close(self):
        if getattr(self, '_ctx', None):
            self._lib.slamhip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def _params(p, batched):
        a = f64(p)
        if batched:
            a = np.atleast_2d(a)
            if a.ndim != 2 or a.shape[1] != _lib.NPARAM:
                raise ValueError('params must be [nw, 13]')
        elif a.shape != (_lib.NPARAM,):
            raise ValueError('params must have 13 entries')
        return a

    # -- hot path ----------------------------------------------------------
    def chi2_batch(self, P):
        """chi2 for every row of natural-unit parameters P[nw, 13] (host)."""
        a = self._params(P, True)
        out = np.empty(a.shape[0], dtype=np.float64)
        if a.shape[0]:
            self._check(self._lib.slamhip_chi2_batch(
                self._ctx, as_dp(a), a.shape[0], as_dp(out)), 'chi2_batch')
        return out

    def chi2_stats(self, P):
        """Hot path with get_scale's sums exposed: dict of arrays gf, ff, scale,
        chi2, live_tiles, tiles (one entry per row of P)."""
        a = self._params(P, True)
        st = np.empty((a.shape[0], 6), dtype=np.float64)
        if a.shape[0]:
            self._check(self._lib.slamhip_chi2_stats(
                self._ctx, as_dp(a), a.shape[0], as_dp(st)), 'chi2_stats')
        return dict(gf=st[:, 0], ff=st[:, 1], scale=st[:, 2], chi2=st[:, 3],
                    live_tiles=st[:, 4], tiles=st[:, 5])

    def chi2_batch_torch(self, P, out=None, stream=None):
        """Same, for a float64 CUDA tensor P[nw, 13]; enqueued on torch's
        current stream (or `stream`), returns a CUDA tensor without syncing.
        Calls on one engine must come from one host thread at a time; the
        library orders them on the device whatever streams they use."""
        import torch
        if not (P.is_cuda and P.dtype == torch.float64 and P.dim() == 2
                and P.shape[1] == _lib.NPARAM):
            raise ValueError('P must be a float64 CUDA tensor [nw, 13]')
        P = P.contiguous()
        if out is None:
            out = torch.empty(P.shape[0], dtype=torch.float64, device=P.device)
        st = stream if stream is not None else torch.cuda.current_stream(P.device)
        self._check(self._lib.slamhip_chi2_batch_dev(
            self._ctx, C.c_void_p(P.data_ptr()), P.shape[0],
            C.c_void_p(out.data_ptr()), C.c_void_p(st.cuda_stream)),
            'chi2_batch_dev')
        return out

    def chi2_batch_static_torch(self, P, out=None, stream=None):
        """`chi2_batch_torch` through slamhip_chi2_batch_static: same kernels and
        results, scratch in the library's static device array (no allocation
        inside the call; `out` should be passed by callers that cannot allocate)."""
        import torch
        if not (P.is_cuda and P.dtype == torch.float64 and P.dim() == 2
                and P.shape[1] == _lib.NPARAM and P.is_contiguous()):
            raise ValueError('P must be a contiguous float64 CUDA tensor [nw, 13]')
        if out is None:
            out = torch.empty(P.shape[0], dtype=torch.float64, device=P.device)
        st = stream if stream is not None else torch.cuda.current_stream(P.device)
        self._check(self._lib.slamhip_chi2_batch_static(
            self._ctx, C.c_void_p(P.data_ptr()), P.shape[0],
            C.c_void_p(out.data_ptr()), C.c_void_p(st.cuda_stream)),
            'chi2_batch_static')
        return out

    def static_info(self):
        """(bytes of the static scratch, bytes one walker needs, walkers per
        launch group of the allocation-free entry point; 0 = does not fit)."""
        a, b, w = C.c_int64(), C.c_int64(), C.c_int32()
        self._check(self._lib.slamhip_static_info(self._ctx, C.byref(a), C.byref(b),
                                                  C.byref(w)), 'static_info')
        return a.value, b.value, w.value

    # -- single-vector products / parity taps ------------------------------
    def cubemodel(self, p, convolving=True, return_stats=False):
        a = self._params(p, False)
        out = np.empty((self.nv, self.ny, self.nx), dtype=np.float64)
        st = np.empty(4, dtype=np.float64)
        self._check(self._lib.slamhip_cubemodel(
            self._ctx, as_dp(a), int(bool(convolving)), as_dp(out), as_dp(st)),
            'cubemodel')
        if return_stats:
            return out, dict(gf=st[0], ff=st[1], scale=st[2], chi2=st[3])
        return out

    def moments(self, p, dv, clip):
        """(mom0, mom1) [ny, nx] of the model cube for `p` over its channels:
        mom0 = sum(d) dv, mom1 = sum(d v) / sum(d), NaN where mom0 < clip."""
        a = self._params(p, False)
        m0 = np.empty((self.ny, self.nx), dtype=np.float64)
        m1 = np.empty((self.ny, self.nx), dtype=np.float64)
        self._check(self._lib.slamhip_moments(
            self._ctx, as_dp(a), float(dv), float(clip), as_dp(m0), as_dp(m1)), 'moments')
        return m0, m1

    def iunif(self, p):
        a = self._params(p, False)
        out = np.empty((self.nv, self.nneed, self.nneed), dtype=np.float64)
        self._check(self._lib.slamhip_iunif(self._ctx, as_dp(a), as_dp(out)),
                    'iunif')
        return out

    def iconv(self, p):
        a = self._params(p, False)
        out = np.empty((self.nv, self.nneed, self.nneed), dtype=np.float64)
        self._check(self._lib.slamhip_iconv(self._ctx, as_dp(a), as_dp(out)),
                    'iconv')
        return out

    def surface(self, p, slot):
        """(xdisk, vlos) of surface slot 0..3 on the nested grid, or None."""
        a = self._params(p, False)
        shape = (self.nlayer, self.npix, self.npix)
        x = np.empty(shape, dtype=np.float64)
        v = np.empty(shape, dtype=np.float64)
        rc = self._check(self._lib.slamhip_surface(
            self._ctx, as_dp(a), int(slot), as_dp(x), as_dp(v)), 'surface')
        return None if rc == 1 else (x, v)

    def prof(self, cs):
        p = np.empty(_lib.PROF_LEN, dtype=np.float64)
        d = C.c_double()
        self._check(self._lib.slamhip_prof(self._ctx, float(cs), as_dp(p),
                                           C.byref(d)), 'prof')
        return p, d.value

    # -- measurement -------------------------------------------------------
    def profile_begin(self):
        self._check(self._lib.slamhip_profile_begin(self._ctx), 'profile_begin')

    def profile_end(self):
        """{kernel name: (total ms, launches)} since profile_begin (HIP events
        on the launch stream)."""
        n = self._lib.slamhip_kernel_count()
        ms = np.zeros(n, dtype=np.float64)
        cnt = (C.c_int64 * n)()
        self._check(self._lib.slamhip_profile_end(self._ctx, as_dp(ms), cnt),
                    'profile_end')
        return {self._lib.slamhip_kernel_name(i).decode(): (float(ms[i]), int(cnt[i]))
                for i in range(n)}

    def group_counts(self, nw):
        """Counts of what the last launch group of `nw` walkers touched (see
        slamhip_group_counts), as a dict."""
        out = (C.c_int64 * 8)()
        self._check(self._lib.slamhip_group_counts(self._ctx, int(nw), out), 'group_counts')
        keys = ('live_windows', 'window_tiles', 'window_rows', 'col_blocks', 'lit_planes',
                'geom_groups', 'profile_hits', 'profile_wave_hits')
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def set_conv_mode(self, mode):
        self._check(self._lib.slamhip_set_conv_mode(self._ctx, int(mode)),
                    'set_conv_mode')
