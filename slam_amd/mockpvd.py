"""MockPVD: mock position-velocity diagrams of a disk + UCM envelope on the GPU.

Drop-in for the reference's `pvfitting.mockpvd.MockPVD` (same constructor and
`generate_mockpvd` signature) plus a batched likelihood.  The grid, tables and
per-cell constants are built once on the host (slam_amd/pvgrid.py); every
evaluation runs in libslampv.so (include/slampv.h).  There is no CPU fallback.

Semantics kept from the reference (pvfitting/mockpvd.py, precalculation.py):
the geometry caches, the beam kernel and the channel edges are fixed by the
FIRST evaluation (its Rc, incl, linewidth and, for axis='both', pa_major) and
never refreshed -- here they belong to the instance instead of module globals.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _pvlib, pvgrid
from ._lib import SlamhipError, as_dp, f64

AXES = ('major', 'minor')
PARAMKEYS = ('Mstar', 'Rc', 'alphainfall', 'taumax', 'frho', 'sig_mdl')


class PVEngine:
    """One slampv_ctx: constants resident on the current GPU."""

    def __init__(self, levels, layouts, v, nx_out, nsubgrid, signs, gauss_v,
                 gauss_xy, obs=None, sigma=1.0, rarea=1.0, max_scratch_bytes=0):
        self._lib = _pvlib.load()
        self._ctx = C.c_void_p()
        if self._lib.slampv_device_count() < 1:
            raise SlamhipError('no HIP device visible: the mock-PV model runs '
                               'on the GPU only')
        T = pvgrid.tables()
        keep = [T.packed(), f64(pvgrid.velocity_edges(f64(v)))]
        lv = (_pvlib.SlampvLevel * len(levels))()
        for l, (L, lay) in enumerate(zip(levels, layouts)):
            nx, ny, nz = L.shape
            lv[l].nx, lv[l].ny, lv[l].nz = nx, ny, nz
            if L.box is not None:
                (lv[l].bx0, lv[l].bx1), (lv[l].by0, lv[l].by1) = L.box[0], L.box[1]
            else:
                lv[l].bx0 = lv[l].bx1 = lv[l].by0 = lv[l].by1 = -1
            lv[l].nsub = L.nsub
            lv[l].dz = L.z[1] - L.z[0]
            for a, axis in enumerate(AXES):
                d = lay[axis]
                if d['idx_t'].shape != (nz, ny, nx):
                    raise ValueError('cell arrays must be [nz][ny][nx]')
                keep.append(d)
                lv[l].idx_t[a] = d['idx_t'].ctypes.data_as(_pvlib._ip)
                for k in ('j_org', 'elos_r', 'elos_t', 'elos_p'):
                    getattr(lv[l], k)[a] = as_dp(d[k])
        nv = len(v)
        self.nv, self.nx_out = nv, int(nx_out)
        self.shapes = [L.shape for L in levels]
        nx0, ny0, _ = self.shapes[0]
        if nx0 != nx_out * nsubgrid:
            raise ValueError('level-0 nx must be nx_out * nsubgrid')
        d = _pvlib.SlampvDesc(
            nlevels=len(levels), nv=nv, nx_out=int(nx_out), nsubgrid=int(nsubgrid),
            ntheta=T.ntheta, nr=T.nr, signmajor=int(signs[0]), signminor=int(signs[1]),
            dlnr=T.dlnr, g_si=pvgrid.G_SI, m_sun=pvgrid.M_SUN, au_m=pvgrid.AU_M,
            sigma=float(sigma), rarea=float(rarea), table=as_dp(keep[0]),
            vedge=as_dp(keep[1]), levels=lv, max_scratch_bytes=int(max_scratch_bytes))
        if gauss_v is not None:
            gv = f64(np.ravel(gauss_v))
            if gv.shape != (nv,):
                raise ValueError('gauss_v must have nv entries')
            keep.append(gv)
            d.gauss_v = as_dp(gv)
        if gauss_xy is not None:
            gxy = f64(gauss_xy)
            if gxy.shape != (ny0, nx0):
                raise ValueError('gauss_xy must be [ny0][nx0]')
            keep.append(gxy)
            d.gauss_xy = as_dp(gxy)
        self.has_obs = obs is not None
        if obs is not None:
            om, on = f64(obs[0]), f64(obs[1])
            if om.shape != (nv, nx_out) or on.shape != (nv, nx_out):
                raise ValueError('observed PV diagrams must be [nv][nx]')
            keep += [om, on]
            d.obs_major, d.obs_minor = as_dp(om), as_dp(on)
        rc = self._lib.slampv_create(C.byref(d), C.byref(self._ctx))
        del keep
        if rc != 0:
            msg = self._errmsg()
            self.close()
            raise SlamhipError(f'slampv_create failed ({rc}): {msg}')
        live, wg, nt = C.c_int64(), C.c_int32(), C.c_int32()
        self._lib.slampv_info(self._ctx, C.byref(live), C.byref(wg), C.byref(nt))
        self.live_cells, self.walkers_per_group = live.value, wg.value
        self.tasks_per_walker = nt.value

    def _errmsg(self):
        if not self._ctx:
            return 'no context'
        m = self._lib.slampv_last_error(self._ctx)
        return m.decode() if m else ''

    def _check(self, rc, what):
        if rc < 0:
            raise SlamhipError(f'{what} failed ({rc}): {self._errmsg()}')
        return rc

    def close(self):
        if getattr(self, '_ctx', None):
            self._lib.slampv_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _params(q, batched):
        a = f64(q)
        if batched:
            a = np.atleast_2d(a)
            if a.ndim != 2 or a.shape[1] != _pvlib.NPARAM:
                raise ValueError('params must be [nw, 6]')
        elif a.shape != (_pvlib.NPARAM,):
            raise ValueError('params must have 6 entries')
        return a

    def lnprob_batch(self, Q):
        """lnprob (_pvfitting.py:181-195) for every row of natural-unit Q[nw, 6]."""
        a = self._params(Q, True)
        out = np.empty(a.shape[0], dtype=np.float64)
        if a.shape[0]:
            self._check(self._lib.slampv_lnprob_batch(
                self._ctx, as_dp(a), a.shape[0], as_dp(out)), 'lnprob_batch')
        return out

    def lnprob_batch_torch(self, Q, out=None, stream=None):
        """Same for a float64 CUDA tensor Q[nw, 6]; enqueued on `stream` (default:
        torch's current stream), not synchronised.  Calls on one engine share its scratch
        set: the library orders a call behind the previous one when their streams differ
        (they never overlap); use one engine per stream for concurrency."""
        import torch
        if not (Q.is_cuda and Q.dtype == torch.float64 and Q.dim() == 2
                and Q.shape[1] == _pvlib.NPARAM):
            raise ValueError('Q must be a float64 CUDA tensor [nw, 6]')
        Q = Q.contiguous()
        if out is None:
            out = torch.empty(Q.shape[0], dtype=torch.float64, device=Q.device)
        st = stream if stream is not None else torch.cuda.current_stream(Q.device)
        self._check(self._lib.slampv_lnprob_batch_dev(
            self._ctx, C.c_void_p(Q.data_ptr()), Q.shape[0],
            C.c_void_p(out.data_ptr()), C.c_void_p(st.cuda_stream)),
            'lnprob_batch_dev')
        return out

    def stats(self, Q):
        a = self._params(Q, True)
        st = np.empty((a.shape[0], 6), dtype=np.float64)
        if a.shape[0]:
            self._check(self._lib.slampv_stats(self._ctx, as_dp(a), a.shape[0],
                                               as_dp(st)), 'stats')
        return dict(fflux=st[:, 0], chi2=st[:, 1], lnprob=st[:, 2],
                    taumax_major=st[:, 3], taumax_minor=st[:, 4], npix=st[:, 5])

    def mockpvd(self, q):
        """Unscaled [major, minor] PV diagrams, each [nv][nx_out]."""
        a = self._params(q, False)
        major = np.empty((self.nv, self.nx_out), dtype=np.float64)
        minor = np.empty_like(major)
        self._check(self._lib.slampv_mockpvd(self._ctx, as_dp(a), as_dp(major),
                                             as_dp(minor)), 'mockpvd')
        return major, minor

    def own_tau(self, q, axis, level):
        a = self._params(q, False)
        nx, ny, _ = self.shapes[level]
        out = np.empty((self.nv, ny, nx), dtype=np.float64)
        self._check(self._lib.slampv_own_tau(self._ctx, as_dp(a), AXES.index(axis),
                                             int(level), as_dp(out)), 'own_tau')
        return out

    def tau(self, q, axis):
        a = self._params(q, False)
        nx, ny, _ = self.shapes[0]
        out = np.empty((self.nv, ny, nx), dtype=np.float64)
        self._check(self._lib.slampv_tau(self._ctx, as_dp(a), AXES.index(axis),
                                         as_dp(out)), 'tau')
        return out

    def profile_begin(self):
        self._check(self._lib.slampv_profile_begin(self._ctx), 'profile_begin')

    def profile_end(self):
        n = self._lib.slampv_kernel_count()
        ms = np.zeros(n, dtype=np.float64)
        cnt = (C.c_int64 * n)()
        self._check(self._lib.slampv_profile_end(self._ctx, as_dp(ms), cnt),
                    'profile_end')
        return {self._lib.slampv_kernel_name(i).decode(): (float(ms[i]), int(cnt[i]))
                for i in range(n)}


class MockPVD:
    """Same constructor as the reference's MockPVD (pvfitting/mockpvd.py:31-95)."""

    def __init__(self, x, z, v, nnest=None, nsubgrid: int = 1, xlim=None,
                 ylim=None, zlim=None, beam=None, reslim: float = 10,
                 signmajor: int = 1, signminor: int = 1, pa_major: float = 0,
                 pa_minor: float = 90, max_scratch_bytes: int = 0):
        self._x, self._z = np.asarray(x, float), np.asarray(z, float)
        self._nx, self._nz = len(x), len(z)
        self.nsubgrid = nsubgrid
        if nsubgrid > 1:
            self.x = pvgrid.subgrid(self._x, nsubgrid)
            self.z = pvgrid.subgrid(self._z, nsubgrid)
        else:
            self.x, self.z = self._x, self._z
        self.nx, self.nz = len(self.x), len(self.z)
        self.v = np.asarray(v, float)
        self.nnest = nnest
        self.beam = beam
        self.pa_major, self.pa_minor = pa_major, pa_minor
        self.iradshift, self.signmajor, self.signminor = pvgrid.sign_convention(
            signmajor, signminor, pa_major, pa_minor)
        self.y = pvgrid.model_y_axis(self.x, beam)
        self.levels = pvgrid.build_levels(self.x, self.y, self.z, nnest or [],
                                          reslim, xlim, ylim, zlim)
        self.xx, self.vv = np.meshgrid(self._x, self.v)
        self.max_scratch_bytes = max_scratch_bytes
        self._consts = None      # per axis, per level: the first evaluation's caches
        self._frozen = None      # what the first evaluation fixed
        self._engine = None
        self._engine_key = None
        self._obs = None

    # -- reference-style introspection ------------------------------------
    def gridinfo(self):
        print('Nesting level: %i' % len(self.levels))
        print('Resolutions:')
        for l, L in enumerate(self.levels):
            print('   l=%i: (dx, dy, dz) = (%.2e au, %.2e au, %.2e au)' % (
                l, L.x[1] - L.x[0], L.y[1] - L.y[0], L.z[1] - L.z[0]))
            print('      : (nx, ny, nz) = (%i, %i, %i)' % L.shape)

    # -- setup --------------------------------------------------------------
    def set_observations(self, obs_major, obs_minor, sigma, rarea):
        """Observed PV diagrams [nv][nx] (NaN = masked channel), noise rms and
        the beam-area / pixel-area ratio (_pvfitting.py:110-123)."""
        self._obs = (f64(obs_major), f64(obs_minor), float(sigma), float(rarea))
        self._engine_key = None

    def _freeze(self, Rc, incl, linewidth, pa_beam):
        """Fill the caches the way the reference's first call does."""
        irad = np.abs(np.arcsin(np.sin(np.radians(incl)))) + self.iradshift
        self._consts = {axis: [pvgrid.cell_constants(L, axis, irad, Rc)
                               for L in self.levels] for axis in AXES}
        gauss_v = None if linewidth is None else pvgrid.spectral_kernel(self.v, linewidth)
        gauss_xy = None if self.beam is None else pvgrid.beam_kernel(
            self.x, self.y, self.beam, pa_beam)
        self._frozen = dict(rc_first=float(Rc), incl=incl, irad=irad,
                            linewidth=linewidth, gauss_v=gauss_v, gauss_xy=gauss_xy)

    def engine(self, Rc, incl, linewidth, rin, rout, pa_beam=None):
        if self._frozen is None:
            self._freeze(Rc, incl, linewidth,
                         self.pa_major if pa_beam is None else pa_beam)
        key = (rin, rout, id(self._obs))
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            layouts = [{axis: pvgrid.device_layout(L, self._consts[axis][l], rin, rout)
                        for axis in AXES} for l, L in enumerate(self.levels)]
            fz = self._frozen
            # the spectral kernel is applied only when the CALL asks for it
            # (mockpvd.py:307), with the first call's kernel
            obs = self._obs
            self._engine = PVEngine(
                self.levels, layouts, self.v, self._nx, self.nsubgrid,
                (self.signmajor, self.signminor), fz['gauss_v'], fz['gauss_xy'],
                obs=None if obs is None else obs[:2],
                sigma=1.0 if obs is None else obs[2],
                rarea=1.0 if obs is None else obs[3],
                max_scratch_bytes=self.max_scratch_bytes)
            self._engine_key = key
        return self._engine

    # -- the reference's entry point ---------------------------------------
    def generate_mockpvd(self, Mstar: float, Rc: float, alphainfall: float = 1.,
                         taumax: float = 1., frho: float = 1., incl: float = 89.,
                         pa: float | list = 0., linewidth: float | None = None,
                         rin: float = 1., rout: float | None = None,
                         axis: str = 'both'):
        """Mock PV diagram(s) (mockpvd.py:97-156): [major, minor] for
        axis='both', else the one cut (not flipped, like the reference)."""
        if axis not in ['major', 'minor', 'both']:
            print("ERROR\tgenerate_mockpvd: axis input must be 'major', 'minor' or 'both'.")
            return 0
        if self._frozen is not None and (linewidth is None) != (self._frozen['gauss_v'] is None):
            raise SlamhipError(
                'linewidth was %s on the first evaluation; the spectral kernel is '
                'fixed by that call' % ('None' if self._frozen['gauss_v'] is None else 'given'))
        eng = self.engine(Rc, incl, linewidth, rin, rout,
                          pa_beam=self.pa_major if axis == 'both' else pa)
        major, minor = eng.mockpvd([Mstar, Rc, alphainfall, taumax, frho, 0.])
        if axis == 'both':
            return [major, minor]
        if axis == 'major':
            return major[:, ::self.signmajor]
        return minor[:, ::self.signminor]

    def lnprob_batch(self, Q, incl, linewidth=None, rin=1., rout=None):
        """Batched modified log-likelihood for natural-unit rows
        (Mstar, Rc, alphainfall, taumax, frho, sig_mdl)."""
        Q = np.atleast_2d(f64(Q))
        if self._obs is None:
            raise SlamhipError('set_observations() first')
        if len(Q) == 0:
            return np.empty(0)
        return self.engine(Q[0, 1], incl, linewidth, rin, rout).lnprob_batch(Q)
