"""Drop-in `ChannelFit` whose likelihood runs on the MI355X.

Same constructor, `makegrid`, `fitting`, `cubemodel`, `modeltofits`
signatures and result attributes (`popt/plow/pmid/phigh` dicts, `chi2r`,
`<filename>.popt.txt`) as the reference class
(jinshisai/SLAM `channelfit/_channelfit.py:212-792`).  What changes:

* `makegrid` (host, numpy, once per fit) ends by uploading its constants to the
  GPU (`HipLikelihood`, libslamhip.so).
* `fitting` hands the sampler a BATCHED log-probability: one C-ABI call per
  half-ensemble instead of one Python `cubemodel` call per walker (:594-603).
* `cubemodel` is one C-ABI call (:499-537).

`scaling` is 'uniform' (least-squares amplitude, :533-536) or one of 'mom0ft',
'mom0clean', 'mom0model' (per-pixel moment-0 scaling, :343-371, :520-525): the
deconvolved moment-0 map is made on the host in `makegrid` (slam_amd/deconv.py)
and the reordered forward model runs on the GPU.  Anything else raises
ValueError rather than silently taking another path.
"""
from __future__ import annotations

import numpy as np

from .engine import PARAMKEYS, HipLikelihood
from .fitspace import FitSpace
from .utils import ReadFits, emcee_corner, rot

# astropy.constants / units values the reference reads at import (:29-32):
# CODATA 2018 G, IAU 2015 nominal GM_sun / G, the au in metres.
_G = 6.6743e-11
_MSUN = 1.3271244e20 / 6.6743e-11
_AU = 149597870700.0
vunit = np.sqrt(_G * _MSUN / _AU) * 1e-3

_UNITS = ['Msun', 'au', 'km/s', '', '', '', 'au', '', 'au', 'au', 'km/s',
          'deg', 'deg']
_DIGITS = [2, 0, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2]
_ILOG = np.array([0, 1, 7])          # Mstar, Rc, Ienv are sampled in log10 (:578)


SCALINGS = ('uniform', 'mom0ft', 'mom0clean', 'mom0model')


def makemom01(d, v, sigma):
    """Moment maps of a cube over its channel axis, as the reference's helper of
    the same name returns them (:40-50): `mom0` = sum(d) dv with dv the smallest
    channel step, `mom1` = sum(d v) / sum(d), blanked (NaN) where mom0 is under
    3 `sigma_mom0` = 3 sigma dv sqrt(channels), and `mom2` = sqrt(sum(d (v - mom1)^2))
    (the reference's second argument to np.sqrt is an output buffer, not a divisor)."""
    v = np.asarray(v)
    dv = np.diff(v).min()
    vcol = v.reshape(-1, 1, 1)
    with np.errstate(all='ignore'):
        total = np.sum(d, axis=0)
        mom0 = total * dv
        mom1 = np.sum(d * vcol, axis=0) / total
        mom2 = np.sqrt(np.sum(d * (vcol - mom1)**2, axis=0))
        sigma_mom0 = sigma * dv * np.sqrt(len(d))
        mom1[mom0 < 3 * sigma_mom0] = np.nan
    return dict(mom0=mom0, mom1=mom1, mom2=mom2, sigma_mom0=sigma_mom0)


class ChannelFit(ReadFits):

    def __init__(self, disk: bool = True, envelope: bool = False,
                 scaling: str = 'uniform', progressbar: bool = True,
                 precision: str = 'f64', beam_tol: float = 0.0):
        self.paramkeys = list(PARAMKEYS)
        self.precision = precision      # extension: 'f32' runs the likelihood in single precision
        # extension, opt-in: a rotated (non-separable) beam may be replaced in the batched
        # likelihood by a truncated factorisation, relative error ~beam_tol (slamhip_desc::beam_tol);
        # 0 = the reference's beam exactly
        self.beam_tol = float(beam_tol)
        self.disk = disk
        self.envelope = envelope
        self.scaling = scaling
        self.progressbar = progressbar
        self._engine = None
        self._max_scratch_bytes = 0
        if scaling not in SCALINGS:
            raise ValueError(f"scaling='{scaling}': expected one of {SCALINGS}")

    # -- synthetic / in-memory input (extension; the reference reads FITS only)
    def set_cube(self, x, y, v, data, bmaj, bmin, bpa, sigma):
        """Provide the cube as arrays instead of a FITS file; then call
        `makegrid(cubefits=None, ...)`.  x, y in au, v in km/s relative to vsys."""
        self.x, self.y, self.v = (np.asarray(a, dtype=float) for a in (x, y, v))
        self.dx, self.dy, self.dv = (self.x[1] - self.x[0], self.y[1] - self.y[0],
                                     self.v[1] - self.v[0])
        self.nx, self.ny, self.nv = len(self.x), len(self.y), len(self.v)
        self.data = np.asarray(data)
        self.bmaj, self.bmin, self.bpa = float(bmaj), float(bmin), float(bpa)
        self.sigma = float(sigma)
        self.offpix = (0, 0, 0)
        # a header for modeltofits: linear axes in the units the arrays are in
        self.header = {'NAXIS': 3, 'BUNIT': 'Jy/beam', 'BMAJ': self.bmaj, 'BMIN': self.bmin,
                       'BPA': self.bpa}
        for ax, (c, unit) in enumerate(((self.x, 'au'), (self.y, 'au'), (self.v, 'km/s')), 1):
            self.header.update({f'NAXIS{ax}': len(c), f'CRPIX{ax}': 1.0, f'CRVAL{ax}': float(c[0]),
                                f'CDELT{ax}': float(c[1] - c[0]), f'CUNIT{ax}': unit})

    def makegrid(self, cubefits: str | None = None,
                 pa: float = 0, incl: float = 90, dist: float = 1,
                 center: str | None = None, vsys: float = 0,
                 rmax: float = 1e4,
                 vlim: tuple = (-100, 0, 0, 100),
                 sigma: float | None = None, nlayer: int = 3,
                 xskip: int = 1, yskip: int = 1, skipto: int | None = False,
                 gaussmargin: float = 1.6,
                 savedeconvolved: str | None = None,
                 loaddeconvolved: str | None = None,
                 signmajor: int | None = None,
                 signminor: int | None = None,
                 max_scratch_bytes: int = 0):
        """One-time setup (:227-341), then upload to the GPU."""
        if cubefits is not None:
            self.read_cubefits(cubefits, center, dist, vsys, -rmax, rmax,
                               -rmax, rmax, None, None, xskip, yskip, sigma)
            if type(skipto) is int:                                  # :244-256
                dpix = min(abs(self.dx), abs(self.dy))
                iskip = int(self.bmin / (dpix / xskip) / skipto)
                if iskip == 0:
                    print("WARNING: 'skipto' is ignored because the beam minor"
                          " axis is smaller than 'skipto' pixels.")
                    iskip = 1
                self.read_cubefits(cubefits, center, dist, vsys, -rmax, rmax,
                                   -rmax, rmax, None, None, iskip, iskip, sigma)
                dpix = min(abs(self.dx), abs(self.dy))
                print(f'Adopt xskip={iskip:d} and yskip={iskip:d}.')
                print(f'Beam major/minor axis is {self.bmaj / dpix:.1f}'
                      f'/{self.bmin / dpix:.1f} pixels.')
        elif not hasattr(self, 'data'):
            raise ValueError('makegrid needs cubefits=... or a prior set_cube()')
        if sigma is None:
            sigma = self.sigma
        self.sigma = sigma
        dpix = min(abs(self.dx), abs(self.dy))
        v = self.v
        self.incl0 = incl
        self.pa_rad = np.radians(pa)
        self.cospa, self.sinpa = np.cos(self.pa_rad), np.sin(self.pa_rad)
        self.X, self.Y = np.meshgrid(self.x, self.y)

        # velocity windows (:265-274)
        in_blue = (vlim[0] <= v) * (v <= vlim[1])
        in_red = (vlim[2] <= v) * (v <= vlim[3])
        self.v_nanblue = v[v < vlim[0]]
        self.v_blue = v[in_blue]
        self.v_nanmid = v[(vlim[1] < v) * (v < vlim[2])]
        self.v_red = v[in_red]
        self.v_nanred = v[vlim[3] < v]
        self.v_valid = np.r_[self.v_blue, self.v_red]
        self.data_blue = self.data[in_blue]
        self.data_red = self.data[in_red]
        self.data_valid = np.append(self.data_blue, self.data_red, axis=0)
        print(f'{len(self.v_valid):d} channels will be fitted.')

        # rotation signs from the moment-1 map (:276-289)
        m = makemom01(self.data_valid, self.v_valid, sigma)
        self.mom0, self.mom1, self.mom2 = m['mom0'], m['mom1'], m['mom2']
        self.sigma_mom0 = m['sigma_mom0']
        Xr, Yr = rot(self.X, self.Y, self.pa_rad)
        self.signmajor = (np.sign(np.nansum(self.mom1 * Yr))
                          if signmajor is None else signmajor)
        self.signminor = (np.sign(np.nansum(self.mom1 * Xr)) * (-1)
                          if signminor is None else signminor)

        # nested grid on the disk plane (:293-320)
        r_need = rmax + gaussmargin * self.bmaj
        npix = int(2 * r_need / dpix + 0.5)
        npix = int(4 * np.ceil(npix / 4))
        self.npix = npix
        self.nq1 = npix // 2 - npix // 2 // 2
        self.nq3 = self.nq1 + npix // 2
        self.nlayer = nlayer
        half = npix // 2 - 0.5
        self.xnest = np.array([np.linspace(-half, half, npix) * dpix / 2**l
                               for l in range(nlayer)])
        self.ynest = self.xnest.copy()
        Xn = np.array([np.meshgrid(s, s)[0] for s in self.xnest])
        Yn = np.array([np.meshgrid(s, s)[1] for s in self.xnest])
        self.Xnest, self.Ynest = rot(Xn, Yn, self.pa_rad)
        self.Xnest0, self.Ynest0 = self.Xnest * 1, self.Ynest * 1
        print('-------- nested grid --------')
        for s in self.xnest:
            print(f'x, dx, npix: +/-{s[-1]:.2f}, {s[1] - s[0]:.2f} au, {npix:d}')
        print('-----------------------------')

        # beam kernel on the model grid (:329-335)
        ngauss = int(gaussmargin * self.bmaj / dpix + 0.5)
        t = (np.arange(2 * ngauss + 1) - ngauss) * dpix
        xb, yb = rot(*np.meshgrid(t, t), np.radians(self.bpa))
        self.gaussbeam = np.exp2(-4 * ((yb / self.bmaj)**2 + (xb / self.bmin)**2))
        self.pixperbeam = np.sum(self.gaussbeam)

        # crop (:337-341)
        n_need = int(r_need / dpix + 0.5)
        self.ineed0 = npix // 2 - n_need
        self.ineed1 = npix // 2 + n_need
        self.xneed = self.xnest[0][self.ineed0:self.ineed1]
        self.yneed = self.ynest[0][self.ineed0:self.ineed1]

        # per-pixel moment-0 scaling (:343-371): x axis of the beam into the
        # observed order, deconvolved moment-0 map by the chosen method
        self.mom0decon = None
        if 'mom0' in self.scaling:
            from . import deconv
            self.gaussbeam = self.gaussbeam[:, ::-1]
            if self.scaling == 'mom0clean':
                self.mom0decon = deconv.clean(
                    data=self.mom0, beam=self.gaussbeam, sigma=self.sigma_mom0,
                    savetxt=savedeconvolved, loadtxt=loaddeconvolved,
                    verbose=self.progressbar)
            elif self.scaling == 'mom0model':
                d = deconv.modeldeconvolve(
                    x=self.x, y=self.y, data=self.mom0, bmaj=self.bmaj,
                    bmin=self.bmin, bpa=self.bpa, sigma=self.sigma_mom0,
                    savetxt=savedeconvolved, loadtxt=loaddeconvolved,
                    progressbar=self.progressbar)
                self.mom0decon, self.xdecon, self.ydecon, self.zdecon = d
                print('Found a deconvolved solution.')
            else:
                self.mom0decon = deconv.ftdeconvolve(
                    x=self.x, y=self.y, data=self.mom0, bmaj=self.bmaj,
                    bmin=self.bmin, bpa=self.bpa, sigma=self.sigma_mom0,
                    threshold=3)
                print('Divided in the Fourier space.')
            from scipy.signal import convolve
            self.resdecon = self.mom0 - convolve(self.mom0decon, self.gaussbeam,
                                                 mode='same')
            maxres = np.max(self.resdecon) / self.sigma_mom0
            rmsres = np.sqrt(np.mean(self.resdecon**2)) / self.sigma_mom0
            print(f'Max and rms are {maxres:.1f}sigma '
                  + f'and {rmsres:.1f}sigma in Moment 0 residual.')

        # the constants go to the GPU on first use of the likelihood
        self.close()
        self._max_scratch_bytes = max_scratch_bytes

    @property
    def engine(self):
        """The GPU-resident likelihood (created on first use; there is no CPU
        fallback: without libslamhip.so or a GPU this raises SlamhipError)."""
        if self._engine is None:
            if not hasattr(self, 'data_valid'):
                raise RuntimeError('call makegrid() first')
            self._engine = HipLikelihood(self.setup_dict(),
                                         max_scratch_bytes=self._max_scratch_bytes,
                                         precision=self.precision, beam_tol=self.beam_tol)
        return self._engine

    @engine.setter
    def engine(self, value):
        self._engine = value

    def close(self):
        """Release the GPU context."""
        if self._engine is not None and hasattr(self._engine, 'close'):
            self._engine.close()
        self._engine = None

    def setup_dict(self):
        """The constants libslamhip.so needs (slamhip_desc)."""
        keys = ('nlayer', 'npix', 'nq1', 'nq3', 'ineed0', 'ineed1', 'dv',
                'sigma', 'pixperbeam', 'incl0', 'signmajor', 'signminor',
                'disk', 'envelope', 'Xnest0', 'Ynest0', 'xneed', 'yneed', 'x',
                'y', 'v_valid', 'data_valid', 'gaussbeam')
        d = {k: getattr(self, k) for k in keys}
        d['vunit'] = vunit
        d['mom0decon'] = getattr(self, 'mom0decon', None)
        return d

    # -- forward model -------------------------------------------------------
    def _natural(self, Mstar, Rc, cs, h1, h2, pI, Rin, Ienv, xoff, yoff, voff,
                 incloff, paoff):
        return np.array([Mstar, Rc, cs, h1, h2, pI, Rin, Ienv, xoff, yoff,
                         voff, incloff, paoff], dtype=float)

    def cubemodel(self, Mstar: float, Rc: float, cs: float,
                  h1: float = 0, h2: float = -1, pI: float = 0,
                  Rin: float = 0, Ienv: float = 0,
                  xoff: float = 0, yoff: float = 0, voff: float = 0,
                  incloff: float = 90, paoff: float = 0,
                  convolving: bool = True):
        """Model cube (nv_valid, ny, nx), least-squares scaled (:499-537)."""
        p = self._natural(Mstar, Rc, cs, h1, h2, pI, Rin, Ienv, xoff, yoff,
                          voff, incloff, paoff)
        return self.engine.cubemodel(p, convolving=convolving)

    def modelmom(self, mode: str = 'mod', **kwargs):
        """(mom0, mom1) that the reference's `plotmom(mode)` draws (:720-737):
        'obs' the observed maps of `makegrid`, 'mod' the model's for `self.popt`
        (or the parameters given, which then become `self.popt` as there), 'res'
        observed minus model.  The model cube is built and reduced over its
        channels on the GPU (`slamhip_moments`); the figure itself is out of
        this package's scope."""
        if 'obs' in mode:
            return self.mom0, self.mom1
        if not ('mod' in mode or 'res' in mode):
            raise ValueError("mode must contain 'obs', 'mod' or 'res'")
        if kwargs != {}:
            self.popt = kwargs
        p = self._natural(**self.popt)
        dv = np.diff(np.asarray(self.v_valid)).min()
        mom0, mom1 = self.engine.moments(p, dv, 3 * self.sigma_mom0)
        if 'res' in mode:
            return self.mom0 - mom0, self.mom1 - mom1
        return mom0, mom1

    def chi2_batch(self, P):
        """chi2 (:601-602) for natural-unit parameter rows P[nw, 13]."""
        return self.engine.chi2_batch(P)

    # -- fit driver ----------------------------------------------------------
    def fitting(self, Mstar_range: list = [0.01, 10],
                Rc_range: list = [1, 1000],
                cs_range: list = [0.01, 1],
                h1_range: list = [0.01, 1],
                h2_range: list = [0.01, 1],
                pI_range: list = [-2, 2],
                Rin_range: list = [0, 1000],
                Ienv_range: list = [0.01, 100],
                xoff_range: list = [-100, 100],
                yoff_range: list = [-100, 100],
                voff_range: list = [-0.2, 0.2],
                incl_range: list = [-45, 45],
                pa_range: list = [-45, 45],
                fixed_params: dict = {},
                filename: str = 'channelfit',
                show: bool = False,
                kwargs_emcee_corner: dict = {}):
        """Free/fixed bookkeeping, log10 sampling of Mstar/Rc/Ienv, sampler
        call, result table (:539-663).  The sampler receives a vectorised
        log-probability: p[nw, ndim] -> lnp[nw]."""
        space = FitSpace(self.paramkeys, _ILOG, fixed_params,
                         [Mstar_range, Rc_range, cs_range, h1_range, h2_range, pI_range,
                          Rin_range, Ienv_range, xoff_range, yoff_range, voff_range,
                          incl_range, pa_range])
        self.free = space.free
        if space.any_free:
            kw = {'nwalkers_per_ndim': 16, 'nburnin': 200, 'nsteps': 500,
                  'labels': space.labels(self.paramkeys), 'rangelevel': None,
                  'range_corner': None, 'figname': filename + '.corner.png',
                  'show_corner': show}
            kw.update(kwargs_emcee_corner)
            kw['range_corner'] = space.corner_ranges(kw['range_corner'])     # :611-616
            bar = None
            if self.progressbar:
                from tqdm import tqdm
                bar = tqdm(total=kw['nwalkers_per_ndim'] * space.ndim
                           * (kw['nburnin'] + kw['nsteps'] + 2))
                bar.set_description('Within the ranges')

            def lnprob(P):
                """sampled [nw, ndim] (or [ndim]) -> -chi2/2 per row (:597-603)"""
                Q = space.to_natural(P)
                if bar is not None:
                    bar.update(len(Q))
                lnp = -0.5 * self.engine.chi2_batch(Q)
                return lnp[0] if np.ndim(P) == 1 else lnp

            self.lnprob, self.to_natural, self.plim = lnprob, space.to_natural, space.plim
            mcmc = emcee_corner(space.plim, lnprob, simpleoutput=False, vectorize=True, **kw)
            if bar is not None:
                bar.close()
            self.popt, self.plow, self.pmid, self.phigh = (space.to_natural(m)[0] for m in mcmc)
        else:
            chi2 = self.engine.chi2_batch(space.base[None])[0]         # :629-634
            dof = np.prod(np.shape(self.data_valid)) / self.pixperbeam - 6 - 1
            self.chi2r = chi2 / dof
            self.popt = self.plow = self.pmid = self.phigh = space.base

        self.pa_rad = self.pa_rad + np.radians(self.popt[12])           # :645-647
        self.sinpa, self.cospa = np.sin(self.pa_rad), np.cos(self.pa_rad)
        space.report(self.plow, self.popt, self.phigh, _DIGITS, _UNITS)
        if space.any_free:
            space.write_table(filename + '.popt.txt', self.popt, self.plow, self.pmid, self.phigh)
        for name in ('popt', 'plow', 'pmid', 'phigh'):
            setattr(self, name, dict(zip(self.paramkeys, getattr(self, name))))

    # -- outputs -------------------------------------------------------------
    def _full_velocity_cube(self, m):
        """Valid-channel cube -> full velocity axis, NaN where not fitted
        (`concat` in :682-705)."""
        full = np.full((len(self.v), len(self.y), len(self.x)), np.nan)
        vv = np.asarray(self.v)
        for k, vk in enumerate(self.v_valid):
            full[np.where(vv == vk)[0][0]] = m[k]
        return full

    def modeltofits(self, filehead: str = 'best', **kwargs):
        """Write model / residual / unconvolved cubes as FITS (:665-718)."""
        from .fitsio import write_cube
        p = self.popt if kwargs == {} else kwargs
        for k in self.free.keys():
            self.free[k] = True
        m = self.cubemodel(**p)
        m0 = self.cubemodel(**p, convolving=False)
        h = dict(self.header)
        h['NAXIS1'], h['NAXIS2'], h['NAXIS3'] = len(self.x), len(self.y), len(self.v)
        for ax in (1, 2, 3):
            h[f'CRPIX{ax}'] = h[f'CRPIX{ax}'] - self.offpix[ax - 1]
        model = self._full_velocity_cube(m)
        write_cube(f'{filehead}.model.fits', model, h)
        write_cube(f'{filehead}.residual.fits', self.data - model, h)
        write_cube(f'{filehead}.beforeconvolving.fits',
                   self._full_velocity_cube(m0), h)
