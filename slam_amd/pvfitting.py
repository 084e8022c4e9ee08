"""PVFitting: drop-in for the reference's `pvfitting.PVFitting`
(pvfitting/_pvfitting.py) whose model evaluations run on the MI355X.

Same methods and arguments for the fitting path (put_PV, check_modelgrid,
fit_mockpvd, read_fitres, modeltofits).  Figures are out of this package's scope:
`plot`, `show`, `title`, `log` are accepted, and a call that asks for a figure
(`plot=True`, the reference's default) says once that none is drawn -- `makemodel`
returns the model diagrams and `modeltofits` writes them.  The sampler receives a
vectorised log-probability (p[nw, ndim] -> lnp[nw]) that calls libslampv.so once
per ensemble half.
"""
from __future__ import annotations

import numpy as np

from . import fitsio
from .mockpvd import PARAMKEYS, MockPVD
from .fitspace import FitSpace
from .utils import ReadFits, emcee_corner

_ILOG = np.array([0, 1, 2, 3, 4], dtype=int)        # _pvfitting.py:162


def getquad(m):
    """+1 / -1: which diagonal pair of quadrants of a PV diagram holds more
    flux (the rotation / infall sign the reference derives at
    _pvfitting.py:462-469)."""
    m = np.asarray(m)
    kv, kx = m.shape[0] // 2, m.shape[1] // 2
    lowleft, upright = np.sum(m[:kv, :kx]), np.sum(m[kv:, kx:])
    upleft, lowright = np.sum(m[kv:, :kx]), np.sum(m[:kv, kx:])
    return int(np.sign(lowleft + upright - upleft - lowright))


class PVFitting(ReadFits):

    def put_PV(self, pvmajorfits: str, pvminorfits: str,
               dist: float, vsys: float, rmax: float | None,
               vmin: float | None, vmax: float | None, sigma: float | None,
               xskip: int = 1, skipto: int | None = None,
               dNsampling: list | None = [5, 1]):
        """Read the major- and minor-axis PV diagrams (same arguments as the
        reference's, _pvfitting.py:31-61).  Each file is read twice: first over
        the whole velocity axis (kept as `vorg`, and to turn `skipto` = pixels
        per beam minor axis into a decimation factor), then decimated and
        cropped to [vmin, vmax]."""
        diagrams = []
        for path in (pvmajorfits, pvminorfits):
            common = dict(pvfits=path, dist=dist, vsys=vsys, xmin=-rmax, xmax=rmax, sigma=sigma)
            self.read_pvfits(xskip=xskip, **common)
            step = xskip
            if type(skipto) is int:
                pix = np.abs(self.dx) / xskip          # undecimated pixel [au]
                step = int(self.bmin / pix / skipto)
                if step == 0:
                    step = 1
                    print("WARNING: 'skipto' is ignored because the beam minor axis is "
                          "smaller than 'skipto' pixels.")
                else:
                    print(f'Adopt xskip={step:d}.')
                    print(f'Beam major/minor axis is {self.bmaj / (pix * xskip * step):.1f}'
                          f'/{self.bmin / (pix * xskip * step):.1f} pixels.')
            self.vorg = self.v
            self.read_pvfits(xskip=step, vmin=vmin, vmax=vmax, **common)
            diagrams.append(self.data)
        self.dpvmajor, self.dpvminor = diagrams

    def check_modelgrid(self, nsubgrid: float = 1, n_nest: list | None = None,
                        reslim: float = 5):
        mpvd = MockPVD(self.x, self.x, self.v, nsubgrid=nsubgrid, nnest=n_nest,
                       beam=self.beam, reslim=reslim)
        mpvd.gridinfo()

    def fit_mockpvd(self, incl: float = 89.,
                    Mstar_range: list = [0.01, 10],
                    Rc_range: list = [1., 1000.],
                    alphainfall_range: list = [0.0, 1],
                    taumax_range: list = [0.1, 1e3],
                    frho_range: list = [1., 1e4],
                    sig_mdl_range: list = [0., 10.],
                    fixed_params: dict = {'Mstar': None, 'Rc': None,
                                          'alphainfall': None, 'taumax': None,
                                          'frho': None, 'sig_mdl': None},
                    vmask: list = [0, 0],
                    zmax: float | None = None,
                    filename: str = 'PVsilhouette',
                    show: bool = False, progressbar: bool = True,
                    kwargs_emcee_corner: dict = {},
                    signmajor: int | None = None, signminor: int | None = None,
                    pa_major: float = 0., pa_minor: float = 90.,
                    linewidth: float | None = None,
                    nsubgrid: int = 1,
                    n_nest: list = [2, 2, 2, 2, 2, 2],
                    reslim: float = 10,
                    title: str | None = None,
                    log: bool = False, plot: bool = True):
        """Fit the two PV diagrams with the mock-PV model (_pvfitting.py:87-265)."""
        majobs = np.array(self.dpvmajor, dtype=float)
        minobs = np.array(self.dpvminor, dtype=float)
        beam_area = np.pi / (4. * np.log(2.)) * self.bmaj * self.bmin
        Rarea = beam_area / self.dx / self.dx
        x, v = np.meshgrid(self.x, self.v)
        mask = (vmask[0] < v) * (v < vmask[1])
        majobs = np.where(mask, np.nan, majobs)
        minobs = np.where(mask, np.nan, minobs)
        majquad = getquad(self.dpvmajor) if signmajor is None else signmajor
        minquad = getquad(self.dpvminor) * (-1) if signminor is None else signminor
        if zmax is None:
            z = self.x
        else:
            dx = self.x[1] - self.x[0]
            nz = int(zmax / dx + 0.5)
            z = (np.arange(2 * nz + 1) - nz) * dx
        mpvd = MockPVD(self.x, z, self.v, nsubgrid=nsubgrid, nnest=n_nest,
                       beam=self.beam, reslim=reslim, signmajor=majquad,
                       signminor=minquad, pa_major=pa_major, pa_minor=pa_minor)
        mpvd.set_observations(majobs, minobs, self.sigma, Rarea)
        self.mpvd = mpvd
        rout = np.max(z)
        self._fitargs = dict(incl=incl, linewidth=linewidth, rin=1., rout=rout)

        def stats_of(Q):
            Q = np.atleast_2d(np.asarray(Q, dtype=float))
            eng = mpvd.engine(Q[0, 1], incl, linewidth, 1., rout)
            return eng, eng.stats(Q)

        def makemodel(Mstar, Rc, alphainfall, taumax, frho):
            q = [Mstar, Rc, alphainfall, taumax, frho, 0.]
            eng, st = stats_of(q)
            major, minor = eng.mockpvd(q)
            return st['fflux'][0] * major, st['fflux'][0] * minor
        self.makemodel = makemodel

        paramkeys = list(PARAMKEYS)
        space = FitSpace(paramkeys, _ILOG, fixed_params,
                         [Mstar_range, Rc_range, alphainfall_range, taumax_range, frho_range,
                          sig_mdl_range])
        if space.any_free:
            names = ['Mstar', 'Rc', r'$\alpha$', r'$\tau_\mathrm{max}$', r'$f_\rho$',
                     r'$\sigma_\mathrm{model}$']
            kw = {'nwalkers_per_ndim': 4, 'nburnin': 500, 'nsteps': 500,
                  'rangelevel': None, 'range_corner': None, 'labels': space.labels(names),
                  'figname': filename + '.corner.png', 'show_corner': show,
                  'plot_chain': False, 'show_chain': False}
            kw.update(kwargs_emcee_corner)
            kw['range_corner'] = space.corner_ranges(kw['range_corner'])      # :201-206
            bar = None
            if progressbar:
                from tqdm import tqdm
                bar = tqdm(total=kw['nwalkers_per_ndim'] * space.ndim
                           * (kw['nburnin'] + kw['nsteps'] + 2))
                bar.set_description('Within the ranges')

            def lnprob(P):
                """sampled [nw, ndim] (or [ndim]) -> modified log-likelihood per row (:181-195)"""
                Q = space.to_natural(P)
                if bar is not None:
                    bar.update(len(Q))
                lnp = mpvd.lnprob_batch(Q, incl, linewidth, 1., rout)
                return lnp[0] if np.ndim(P) == 1 else lnp
            self.lnprob, self.to_natural, self.plim = lnprob, space.to_natural, space.plim
            mcmc = emcee_corner(space.plim, lnprob, simpleoutput=False, vectorize=True, **kw)
            if bar is not None:
                bar.close()
            popt, plow, pmid, phigh = (space.to_natural(m)[0] for m in mcmc)
        else:
            _, st = stats_of(space.base)                             # :221-236
            dof = np.prod(np.shape(majobs)) + np.prod(np.shape(minobs))
            self.chi2r = st['chi2'][0] / (dof / np.sqrt(Rarea) - 6 - 1)
            popt = plow = pmid = phigh = space.base
        space.report(plow, popt, phigh, digits=[2, 0, 2, 2, 2, 2],
                     units=['Msun', 'au', '', '', '', 'sig_obs'],
                     formats=['f', 'f', 'f', 'e', 'e', 'f'])
        if space.any_free:
            space.write_table(filename + '.popt.txt', popt, plow, pmid, phigh)   # :253-257
        self.popt, self.plow, self.pmid, self.phigh = (dict(zip(paramkeys, v))
                                                       for v in (popt, plow, pmid, phigh))
        if plot:
            # the reference ends with plot_pvds (:264-265); figures are not drawn here
            print('slam_amd.PVFitting.fit_mockpvd: no figure is drawn (plot / show / title / log '
                  'ignored); makemodel(**popt) returns the diagrams, modeltofits() writes them.')

    def modeltofits(self, filehead: str = 'best', **kwargs):
        """Write the model and residual PV diagrams of the best-fit (or given)
        parameters as `<filehead>.{model,residual}.{major,minor}.fits`, padded
        with NaN rows back to the file's full velocity axis (_pvfitting.py:426-459).
        Data only, written with this package's FITS writer.  Unlike the
        reference, repeated calls do not keep shifting CRPIX1 (it edits the
        stored header in place; here a copy is edited)."""
        p = dict(self.popt if kwargs == {} else kwargs)
        p.pop('sig_mdl', None)
        majmod, minmod = self.makemodel(**p)
        h = dict(self.header)
        nx = len(self.x)
        h['CRPIX1'] = h['CRPIX1'] - self.offpix[0]
        blue = np.full((int(np.sum(self.vorg < np.min(self.v))), nx), np.nan)
        red = np.full((int(np.sum(np.max(self.v) < self.vorg)), nx), np.nan)
        out = {'model.major': majmod, 'model.minor': minmod,
               'residual.major': self.dpvmajor - majmod,
               'residual.minor': self.dpvminor - minmod}
        for ext, d in out.items():
            fitsio.write_cube(f'{filehead}.{ext}.fits', np.concatenate((blue, d, red), axis=0), h)

    def read_fitres(self, f: str):
        """Read a <filename>.popt.txt table (_pvfitting.py:267-275)."""
        self.popt, self.plow, self.pmid, self.phigh = np.loadtxt(f).T
