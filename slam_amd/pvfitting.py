"""PVFitting: drop-in for the reference's `pvfitting.PVFitting`
(pvfitting/_pvfitting.py) whose model evaluations run on the MI355X.

Same methods and arguments (put_PV, check_modelgrid, fit_mockpvd, read_fitres,
plot_pvds, modeltofits); the sampler receives a vectorised log-probability
(p[nw, ndim] -> lnp[nw]) that calls libslampv.so once per ensemble half.
"""
from __future__ import annotations

import numpy as np

from .mockpvd import PARAMKEYS, MockPVD
from .utils import ReadFits, emcee_corner

_ILOG = np.array([0, 1, 2, 3, 4], dtype=int)        # _pvfitting.py:162


def getquad(m):
    """Sign of the PV diagram's dominant quadrant pair (_pvfitting.py:462-469)."""
    nv, nx = np.shape(m)
    q = np.sum(m[:nv // 2, :nx // 2]) + np.sum(m[nv // 2:, nx // 2:]) \
        - np.sum(m[nv // 2:, :nx // 2]) - np.sum(m[:nv // 2, nx // 2:])
    return int(np.sign(q))


class PVFitting(ReadFits):

    def put_PV(self, pvmajorfits: str, pvminorfits: str,
               dist: float, vsys: float, rmax: float | None,
               vmin: float | None, vmax: float | None, sigma: float | None,
               xskip: int = 1, skipto: int | None = None,
               dNsampling: list | None = [5, 1]):
        """Read the two observed PV diagrams (_pvfitting.py:31-61)."""
        d = []
        for pvfits in [pvmajorfits, pvminorfits]:
            self.read_pvfits(pvfits=pvfits, dist=dist, vsys=vsys,
                             xmin=-rmax, xmax=rmax, xskip=xskip, sigma=sigma)
            if type(skipto) is int:
                iskip = int(self.bmin / (np.abs(self.dx) / xskip) / skipto)
                if iskip == 0:
                    print('WARNING: \'skipto\' is ignored because the beam minor '
                          'axis is smaller than \'skipto\' pixels.')
                    iskip = 1
                else:
                    print(f'Adopt xskip={iskip:d}.')
                    ibmaj = self.bmaj / (np.abs(self.dx) * iskip)
                    ibmin = self.bmin / (np.abs(self.dx) * iskip)
                    print(f'Beam major/minor axis is {ibmaj:.1f}/{ibmin:.1f} pixels.')
            else:
                iskip = xskip
            self.vorg = self.v
            self.read_pvfits(pvfits=pvfits, dist=dist, vsys=vsys,
                             xmin=-rmax, xmax=rmax, xskip=iskip, sigma=sigma,
                             vmin=vmin, vmax=vmax)
            d.append(self.data)
        self.dpvmajor, self.dpvminor = d

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
        fixed = [fixed_params[k] if k in fixed_params else None for k in paramkeys]
        free = {k: f is None for k, f in zip(paramkeys, fixed)}
        notfixed = np.array([f is None for f in fixed])
        base = np.array([np.nan if f is None else float(f) for f in fixed])
        runfit = bool(notfixed.any())
        if runfit:
            logfixed = _ILOG[~notfixed[_ILOG]]
            with np.errstate(all='ignore'):
                base[logfixed] = np.log10(base[logfixed])            # :163-164
            labels = np.array(['Mstar', 'Rc', r'$\alpha$', r'$\tau_\mathrm{max}$',
                               r'$f_\rho$', r'$\sigma_\mathrm{model}$'])
            labels[_ILOG] = ['log' + labels[i] for i in _ILOG]
            labels = labels[notfixed]
            kw = {'nwalkers_per_ndim': 4, 'nburnin': 500, 'nsteps': 500,
                  'rangelevel': None, 'range_corner': None, 'labels': labels,
                  'figname': filename + '.corner.png', 'show_corner': show,
                  'plot_chain': True, 'show_chain': show}
            kw.update(kwargs_emcee_corner)
            bar = None
            if progressbar:
                from tqdm import tqdm
                total = kw['nwalkers_per_ndim'] * int(notfixed.sum())
                total *= kw['nburnin'] + kw['nsteps'] + 2
                bar = tqdm(total=total)
                bar.set_description('Within the ranges')

            def to_natural(P):
                """sampled [nw, ndim] -> natural 6-vectors (:185-187)"""
                P = np.atleast_2d(np.asarray(P, dtype=float))
                Q = np.tile(base, (len(P), 1))
                Q[:, notfixed] = P
                Q[:, _ILOG] = 10**Q[:, _ILOG]
                return Q

            def lnprob(P):
                single = np.ndim(P) == 1
                Q = to_natural(P)
                if bar is not None:
                    bar.update(len(Q))
                lnp = mpvd.lnprob_batch(Q, incl, linewidth, 1., rout)   # :189-195
                return lnp[0] if single else lnp
            self.lnprob, self.to_natural = lnprob, to_natural
            plim = np.array([Mstar_range, Rc_range, alphainfall_range,
                             taumax_range, frho_range, sig_mdl_range], dtype=float)
            with np.errstate(all='ignore'):
                plim[_ILOG] = np.log10(plim[_ILOG])
            plim = plim[notfixed].T
            if type(r_c := kw['range_corner']) is dict:              # :201-206
                r_c = [r_c[k] if k in r_c else 0.8 for k in paramkeys]
                for i in _ILOG:
                    r_c[i] = r_c[i] if type(r_c[i]) is float else np.log10(r_c[i])
                kw['range_corner'] = [a for a, k in zip(r_c, paramkeys) if free[k]]
            self.plim = plim
            mcmc = emcee_corner(plim, lnprob, simpleoutput=False, vectorize=True, **kw)
            if bar is not None:
                bar.close()
            popt, plow, pmid, phigh = (to_natural(m)[0] for m in mcmc)
        else:
            _, st = stats_of(base)                                   # :221-236
            dof = np.prod(np.shape(majobs)) + np.prod(np.shape(minobs))
            dof = dof / np.sqrt(Rarea) - 6 - 1
            self.chi2r = st['chi2'][0] / dof
            popt = plow = pmid = phigh = base
        self.popt, self.plow, self.pmid, self.phigh = popt, plow, pmid, phigh
        ulist = ['Msun', 'au', '', '', '', 'sig_obs']
        digits = [2, 0, 2, 2, 2, 2]
        flist = ['f', 'f', 'f', 'e', 'e', 'f']
        for i, (k, d, u, f) in enumerate(zip(paramkeys, digits, ulist, flist)):
            p = [self.plow[i], self.popt[i], self.phigh[i]]
            print(f'{k} = {p[0]:.{d:d}{f}}, {p[1]:.{d:d}{f}}, {p[2]:.{d:d}{f}} {u}')
        if runfit:
            with open(filename + '.popt.txt', 'w') as f:             # :253-257
                f.write('#Rows:' + ','.join(paramkeys) + '\n')
                f.write('#Columns:' + ','.join(['popt', 'plow', 'pmid', 'phigh']) + '\n')
                np.savetxt(f, np.transpose([self.popt, self.plow, self.pmid, self.phigh]))
        self.popt = dict(zip(paramkeys, self.popt))
        self.plow = dict(zip(paramkeys, self.plow))
        self.pmid = dict(zip(paramkeys, self.pmid))
        self.phigh = dict(zip(paramkeys, self.phigh))
        if plot:
            self.plot_pvds(filename=filename, color='model', contour='obs',
                           vmask=vmask, title=title, show=show, log=log)

    def read_fitres(self, f: str):
        """Read a <filename>.popt.txt table (_pvfitting.py:267-275)."""
        self.popt, self.plow, self.pmid, self.phigh = np.loadtxt(f).T

    def plot_pvds(self, filename: str = 'PVfitting', color: str = 'model',
                  contour: str = 'obs', vmask: list = [0., 0.],
                  cmap: str = 'viridis', cmap_residual: str = 'bwr',
                  ext: str = '.png', title: str | None = None,
                  show: bool = False, shadecolor: str = 'white',
                  clevels: list | None = None, log: bool = False):
        """Observed vs model PV diagrams and residuals (_pvfitting.py:277-424)."""
        import matplotlib.pyplot as plt
        majobs = np.array(self.dpvmajor, dtype=float)
        minobs = np.array(self.dpvminor, dtype=float)
        x, v = np.meshgrid(self.x, self.v)
        mask = (vmask[0] < v) * (v < vmask[1])
        use_model = 'model' in [color, contour]
        if use_model:
            if not hasattr(self, 'popt'):
                print('ERROR\tplot_pvds: No optimized parameters are found.')
                print('ERROR\tplot_pvds: Run fitting or read fitting result first.')
                return 0
            popt = np.array(list(self.popt.values())) if isinstance(self.popt, dict) \
                else np.asarray(self.popt)
            majmod, minmod = self.makemodel(*popt[:-1])
            majres = np.where(mask, -(mask.astype('int')), (majobs - majmod) / self.sigma)
            minres = np.where(mask, -(mask.astype('int')), (minobs - minmod) / self.sigma)
        outlabel = '.model' if use_model else '.obs'
        if clevels is None:
            clevels = (2**np.arange(0, 10) if log else np.arange(1, 11)) * 3 * self.sigma

        def panel(colors, contours, cm, vmin, vmax, shade, alpha, residual):
            fig, axes = plt.subplots(1, 2)
            fig.set_figheight(3.2)
            fig.set_figwidth(5)
            if log and not residual:
                lo = np.log10(2 * self.sigma) if vmin is None else np.log10(vmin)
                hi = None if vmax is None else np.log10(vmax)
            elif residual:
                top = max(abs(vmin), abs(vmax))
                lo, hi = -top, top
            else:
                lo, hi = vmin, vmax
            for ax, dc, dk, xl in zip(axes, colors, contours,
                                      ('Major offset (au)', 'Minor offset (au)')):
                ax.tick_params('x', rotation=30)
                d = dc * 1
                if log and not residual:
                    d = np.log10(d.clip(vmin, None))
                im = ax.pcolormesh(self.x, self.v, d, cmap=cm, vmin=lo, vmax=hi,
                                   alpha=alpha, rasterized=True)
                ax.contour(self.x, self.v, dk, levels=clevels, colors='k')
                ax.set_xlabel(xl)
                ax.set_xlim(np.min(self.x), np.max(self.x))
                ax.set_ylim(np.min(self.v), np.max(self.v))
            axes[0].set_ylabel(r'$V-V_{\rm sys}$ (km s$^{-1}$)')
            axes[1].set_yticklabels('')
            cax = axes[1].inset_axes([1.02, 0., 0.05, 1.])
            plt.colorbar(im, cax=cax, label=r'Noise level ($\sigma$)' if residual
                         else r'Jy beam$^{-1}$')
            if shade is not None:
                _v = self.v[(self.v > shade[0]) * (self.v < shade[1])]
                if len(_v):
                    for ax in axes:
                        ax.fill_between(self.x,
                                        np.full(len(self.x), np.min(_v) - 0.5 * self.dv),
                                        np.full(len(self.x), np.max(_v) + 0.5 * self.dv),
                                        color=shadecolor, alpha=0.6, edgecolor=None)
            if title is not None:
                fig.suptitle(title, y=0.92)
            fig.tight_layout()
            fig.subplots_adjust(wspace=0.1)
            return fig

        d_col = [majmod, minmod] if color == 'model' else [majobs, minobs]
        d_con = [majmod, minmod] if contour == 'model' else [majobs, minobs]
        vmin = 2 * self.sigma if log else np.nanmin(np.array(d_col))
        vmax = np.nanmax(np.array(d_col))
        shade = vmask if vmask[1] > vmask[0] else None
        figs = [panel(d_col, d_con, cmap, vmin, vmax, shade, 0.8, False)]
        figs[0].savefig(filename + outlabel + ext, dpi=300)
        if show:
            plt.show()
        if use_model:
            d_col = [majres, minres] if color == 'model' else [majobs, minobs]
            d_con = [majres, minres] if contour == 'model' else [majobs, minobs]
            vmin, vmax = np.nanmin(np.array(d_col)), np.nanmax(np.array(d_col))
            figs.append(panel(d_col, d_con, cmap_residual, vmin, vmax, None, 0.5, True))
            figs[1].savefig(filename + '.residual' + ext, dpi=300)
            if show:
                plt.show()
        for f in figs:
            plt.close(f)
        return figs

    def modeltofits(self, filehead: str = 'best', **kwargs):
        """Write model and residual PV diagrams on the full velocity axis
        (_pvfitting.py:426-459)."""
        from .fitsio import write_cube
        h = dict(self.header)
        h['CRPIX1'] = h['CRPIX1'] - self.offpix[0]
        nx = len(self.x)
        p = dict(self.popt if kwargs == {} else kwargs)
        p.pop('sig_mdl', None)
        majmod, minmod = self.makemodel(**p)
        majres = self.dpvmajor - majmod
        minres = self.dpvminor - minmod
        nblue = int(np.sum(self.vorg < np.min(self.v)))
        nred = int(np.sum(np.max(self.v) < self.vorg))
        pad = lambda d: np.concatenate((np.full((nblue, nx), np.nan), d,      # noqa: E731
                                        np.full((nred, nx), np.nan)), axis=0)
        for d, ext in ((majmod, 'model.major'), (minmod, 'model.minor'),
                       (majres, 'residual.major'), (minres, 'residual.minor')):
            write_cube(f'{filehead}.{ext}.fits', pad(d), h)
