"""Host-side, once-per-fit setup of the mock-PV model: the UCM envelope + disk
lookup tables, the nested 3-D grid and the per-cell constants the GPU kernels
read (include/slampv.h).  Plain numpy; nothing here runs per walker.

Every number is produced with the reference's own float64 arithmetic
(pvfitting/precalculation.py, pvfitting/grid.py, pvfitting/mockpvd.py:64-95,
168-219), so the constants are bit-equal to what the reference caches on its
first evaluation (tests/test_pv_host.py checks them against the golden vectors).
The layout differs: each level is kept as its FULL box, z-slowest / x-fastest,
with the cells that cannot contribute flagged instead of removed.
"""
from __future__ import annotations

import functools

import numpy as np

# astropy's CODATA 2018 / IAU 2015 values behind mockpvd.py:11-13
G_SI = 6.6743e-11
M_SUN = 1.3271244e20 / G_SI
AU_M = 149597870700.0

H0, PLH, PLS = 0.2, 0.25, 1.0          # precalculation.py:10 defaults
NR = 1600                              # precalculation.py:146


class EnvelopeDiskTables:
    """(Ntheta, Nr) tables on a (theta, ln r) mesh, radii in units of Rc
    (precalculation.py:146-159)."""

    def __init__(self):
        lnr = np.linspace(np.log(1e-4), np.log(1e4), NR)
        self.nr = NR
        self.lnr0 = lnr[0]
        self.dlnr = np.exp(lnr[1] - lnr[0]) - 1            # :149 (sic)
        self.ntheta = int(np.pi / self.dlnr + 0.5)
        theta = np.linspace(0, np.pi, self.ntheta)
        self.theta0 = theta[0]
        self.dtheta = theta[1] - theta[0]
        lnr_m, th = np.meshgrid(lnr, theta)
        self.r = np.exp(lnr_m)
        c = np.cos(th)
        self.sign_mu = np.sign(c)
        self.mu = np.abs(c)
        self.sin_theta = np.sin(th).clip(1e-10, None)
        R = self.r * self.sin_theta
        self.zcyl = self.r * self.mu
        self.H = H0 * R**(1. + PLH)
        self.in_disk = (R < 1) * (self.zcyl < 3 * self.H)
        vr, vt, vp, self.rho_env = self._envelope()
        vp_disk, self.rho_disk = self._disk()
        self.vr_env, self.vt_env = vr, vt
        self.vp_all = vp + vp_disk
        for name in ('r', 'sign_mu', 'mu', 'sin_theta', 'zcyl', 'H', 'in_disk'):
            delattr(self, name)

    def _origin(self, mu):
        """cos(theta0) of the streamline through (r, mu)  (:50-70)."""
        r = self.r
        p = (r - 1) / 3.
        q = mu * r / 2.
        D = q**2 + p**3
        m0 = np.full_like(r, np.nan)
        sel = r >= 1
        qD = q[sel] * (D[sel])**(-1 / 2)
        m0[sel] = D[sel]**(1 / 6) * ((1 + qD)**(1 / 3) - (1 - qD)**(1 / 3))
        sel = (r < 1) * (D >= 0)
        m0[sel] = (q[sel] + np.sqrt(D[sel]))**(1 / 3) + (q[sel] - np.sqrt(D[sel]))**(1 / 3)
        sel = D < 0
        root = np.sqrt((-p[sel]).clip(0, None))
        m0[sel] = 2 * root * np.cos(np.arccos(q[sel] * root**(-3)) / 3.)
        return m0.clip(0, 1)

    def _envelope(self):
        r, mu, st = self.r, self.mu, self.sin_theta
        mu0 = self._origin(mu)
        st0 = np.sqrt(1 - mu0**2)
        mm = 1 - st0**2 / r
        vr = -np.sqrt(1 + mm) / np.sqrt(r)
        vt = self.sign_mu * np.sqrt(1 + mm) * (mu0 - mu) / st / np.sqrt(r)
        vp = np.sqrt(1 - mm) * st0 / st / np.sqrt(r)
        cmu = mu.clip(0.17, 1.0)
        cmu0 = self._origin(cmu)
        cmm = cmu0 / cmu
        rho = (np.sqrt((1 + cmm) / 2) * (2 * cmu0**2 / r + cmm))**(-1) / np.sqrt(r)**3
        d = self.in_disk
        vr[d] = 0
        vt[d] = 0
        vp[d] = 0
        rho[d] = 0
        return vr, vt, vp, rho

    def _disk(self):
        r = self.r
        vp = self.sin_theta / np.sqrt(r)
        rho = r**(-PLS) * H0 / self.H * np.exp(-0.5 * (self.zcyl / self.H)**2)
        out = ~self.in_disk
        vp[out] = 0
        rho[out] = 0
        return vp, rho

    def packed(self):
        """[ntheta][nr][5] array in include/slampv.h's order."""
        return np.ascontiguousarray(np.stack(
            [self.rho_env, self.rho_disk, self.vr_env, self.vt_env, self.vp_all], axis=-1))


@functools.lru_cache(maxsize=1)
def tables():
    return EnvelopeDiskTables()


# ----------------------------------------------------------------------------
class Level:
    """One nesting level: axes of its full box and (unless innermost) the index
    box [i0, i1] per axis that the next level refines `nsub`-fold."""

    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z
        self.box = None
        self.nsub = 1

    @property
    def shape(self):
        return len(self.x), len(self.y), len(self.z)

    def covered(self):
        """bool [nx][ny][nz]: cells handed to the next level (grid.py:149-176)."""
        m = np.zeros(self.shape, dtype=bool)
        if self.box is not None:
            (x0, x1), (y0, y1), (z0, z1) = self.box
            m[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = True
        return m


def _span(t, lim):
    inside = np.nonzero((lim[0] <= t) * (t <= lim[1]))[0]       # grid.py:358-359
    return int(inside.min()), int(inside.max())


def _split(t, i0, i1, nsub):
    step = t[1] - t[0]                                          # grid.py:374-393
    edges = np.linspace(t[i0] - 0.5 * step, t[i1] + 0.5 * step, (i1 - i0 + 1) * nsub + 1)
    return 0.5 * (edges[:-1] + edges[1:])


def build_levels(x, y, z, nnest, reslim=10, xlim=None, ylim=None, zlim=None):
    """Nested3DGrid(x, y, z, xlim, ylim, zlim, nnest, nlevels=len(nnest), reslim)
    (grid.py:7-69): a list of Level."""
    levels = [Level(np.asarray(x, float), np.asarray(y, float), np.asarray(z, float))]
    if not nnest:
        return levels
    steps = np.array([x[1] - x[0], y[1] - y[0], z[1] - z[0]])
    auto = [[], [], []]
    for f in nnest:                                             # grid.py:84-95
        for a in range(3):
            auto[a].append([-steps[a] * reslim, steps[a] * reslim])
        steps = steps / f
    lims = [auto[a] if given is None else given
            for a, given in enumerate((xlim, ylim, zlim))]
    for l, f in enumerate(nnest):
        up = levels[-1]
        axes = (up.x, up.y, up.z)
        box = tuple(_span(t, lims[a][l]) for a, t in enumerate(axes))
        up.box, up.nsub = box, int(f)
        levels.append(Level(*(_split(t, i0, i1, f) for t, (i0, i1) in zip(axes, box))))
    return levels


def subgrid(t, n):
    """mockpvd.py:158-166."""
    step = t[1] - t[0]
    edges = np.linspace(t[0] - 0.5 * step, t[-1] + 0.5 * step, len(t) * n + 1)
    return 0.5 * (edges[:-1] + edges[1:])


def model_y_axis(x, beam):
    """mockpvd.py:174-184."""
    dx = x[1] - x[0]
    if beam is None:
        return np.array([-dx, 0., dx])
    n = int(beam[0] / dx * 3. / 2.35)
    return np.arange(-n - 1, n + 2, 1) * dx


def sign_convention(signmajor, signminor, pa_major, pa_minor):
    """(iradshift, signmajor, signminor) as MockPVD.__init__ fixes them
    (mockpvd.py:84-90)."""
    red = pa_major + (0. if signmajor > 0 else 180.)
    blue = pa_minor + (0. if signminor > 0 else 180.)
    dpa = np.angle(np.exp(1j * np.radians(blue - red)))
    if dpa > 0:
        return 0, signmajor, signminor
    return np.pi, -signmajor, signminor


def cell_constants(level, axis, irad, rc_first, T=None):
    """The reference's per-cell caches for one level and PV axis, on the level's
    full (nx, ny, nz) box (mockpvd.py:210-219; precalculation.py:29-48, 104-131,
    169-179).  `rc_first` is the Rc of the evaluation that fills the caches."""
    T = T or tables()
    X, Y, Z = np.meshgrid(level.x, level.y, level.z, indexing='ij')
    X, Y, Z = X / rc_first, Y / rc_first, Z / rc_first
    if axis == 'minor':
        X, Y = -Y, X
    # basis of the inclined frame, azimuth 0 (rotbase)
    t, p = irad, 0
    er = (np.sin(t) * np.sin(p), -np.sin(t) * np.cos(p), np.cos(t))
    et = (np.cos(t) * np.sin(p), -np.cos(t) * np.cos(p), -np.sin(t))
    ep = (np.cos(p), np.sin(p), 0.0)
    xe, ye, ze = (ep[i] * X + (-et[i]) * Y + er[i] * Z for i in range(3))
    r = np.sqrt(xe * xe + ye * ye + ze * ze).clip(1e-10, None)
    theta = np.arccos(ze / r)
    phi = np.arctan2(xe, -ye)
    r_org = r * rc_first
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    l0, l1, l2 = 0, -np.sin(irad), np.cos(irad)
    zero = np.zeros_like(phi)
    elos_r = -((st * sp) * l0 + (-st * cp) * l1 + ct * l2)
    elos_t = -((ct * sp) * l0 + (-ct * cp) * l1 + (-st) * l2)
    elos_p = -(cp * l0 + sp * l1 + zero * l2)
    idx_t = ((theta - T.theta0) / T.dtheta + 0.5).astype(int)
    j_org = (np.log(r_org) - T.lnr0) / T.dlnr + 0.5
    return dict(idx_t=idx_t, j_org=j_org, r_org=r_org, elos_r=elos_r,
                elos_t=elos_t, elos_p=elos_p)


def device_layout(level, consts, rin, rout):
    """-> dict of [nz][ny][nx] arrays for slampv_level; cells that the next level
    covers, or with r <= rin / r > rout (mockpvd.py:228-229, 298), get
    idx_t = -1."""
    dead = level.covered() | (consts['r_org'] <= rin)
    if rout is not None:
        dead |= consts['r_org'] > rout
    it = np.where(dead, -1, consts['idx_t']).astype(np.int32)
    out = {'idx_t': np.ascontiguousarray(it.transpose(2, 1, 0))}
    for k in ('j_org', 'elos_r', 'elos_t', 'elos_p'):
        out[k] = np.ascontiguousarray(consts[k].transpose(2, 1, 0), dtype=np.float64)
    return out


def velocity_edges(v):
    dv = v[1] - v[0]                                            # mockpvd.py:259-261
    return np.hstack([v - dv * 0.5, v[-1] + 0.5 * dv])


def spectral_kernel(v, linewidth):
    nv = len(v)                                                 # mockpvd.py:310-311
    g = np.exp(-(v - v[nv // 2 - 1 + nv % 2])**2. / linewidth**2.)
    return g / np.sum(g)


def beam_kernel(x, y, beam, pa):
    """Normalised rotated Gaussian on the level-0 (y, x) pixels
    (mockpvd.py:321-324 with utils.rot)."""
    bmaj, bmin, bpa = beam
    ang = -np.radians(bpa - pa)
    gx, gy = np.meshgrid(x, y)
    xb = gx * np.cos(ang) - gy * np.sin(ang)
    yb = gx * np.sin(ang) + gy * np.cos(ang)
    g = np.exp2(-4 * ((xb / bmaj)**2 + (yb / bmin)**2))
    return g / np.sum(g)
