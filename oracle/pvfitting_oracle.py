"""CPU oracle of SLAM pvfitting's mock-PV likelihood.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by tests/golden/make_golden_pv.py
and by bench.py's cpu_baseline leg; never by the product (slam_amd/), which
runs on the GPU only.

It restates, in whole-array float64 numpy, what the reference does for one
parameter vector (Mstar, Rc, alphainfall, taumax, frho[, sig_mdl]):

  tables of the UCM envelope + cut-off disk     pvfitting/precalculation.py:6-102,146-159
  nested 3-D grid and its cell lists            pvfitting/grid.py:5-183,342-394
  per-grid caches (filled on the FIRST call)    pvfitting/precalculation.py:161-179
                                                pvfitting/mockpvd.py:203-219
  density / line-of-sight velocity gather       pvfitting/precalculation.py:181-192
  masks, joint normalisation                    pvfitting/mockpvd.py:127-137,228-229
  collapse, rho2tau, diagonal binning, nesting  pvfitting/grid.py:186-251
                                                pvfitting/precalculation.py:136-144
                                                pvfitting/mockpvd.py:268-304
  spectral smoothing, 1-exp(-tau), beam, PV cut pvfitting/mockpvd.py:306-335
  flux scale and modified log-likelihood        pvfitting/_pvfitting.py:142-151,181-195

Parity status: PINNED.  tests/golden/pvfit_*.npz were written by
tests/golden/make_golden_pv.py, which imports the reference itself (inert stubs
for astropy/emcee/corner/dynesty/numba, tests/golden/_refstubs.py) and runs it
on the reference's own test.pvfitting.{major,minor}.fits and on small synthetic
grids; tests/test_pv_oracle_golden.py checks this file against them
(bit-for-bit for tables, caches and tau; 1e-13 for the FFT-convolved PVs), and
against the survey's known answers nanmax(major)=0.4066004035049253,
nansum(major)=163.9387398678693, nansum(minor)=59.05831187700667.

Reference quirks kept on purpose (each is visible in the golden vectors):
  * the caches are filled on the first evaluation with THAT evaluation's Rc
    (`rc_first` here) and never refreshed;
  * `dlnr = exp(lnr[1]-lnr[0]) - 1`, not the table's own step;
  * the nested add averages only the DIAGONAL sub-cells (mockpvd.py:272-277);
  * the beam kernel is built once, with the MAJOR axis' position angle, and
    reused for the minor-axis cut (mockpvd.py:320-326 + module cache);
  * the flux scale sums model^2 over masked channels too (_pvfitting.py:149-150).
"""
from __future__ import annotations

import numpy as np
from scipy.signal import convolve

G_SI = 6.6743e-11
GM_SUN = 1.3271244e20
M_SUN = GM_SUN / G_SI
AU_M = 149597870700.0
PARAMKEYS = ('Mstar', 'Rc', 'alphainfall', 'taumax', 'frho', 'sig_mdl')


# ----------------------------------------------------------------------------
# UCM envelope + disk tables                      precalculation.py:6-102,146-159
# ----------------------------------------------------------------------------
def _mu0(r, mu):
    """Streamline origin cos(theta0) of the UCM solution (:50-70)."""
    p = (r - 1) / 3.
    q = mu * r / 2.
    D = q**2 + p**3
    out = np.full_like(r, np.nan)
    far = r >= 1
    qD = q[far] * (D[far])**(-1 / 2)
    out[far] = D[far]**(1 / 6) * ((1 + qD)**(1 / 3) - (1 - qD)**(1 / 3))
    near = (r < 1) * (D >= 0)
    out[near] = (q[near] + np.sqrt(D[near]))**(1 / 3) \
        + (q[near] - np.sqrt(D[near]))**(1 / 3)
    neg = D < 0
    s = np.sqrt((-p[neg]).clip(0, None))
    out[neg] = 2 * s * np.cos(np.arccos(q[neg] * s**(-3)) / 3.)
    return out.clip(0, 1)


_TABLES = None


def tables():
    """dict of the (Ntheta, Nr) lookup tables and their axis constants."""
    global _TABLES
    if _TABLES is not None:
        return _TABLES
    Nr = 1600
    lnr = np.linspace(np.log(1e-4), np.log(1e4), Nr)
    dlnr = np.exp(lnr[1] - lnr[0]) - 1
    Ntheta = int(np.pi / dlnr + 0.5)
    theta = np.linspace(0, np.pi, Ntheta)
    lnr_mesh, th = np.meshgrid(lnr, theta)
    r = np.exp(lnr_mesh)
    H0, plh, pls = 0.2, 0.25, 1.0
    cos_th = np.cos(th)
    sgn = np.sign(cos_th)
    mu = np.abs(cos_th)
    sin_th = np.sin(th).clip(1e-10, None)
    R = r * sin_th
    zz = r * mu
    H = H0 * R**(1. + plh)
    in_disk = (R < 1) * (zz < 3 * H)
    # envelope (:73-92)
    mu0 = _mu0(r, mu)
    sin0 = np.sqrt(1 - mu0**2)
    mm = 1 - sin0**2 / r
    vr = -np.sqrt(1 + mm) / np.sqrt(r)
    vt = sgn * np.sqrt(1 + mm) * (mu0 - mu) / sin_th / np.sqrt(r)
    vp = np.sqrt(1 - mm) * sin0 / sin_th / np.sqrt(r)
    cmu = mu.clip(0.17, 1.0)
    cmu0 = _mu0(r, cmu)
    cmm = cmu0 / cmu
    rho_env = (np.sqrt((1 + cmm) / 2) * (2 * cmu0**2 / r + cmm))**(-1) \
        / np.sqrt(r)**3
    for a in (vr, vt, vp, rho_env):
        a[in_disk] = 0
    # disk (:94-102)
    vp_disk = sin_th / np.sqrt(r)
    rho_disk = r**(-pls) * H0 / H * np.exp(-0.5 * (zz / H)**2)
    vp_disk[~in_disk] = 0
    rho_disk[~in_disk] = 0
    _TABLES = dict(Nr=Nr, Ntheta=Ntheta, lnr0=lnr[0], dlnr=dlnr,
                   theta0=theta[0], dtheta=theta[1] - theta[0],
                   rho_env=rho_env, rho_disk=rho_disk, vr_env=vr, vt_env=vt,
                   vp_all=vp + vp_disk)
    return _TABLES


# ----------------------------------------------------------------------------
# nested grid                                            grid.py:5-183, 342-394
# ----------------------------------------------------------------------------
def _edge_range(t, lim):
    """First/last index with lim[0] <= t <= lim[1] (grid.py:357-359)."""
    hit = np.nonzero((lim[0] <= t) * (t <= lim[1]))[0]
    return int(np.min(hit)), int(np.max(hit))


def _refine(t, i0, i1, nsub):
    """Cell centres of [i0, i1] split nsub-fold (grid.py:374-393)."""
    d = t[1] - t[0]
    e = np.linspace(t[i0] - 0.5 * d, t[i1] + 0.5 * d, (i1 - i0 + 1) * nsub + 1)
    return 0.5 * (e[:-1] + e[1:])


def _shell_perm(shape, box):
    """Positions (C-order flat indices of the full level grid) of the cells a
    level keeps once its inner box went to the child level, in the reference's
    list order: six slabs (grid.py:149-176)."""
    full = np.arange(int(np.prod(shape))).reshape(shape)
    (x0, x1), (y0, y1), (z0, z1) = box
    mid = slice(x0, x1 + 1)
    ymid = slice(y0, y1 + 1)
    return np.concatenate([
        full[:x0].ravel(), full[x1 + 1:].ravel(),
        full[mid, :y0].ravel(), full[mid, y1 + 1:].ravel(),
        full[mid, ymid, :z0].ravel(), full[mid, ymid, z1 + 1:].ravel()])


def nested_grid(x, y, z, nsub, reslim):
    """Levels 0..len(nsub) of the nested grid with automatic nesting limits
    (grid.py:54-61, 84-95, 129-183).  Returns a dict of per-level lists:
    axes [(x,y,z)], shape [(nx,ny,nz)], box [((x0,x1),(y0,y1),(z0,z1)) of the
    CHILD inside this level | None], perm [kept cells | None = all], and the
    flat coordinate lists xyz [(X,Y,Z)]."""
    nsub = list(nsub)
    axes = [(np.asarray(x), np.asarray(y), np.asarray(z))]
    step = [x[1] - x[0], y[1] - y[0], z[1] - z[0]]
    lims = []
    for f in nsub:                                    # :84-95
        lims.append([[-s * reslim, s * reslim] for s in step])
        step = list(np.array(step) / f)
    boxes = []
    for l, f in enumerate(nsub):
        rng = [_edge_range(t, lim) for t, lim in zip(axes[l], lims[l])]
        boxes.append(tuple(rng))
        axes.append(tuple(_refine(t, i0, i1, f)
                          for t, (i0, i1) in zip(axes[l], rng)))
    boxes.append(None)
    shapes = [tuple(len(t) for t in a) for a in axes]
    perms, xyz = [], []
    for a, shp, box in zip(axes, shapes, boxes):
        X, Y, Z = (m.ravel() for m in np.meshgrid(*a, indexing='ij'))
        if box is None:
            perms.append(None)
        else:
            pm = _shell_perm(shp, box)
            perms.append(pm)
            X, Y, Z = X[pm], Y[pm], Z[pm]
        xyz.append((X, Y, Z))
    return dict(nlevels=len(axes), nsub=nsub, axes=axes, shape=shapes,
                box=boxes, perm=perms, xyz=xyz)


def _block_nanmean(a, n):
    """nanmean over n^3 blocks, summed in the reference's k,j,i order
    (grid.py:303-323)."""
    parts = np.array([a[k::n, j::n, i::n]
                      for k in range(n) for j in range(n) for i in range(n)])
    return np.nanmean(parts, axis=0)


def collapse(grid, d, upto=0):
    """Per-level lists d -> full array on level `upto` (grid.py:186-251)."""
    cur = d[-1]
    for l in range(grid['nlevels'] - 1, upto, -1):
        child = _block_nanmean(cur.reshape(grid['shape'][l]), grid['nsub'][l - 1])
        shp = grid['shape'][l - 1]
        full = np.full(int(np.prod(shp)), np.nan)
        full[grid['perm'][l - 1]] = d[l - 1]
        full = full.reshape(shp)
        (x0, x1), (y0, y1), (z0, z1) = grid['box'][l - 1]
        full[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = child
        cur = full
    return cur


# ----------------------------------------------------------------------------
# setup = MockPVD.__init__ + the caches the first evaluation fills
# ----------------------------------------------------------------------------
def subgrid_axis(t, n):
    """mockpvd.py:158-166."""
    d = t[1] - t[0]
    e = np.linspace(t[0] - 0.5 * d, t[-1] + 0.5 * d, len(t) * n + 1)
    return 0.5 * (e[:-1] + e[1:])


def _frame(irad, X, Y, Z):
    """Observer (X, Y, Z) -> envelope (r, theta, phi) (precalculation.py:104-131
    with phi=0)."""
    t, p = irad, 0
    er = np.array([np.sin(t) * np.sin(p), -np.sin(t) * np.cos(p), np.cos(t)])
    et = np.array([np.cos(t) * np.sin(p), -np.cos(t) * np.cos(p), -np.sin(t)])
    ep = np.array([np.cos(p), np.sin(p), np.zeros_like(p)])
    xe, ye, ze = np.outer(ep, X.ravel()) + np.outer(-et, Y.ravel()) \
        + np.outer(er, Z.ravel())
    r = np.linalg.norm([xe, ye, ze], axis=0).clip(1e-10, None)
    return r, np.arccos(ze / r), np.arctan2(xe, -ye)


def _los_basis(theta, phi, irad):
    """Projections of the spherical unit vectors on the line of sight
    (precalculation.py:29-48)."""
    ct, st = np.cos(theta), np.sin(theta)
    cp, sp = np.cos(phi), np.sin(phi)
    er = np.array([st * sp, -st * cp, ct])
    et = np.array([ct * sp, -ct * cp, -st])
    ep = np.array([cp, sp, np.zeros_like(phi)])
    los = np.array([0, -np.sin(irad), np.cos(irad)])[:, np.newaxis]
    return (-np.sum(er * los, axis=0), -np.sum(et * los, axis=0),
            -np.sum(ep * los, axis=0))


def make_setup(x, z, v, beam, *, incl, rc_first, nnest=(2, 2, 2, 2, 2, 2),
               nsubgrid=1, reslim=10, signmajor=1, signminor=1, pa_major=0.,
               pa_minor=90., linewidth=None, rin=1., rout=None):
    """Everything one fit holds constant: MockPVD.__init__ (mockpvd.py:64-95,
    168-192) plus the module-level caches (precalculation.py:133-135, 161-179;
    mockpvd.py:260-261, 308-312, 320-325) as the first call with Rc=rc_first
    leaves them."""
    x = np.asarray(x, dtype=float)
    z = np.asarray(z, dtype=float)
    v = np.asarray(v, dtype=float)
    T = tables()
    x_in = x
    if nsubgrid > 1:
        x, z = subgrid_axis(x, nsubgrid), subgrid_axis(z, nsubgrid)
    # sign bookkeeping (:84-90)
    red = pa_major + (0. if signmajor > 0 else 180.)
    blue = pa_minor + (0. if signminor > 0 else 180.)
    dpa = np.angle(np.exp(1j * np.radians(blue - red)))
    iradshift = 0 if dpa > 0 else np.pi
    sgn_major = signmajor if dpa > 0 else -signmajor
    sgn_minor = signminor
    dx = x[1] - x[0]
    if beam is not None:                               # :176-184
        n = int(beam[0] / dx * 3. / 2.35)
        y = np.arange(-n - 1, n + 2, 1) * dx
    else:
        y = np.array([-dx, 0., dx])
    if nnest is not None:
        grid = nested_grid(x, y, z, nnest, reslim)
    else:
        grid = nested_grid(x, y, z, [], reslim)
    irad = np.abs(np.arcsin(np.sin(np.radians(incl)))) + iradshift
    cache = {}
    for axis in ('major', 'minor'):
        per = []
        for l in range(grid['nlevels']):
            X, Y, Z = (c / rc_first for c in grid['xyz'][l])
            if axis == 'major':
                r, t, p = _frame(irad, X, Y, Z)
            else:
                r, t, p = _frame(irad, -Y, X, Z)
            r_org = r * rc_first
            er, et, ep = _los_basis(t, p, irad)
            it = ((t - T['theta0']) / T['dtheta'] + 0.5).astype(int)
            j_org = (np.log(r_org) - T['lnr0']) / T['dlnr'] + 0.5
            per.append(dict(idx_t=it, j_org=j_org, r_org=r_org,
                            elos_r=er, elos_t=et, elos_p=ep))
        cache[axis] = per
    nv = len(v)
    delv = v[1] - v[0]
    vedge = np.hstack([v - delv * 0.5, v[-1] + 0.5 * delv])
    gauss_v = None
    if linewidth is not None:
        g = np.exp(-(v - v[nv // 2 - 1 + nv % 2])**2. / linewidth**2.)
        gauss_v = (g / np.sum(g))[:, np.newaxis, np.newaxis]
    gauss_xy = None
    if beam is not None:
        bmaj, bmin, bpa = beam
        a = -np.radians(bpa - pa_major)
        gx, gy = np.meshgrid(x, y)
        xb = gx * np.cos(a) - gy * np.sin(a)
        yb = gx * np.sin(a) + gy * np.cos(a)
        g = np.exp2(-4 * ((xb / bmaj)**2 + (yb / bmin)**2))
        gauss_xy = (g / np.sum(g))[np.newaxis, :, :]
    return dict(x_in=x_in, x=x, y=y, z=z, v=v, nsubgrid=nsubgrid, grid=grid,
                cache=cache, irad=irad, iradshift=iradshift,
                signmajor=sgn_major, signminor=sgn_minor, vedge=vedge,
                gauss_v=gauss_v, gauss_xy=gauss_xy, rin=rin, rout=rout,
                rc_first=rc_first, incl=incl, tables=T)


# ----------------------------------------------------------------------------
# forward model
# ----------------------------------------------------------------------------
def rho_vlos(setup, axis, l, Mstar, Rc, alphainfall, frho):
    """Density and line-of-sight velocity [km/s] of level l's cell list
    (precalculation.py:181-192; mockpvd.py:204, 222-229)."""
    T, c = setup['tables'], setup['cache'][axis][l]
    i = c['idx_t']
    j = np.clip(c['j_org'] - np.log(Rc) / T['dlnr'], 0, T['Nr'] - 1).astype(int)
    rho = T['rho_env'][i, j] + T['rho_disk'][i, j] * frho
    vr = T['vr_env'][i, j] * alphainfall
    vlos = vr * c['elos_r'] + T['vt_env'][i, j] * c['elos_t'] \
        + T['vp_all'][i, j] * c['elos_p']
    vunit = np.sqrt(G_SI * Mstar * M_SUN / Rc / AU_M) * 1e-3
    vlos = vlos * vunit
    rho[c['r_org'] <= setup['rin']] = 0.
    if setup['rout'] is not None:
        rho[c['r_org'] > setup['rout']] = 0
    return rho, vlos


def rho2tau(vedge, vlos, rho):
    """Column sums of rho per velocity channel (precalculation.py:136-144):
    (nx, ny, nz) -> (nv, ny, nx)."""
    nv = len(vedge) - 1
    nx, ny, _ = vlos.shape
    tau = np.zeros((nv, ny, nx))
    for k in range(nv):
        sel = (vedge[k] <= vlos) * (vlos < vedge[k + 1])
        tau[k] = np.sum(sel * rho, axis=2).T
    return tau


def _diag_mean(a, n):
    """The nested add's averaging of a (nv, ny, nx) array onto the parent
    pixels: only sub-cells (i, i) enter, each n times (mockpvd.py:272-277)."""
    parts = np.array([a[:, i::n, i::n] for _ in range(n) for i in range(n)])
    return np.nanmean(parts, axis=0)


def tau_cube(setup, rho, vlos, stages=None):
    """Nested optical-depth cube on level 0 (mockpvd.py:255-304)."""
    g = setup['grid']
    L = g['nlevels']
    rho_col = [collapse(g, rho, upto=l) for l in range(L)]
    vlos_col = [collapse(g, vlos, upto=l) for l in range(L)]
    shp = g['shape'][-1]
    dz = g['axes'][-1][2][1] - g['axes'][-1][2][0]
    tau = rho2tau(setup['vedge'], vlos_col[-1].reshape(shp),
                  rho_col[-1].reshape(shp)) * dz
    for l in range(L - 2, -1, -1):
        shp = g['shape'][l]
        r_l = rho_col[l].reshape(shp)
        v_l = vlos_col[l].reshape(shp)
        dz = g['axes'][l][2][1] - g['axes'][l][2][0]
        (x0, x1), (y0, y1), (z0, z1) = g['box'][l]
        r_l[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 0.
        own = rho2tau(setup['vedge'], v_l, r_l) * dz
        own[:, y0:y1 + 1, x0:x1 + 1] += _diag_mean(tau, g['nsub'][l])
        tau = own
    if stages is not None:
        stages.setdefault('tau', []).append(tau)
    return tau


def pv_cut(setup, tau, taumax):
    """tau cube -> PV diagram (mockpvd.py:306-335)."""
    if setup['gauss_v'] is not None:
        tau = convolve(tau, setup['gauss_v'], mode='same')
    I = 1. - np.exp(-tau / np.nanmax(tau) * taumax)
    if setup['gauss_xy'] is not None:
        I = convolve(I, setup['gauss_xy'], mode='same')
    pv = I[:, len(setup['y']) // 2, :]
    n = setup['nsubgrid']
    if n > 1:
        pv = np.nanmean(np.array([pv[:, i::n] for i in range(n)]), axis=0)
    return pv


def mockpvd(setup, Mstar, Rc, alphainfall, taumax, frho, stages=None):
    """[major PV, minor PV], each (nv, nx) (mockpvd.py:122-147)."""
    L = setup['grid']['nlevels']
    rho, vlos = {}, {}
    for axis in ('major', 'minor'):
        pairs = [rho_vlos(setup, axis, l, Mstar, Rc, alphainfall, frho)
                 for l in range(L)]
        rho[axis] = [p[0] for p in pairs]
        vlos[axis] = [p[1] for p in pairs]
    top = np.nanmax([np.nanmax([np.nanmax(a) for a in rho[axis]])
                     for axis in ('major', 'minor')])
    out = []
    for axis, sign in (('major', setup['signmajor']),
                       ('minor', setup['signminor'])):
        r = [a / top for a in rho[axis]]
        tau = tau_cube(setup, r, vlos[axis], stages)
        out.append(pv_cut(setup, tau, taumax)[:, ::sign])
    return out


# ----------------------------------------------------------------------------
# likelihood                                     _pvfitting.py:110-152, 181-195
# ----------------------------------------------------------------------------
def make_obs(x, v, dpvmajor, dpvminor, sigma, bmaj, bmin, vmask=(0, 0)):
    """Masked observations and the oversampling factor (:111-123)."""
    x = np.asarray(x, dtype=float)
    dx = x[1] - x[0]
    rarea = np.pi / (4. * np.log(2.)) * bmaj * bmin / dx / dx
    _, vv = np.meshgrid(x, v)
    hide = (vmask[0] < vv) * (vv < vmask[1])
    return dict(major=np.where(hide, np.nan, dpvmajor),
                minor=np.where(hide, np.nan, dpvminor), sigma=sigma,
                rarea=rarea)


def scaled_model(setup, obs, q):
    """makemodel (:142-151): flux-scaled [major, minor] and the scale."""
    major, minor = mockpvd(setup, *q[:5])
    f = (np.nansum(obs['major'] * major) + np.nansum(obs['minor'] * minor)) \
        / (np.nansum(major * major) + np.nansum(minor * minor))
    return f * major, f * minor, f


def lnprob(setup, obs, q):
    """Modified log-likelihood for natural-unit parameters q[6] (:181-195)."""
    s2 = (1. + q[5]**2) * obs['sigma']**2
    major, minor, _ = scaled_model(setup, obs, q)
    c1 = np.nansum((obs['major'] - major)**2 / s2 + np.log(s2))
    c2 = np.nansum((obs['minor'] - minor)**2 / s2 + np.log(s2))
    return -0.5 * (c1 + c2) / np.sqrt(obs['rarea'])


def chi2_fixed(setup, obs, q):
    """The all-fixed branch (:221-232): no log term."""
    s2 = (1. + q[5]**2) * obs['sigma']**2
    major, minor, _ = scaled_model(setup, obs, q)
    c1 = np.nansum((obs['major'] - major)**2 / s2)
    c2 = np.nansum((obs['minor'] - minor)**2 / s2)
    return (c1 + c2) / np.sqrt(obs['rarea'])


def getquad(m):
    """Sign of the PV diagram's dominant quadrant pair (_pvfitting.py:462-469)."""
    nv, nx = np.shape(m)
    q = np.sum(m[:nv // 2, :nx // 2]) + np.sum(m[nv // 2:, nx // 2:]) \
        - np.sum(m[nv // 2:, :nx // 2]) - np.sum(m[:nv // 2, nx // 2:])
    return int(np.sign(q))
