"""CPU oracle for the channelfit likelihood path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy/scipy restatement of the algorithm in the reference
(jinshisai/SLAM, `channelfit/_channelfit.py`).  It exists to CHECK the HIP
path; nothing in `slam_amd/` imports it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may use it.

Parity status: PINNED.  `tests/golden/make_golden.py` runs the reference itself
(imported from /root/reference in the build container) on seeded synthetic
cubes and stores its inputs/outputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every stage of this restatement against
those vectors (and against the survey's recorded known answer
chi2 = 4229.974367626359).

All arithmetic is float64, whole-array numpy, in the same operation order as
the reference so that the truncating table lookup (`_channelfit.py:462-463`)
selects the same table entry.  Third-party arithmetic on the path is the same
as the reference's: scipy.signal.convolve (`:530`), scipy
RegularGridInterpolator (`:479-481`), scipy.special.erf (`:424-425`).

Citations `:N` below are line numbers in `channelfit/_channelfit.py` unless a
file name is given.
"""
from __future__ import annotations

import numpy as np
from scipy.interpolate import RegularGridInterpolator
from scipy.optimize import curve_fit
from scipy.signal import convolve
from scipy.special import erf

# astropy values the reference reads at import (:29-32): CODATA-2018 G, the IAU
# 2015 nominal solar mass parameter divided by G, and the au in metres.
G_SI = 6.6743e-11
M_SUN_SI = 1.3271244e20 / 6.6743e-11
AU_M = 149597870700.0
VUNIT = np.sqrt(G_SI * M_SUN_SI / AU_M) * 1e-3   # km/s at 1 au for 1 Msun

PARAMKEYS = ('Mstar', 'Rc', 'cs', 'h1', 'h2', 'pI', 'Rin', 'Ienv',
             'xoff', 'yoff', 'voff', 'incloff', 'paoff')       # :219-221


def rot(x, y, pa):
    """utils.py:354-357 -- (minor-axis, major-axis) coordinates."""
    c, s = np.cos(pa), np.sin(pa)
    return np.array([x * c - y * s, x * s + y * c])


def moment_signs(data_valid, v_valid, sigma, X, Y, pa_rad):
    """Rotation signs from the moment-1 map (:40-50, :276-289)."""
    dvmin = np.min(v_valid[1:] - v_valid[:-1])
    with np.errstate(all='ignore'):
        mom0 = np.sum(data_valid, axis=0) * dvmin
        sigma_mom0 = sigma * dvmin * np.sqrt(len(data_valid))
        vv = v_valid[:, None, None]
        mom1 = np.sum(data_valid * vv, axis=0) / np.sum(data_valid, axis=0)
        mom1[mom0 < 3 * sigma_mom0] = np.nan
    Xr, Yr = rot(X, Y, pa_rad)
    signmajor = np.sign(np.nansum(mom1 * Yr))
    signminor = np.sign(np.nansum(mom1 * Xr)) * (-1)
    return signmajor, signminor


def moment0(data_valid, v_valid, sigma):
    """(mom0, sigma_mom0) of `makemom01` (:40-44)."""
    dvmin = np.min(v_valid[1:] - v_valid[:-1])
    return (np.sum(data_valid, axis=0) * dvmin,
            sigma * dvmin * np.sqrt(len(data_valid)))


def clean_mom0(mom0, beam, sigma, threshold=2, gain=0.01, weakest=0.3):
    """Hogbom-style CLEAN of the moment-0 map in Jy/pixel (:52-87): peak by
    peak, a component max(gain*peak, weakest*sigma)/beamarea is moved from the
    residual to the model until the residual peak drops under threshold*sigma;
    the last residual (per beam area) is added back."""
    area = np.sum(beam)
    model = mom0 * 0
    resid = mom0 * 1
    for _ in range(10000000 - 1):
        if np.nanmax(resid) < threshold * sigma:
            break
        at = np.unravel_index(np.nanargmax(resid), mom0.shape)
        comp = np.zeros_like(mom0)
        comp[at] = max(gain * resid[at], weakest * sigma) / area
        resid = resid - convolve(comp, beam, mode='same')
        model = model + comp
    return model + resid / area


def ft_deconvolve(mom0, x, y, bmaj, bmin, bpa, sigma, threshold=3):
    """Division by the beam in the Fourier plane (:180-210): odd-sized crop,
    beam transform made real and floored at 1 % of its peak (set to 1 below),
    pixels under threshold*sigma zeroed, the cropped row/column put back as 0."""
    cx, cy = int(len(x) % 2 == 0), int(len(y) % 2 == 0)
    xd, yd, d = x[cx:], y[cy:], mom0[cy:, cx:]
    ny, nx = d.shape
    hy, hx = (ny - 1) // 2, (nx - 1) // 2
    dx, dy = x[1] - x[0], y[1] - y[0]
    xg = np.linspace(-hx * dx, hx * dx, nx)
    yg = np.linspace(-hy * dy, hy * dy, ny)
    s, t = rot(*np.meshgrid(xg, yg), np.radians(bpa))
    g = np.exp2(-4 * ((t / bmaj)**2 + (s / bmin)**2))
    u, w = np.meshgrid(np.fft.fftshift(np.fft.fftfreq(nx, d=dx)),
                       np.fft.fftshift(np.fft.fftfreq(ny, d=dy)))
    ph = 2 * np.pi * (u * (xg[-1] + dx) + w * (yg[-1] + dy))
    re = np.real(np.fft.fftshift(np.fft.fft2(g)) * np.exp(-1j * ph))
    Fg = re + 1j * 0
    Fg[re < np.max(re) * 1e-2] = 1 + 1j * 0
    ph = 2 * np.pi * (u * (xd[-1] + dx) + w * (yd[-1] + dy))
    Fd = np.fft.fftshift(np.fft.fft2(d)) * np.exp(-1j * ph)
    out = np.real(np.fft.ifft2(np.fft.ifftshift(Fd / Fg * np.exp(1j * ph))))
    out[np.abs(d) < sigma * threshold] = 0
    if cx:
        out = np.concatenate((np.zeros((out.shape[0], 1)), out), axis=1)
    if cy:
        out = np.concatenate((np.zeros((1, out.shape[1])), out), axis=0)
    return out


def model_deconvolve(mom0, x, y, bmaj, bmin, bpa, sigma, niter=20):
    """Coarse-grid deconvolution (:89-178, the default iterative branch): the
    map is rotated into the beam frame, sampled every half beam, and each
    sample is fitted in turn (20 sweeps) so that the bilinear model times the
    beam centred on it reproduces the rotated map there; samples under 2 sigma
    are zeroed.  Returns the model interpolated back onto (y, x)."""
    nx, ny = len(x), len(y)
    dx, dy = np.abs(x[1] - x[0]), np.abs(y[1] - y[0])
    sx, sy = int(np.floor(bmin / 2 / dx)), int(np.floor(bmaj / 2 / dy))
    hx = int(np.floor((nx - 1) / 2 / sx)) * sx
    hy = int(np.floor((ny - 1) / 2 / sy)) * sy
    mx, my = 2 * hx + 1, 2 * hy + 1
    ox, oy = int((nx - mx) / 2), int((ny - my) / 2)
    xi = x[::-1][ox:mx + ox]
    yi = y[oy:my + oy]
    di = mom0[:, ::-1][oy:my + oy, ox:mx + ox]
    Xi, Yi = np.meshgrid(xi, yi)
    g = np.exp2(-((Xi - xi[hx]) / (bmin / 2))**2 - ((Yi - yi[hy]) / (bmaj / 2))**2)
    gsum = np.sum(g)
    xm, ym = xi[::sx], yi[::sy]
    lin = dict(method='linear', bounds_error=False, fill_value=0)
    drot = RegularGridInterpolator((yi, xi), di, **lin)(
        tuple(rot(Xi, Yi, -np.radians(bpa)))[::-1])
    par = drot[::sy, ::sx].clip(0, None) / gsum
    for _ in range(niter):
        for ip in range(len(ym)):
            for jp in range(len(xm)):
                target = drot[ip * sy, jp * sx]
                if target < 2 * sigma:
                    par[ip, jp] = 0
                    continue
                shifted = np.roll(g, (ip * sy - hy, jp * sx - hx), axis=(0, 1))

                def response(pts, value):
                    trial = par + 0
                    trial[ip, jp] = value
                    f = RegularGridInterpolator((ym, xm), trial, **lin)
                    return np.sum(f(tuple(pts)) * shifted)
                best, _ = curve_fit(response, [Yi, Xi], target, p0=par[ip, jp],
                                    sigma=[sigma], absolute_sigma=True,
                                    bounds=[0, target])
                par[ip, jp] = best[0]
    f = RegularGridInterpolator((ym, xm), np.reshape(np.ravel(par), (len(ym), len(xm))),
                                **lin)
    return f(tuple(rot(*np.meshgrid(x, y), np.radians(bpa)))[::-1])


def make_setup(x, y, v, data, *, bmaj, bmin, bpa, sigma, pa, incl, rmax,
               vlim=(-100, 0, 0, 100), nlayer=3, gaussmargin=1.6,
               signmajor=None, signminor=None, disk=True, envelope=False,
               scaling='uniform'):
    """Everything `makegrid` (:227-371) derives once per fit, as a dict."""
    x, y, v = (np.asarray(a, dtype=float) for a in (x, y, v))
    dx, dy, dv = x[1] - x[0], y[1] - y[0], v[1] - v[0]      # utils.py:260-262
    dpix = min(abs(dx), abs(dy))                             # :243
    pa_rad = np.radians(pa)
    X, Y = np.meshgrid(x, y)                                 # :263
    blue = (vlim[0] <= v) * (v <= vlim[1])                   # :266
    red = (vlim[2] <= v) * (v <= vlim[3])                    # :268
    v_valid = np.r_[v[blue], v[red]]                         # :270
    data_valid = np.append(data[blue], data[red], axis=0)    # :272-274
    if signmajor is None or signminor is None:
        sm, sn = moment_signs(data_valid, v_valid, sigma, X, Y, pa_rad)
        signmajor = sm if signmajor is None else signmajor
        signminor = sn if signminor is None else signminor
    # nested grid (:293-320)
    r_need = rmax + gaussmargin * bmaj
    npix = int(2 * r_need / dpix + 0.5)
    npix = int(4 * np.ceil(npix / 4))
    nq1 = npix // 2 - npix // 2 // 2
    nq3 = nq1 + npix // 2
    half = npix // 2 - 0.5
    axes = np.array([np.linspace(-half, half, npix) * dpix / 2**l
                     for l in range(nlayer)])
    Xn = np.array([np.meshgrid(s, s)[0] for s in axes])
    Yn = np.array([np.meshgrid(s, s)[1] for s in axes])
    Xnest0, Ynest0 = rot(Xn, Yn, pa_rad)
    # beam kernel (:329-335)
    ngauss = int(gaussmargin * bmaj / dpix + 0.5)
    t = (np.arange(2 * ngauss + 1) - ngauss) * dpix
    xb, yb = rot(*np.meshgrid(t, t), np.radians(bpa))
    gaussbeam = np.exp2(-4 * ((yb / bmaj)**2 + (xb / bmin)**2))
    pixperbeam = np.sum(gaussbeam)
    # crop (:337-341)
    n_need = int(r_need / dpix + 0.5)
    ineed0, ineed1 = npix // 2 - n_need, npix // 2 + n_need
    xneed = axes[0][ineed0:ineed1]
    mom0decon = None
    if 'mom0' in scaling:                                    # :343-364
        gaussbeam = gaussbeam[:, ::-1]       # x axis into the observed order
        mom0, sig0 = moment0(data_valid, v_valid, sigma)
        if scaling == 'mom0clean':
            mom0decon = clean_mom0(mom0, gaussbeam, sig0)
        elif scaling == 'mom0model':
            mom0decon = model_deconvolve(mom0, x, y, bmaj, bmin, bpa, sig0)
        elif scaling == 'mom0ft':
            mom0decon = ft_deconvolve(mom0, x, y, bmaj, bmin, bpa, sig0)
    return dict(scaling=scaling, mom0decon=mom0decon,
                x=x, y=y, X=X, Y=Y, dv=dv, dpix=dpix, sigma=float(sigma),
                v_valid=v_valid, data_valid=data_valid,
                signmajor=float(signmajor), signminor=float(signminor),
                incl0=float(incl), pa_rad=pa_rad, nlayer=nlayer, npix=npix,
                nq1=nq1, nq3=nq3, Xnest0=Xnest0, Ynest0=Ynest0,
                gaussbeam=gaussbeam, pixperbeam=pixperbeam,
                ineed0=ineed0, ineed1=ineed1, xneed=xneed, yneed=xneed.copy(),
                disk=bool(disk), envelope=bool(envelope))


def prof_table(cs, dv):
    """Line-profile lookup table (:412-427): returns (prof, prof_n, prof_d)."""
    c = cs / dv
    w = max([c * 2.35482, 1])
    vmax = 2 * w
    d = w / 11
    n = 2 * int(2 * 11) + 1
    u = np.linspace(-vmax, vmax, n)
    if c < 0.01:
        p = (1 + np.sign(u + 0.5)) * (1 - np.sign(u - 0.5)) / 4
    else:
        p = erf((u + 0.5) / np.sqrt(2) / c) - erf((u - 0.5) / np.sqrt(2) / c)
        p[0] = p[n - 1] = 0
    return p, n - 1, d


def xdisk_surfaces(Xnest, Ynest, sini, cosi, tani, h1, h2):
    """Deprojected minor-axis coordinate of each emitting surface (:382-410)."""
    out = [None] * 4
    for slot, h in zip((0, 2), (h1, h2)):
        if h < 0:
            xa = xb = None
        elif h < 0.01:
            xa = Xnest / cosi
            xb = Xnest / cosi
        else:
            Xc = Xnest * cosi
            a = tani**(-2) - h**2
            b = (1 + h**2) * Xc
            c = (tani**2 - h**2) * Xc**2 - h**2 * Ynest**2
            if -1e-3 < a < 1e-3:
                xa, xb = Xc + c / b / 2, None
            else:
                D = b**2 - a * c
                ok = D >= 0
                z1 = np.full_like(Xnest, np.nan)
                z2 = np.full_like(Xnest, np.nan)
                rt = np.sqrt(D[ok])
                z1[ok] = (b[ok] + rt) / a
                z2[ok] = (b[ok] - rt) / a
                xa, xb = Xc + z1, Xc + z2
        out[slot], out[slot + 1] = xa, xb
    return out


def vlos_surface(xs, h, Ynest, Rc, Rin, sini, signmajor, signminor,
                 disk, envelope):
    """Line-of-sight velocity for Mstar = 1 on one surface (:429-449)."""
    if xs is None:
        return None
    r = np.hypot(xs, Ynest)
    outer = r > Rc
    vp = r**(-1 / 2) * (1 + h**2)**(-3 / 4)
    vr = r * 0
    if envelope:
        vp[outer] = np.sqrt(Rc) / r[outer]
        vr[outer] = -r[outer]**(-1 / 2) * np.sqrt(2 - Rc / r[outer])
    erot = Ynest * signmajor / r
    erad = xs * signminor / r
    vl = (vp * erot + vr * erad) * sini * VUNIT
    vl[r < Rin] = np.nan
    if not envelope:
        vl[outer] = np.nan
    if not disk:
        vl[~outer] = np.nan
    return vl


def avefour(a):
    """:35-38"""
    return (a[:, 0::2, 0::2] + a[:, 0::2, 1::2]
            + a[:, 1::2, 0::2] + a[:, 1::2, 1::2]) / 4.


def iunif(setup, xs_list, vlos_list, Ynest, prof, prof_n, prof_d,
          Mstar, Rc, pI, Ienv, voff, kchunk=None):
    """Unconvolved model on the cropped layer-0 grid (:454-473).

    `kchunk` (channels per pass) only bounds the memory of the whole-array
    temporaries at the largest cubes: every operation below is elementwise
    along the channel axis, so the result is bit-identical to one pass."""
    nv = len(setup['v_valid'])
    if kchunk is not None and kchunk < nv:
        parts = [iunif(dict(setup, v_valid=setup['v_valid'][k0:k0 + kchunk]), xs_list,
                       vlos_list, Ynest, prof, prof_n, prof_d, Mstar, Rc, pI, Ienv, voff)
                 for k0 in range(0, nv, kchunk)]
        return np.concatenate(parts, axis=0)
    total = 0
    for vl0, xs in zip(vlos_list, xs_list):
        if vl0 is None:
            continue
        vl = vl0 * np.sqrt(Mstar)
        v = np.subtract.outer(setup['v_valid'], vl) - voff
        iv = v / setup['dv'] / prof_d + prof_n // 2 + 0.5
        p = prof[iv.astype('int').clip(0, prof_n)]
        r = np.hypot(xs, Ynest)
        p = np.where(r < Rc, p, p * Ienv) / (Ienv + 1)
        if pI != 0:
            p = p * r**(-pI)
        total = total + np.nan_to_num(p)
    nq1, nq3 = setup['nq1'], setup['nq3']
    for l in range(setup['nlayer'] - 1, 0, -1):
        total[:, l - 1, nq1:nq3, nq1:nq3] = avefour(total[:, l])
    i0, i1 = setup['ineed0'], setup['ineed1']
    return total[:, 0, i0:i1, i0:i1]


def beam_convolve(setup, cube):
    """:530 -- zero-padded 'same' convolution of every channel with the beam."""
    return convolve(cube, [setup['gaussbeam']], mode='same')


def peaktounity(cube):
    """:492-497"""
    peak = np.max(cube, axis=(1, 2))
    with np.errstate(all='ignore'):
        scale = 1 / peak
    scale[peak == 0] = 0
    return cube * scale[:, None, None]


def regrid(setup, xoff, yoff, cube):
    """Bilinear resampling onto the observed pixels, 0 outside (:475-483)."""
    out = [None] * len(cube)
    for k, chan in enumerate(cube):
        f = RegularGridInterpolator((setup['yneed'], setup['xneed']), chan,
                                    method='linear', bounds_error=False,
                                    fill_value=0)
        out[k] = f((setup['Y'] - yoff, setup['X'] - xoff))
    return np.array(out)


def scale_factor(setup, cube):
    """Least-squares amplitude, clamped at 0 (:485-490). Returns (s, gf, ff)."""
    gf = np.sum(cube * setup['data_valid'])
    ff = np.sum(cube * cube)
    with np.errstate(all='ignore'):
        s = gf / ff
    if ff == 0 or s < 0:
        s = 0.0
    return s, gf, ff


def cubemodel(setup, p, convolving=True, stages=None, kchunk=None):
    """`ChannelFit.cubemodel` with every parameter group recomputed (:499-537);
    `setup['scaling']` selects 'uniform' or one of the 'mom0*' variants (then
    stages 'Iregrid' = regridded unconvolved cube, 'Iconv' = the same after
    the per-pixel moment-0 scale).  `p` = 13 values in PARAMKEYS order (natural units).
    If `stages` is a dict it receives the intermediate arrays.
    """
    (Mstar, Rc, cs, h1, h2, pI, Rin, Ienv,
     xoff, yoff, voff, incloff, paoff) = [float(q) for q in p]
    with np.errstate(all='ignore'):
        Xnest, Ynest = rot(setup['Xnest0'], setup['Ynest0'],
                           np.radians(paoff))                      # :373-374
        inc = np.radians(setup['incl0'] + incloff)                 # :376-380
        sini, cosi, tani = np.sin(inc), np.cos(inc), np.tan(inc)
        prof, prof_n, prof_d = prof_table(cs, setup['dv'])
        xs = xdisk_surfaces(Xnest, Ynest, sini, cosi, tani, h1, h2)
        vl = [vlos_surface(x_, h_, Ynest, Rc, Rin, sini, setup['signmajor'],
                           setup['signminor'], setup['disk'],
                           setup['envelope'])
              for x_, h_ in zip(xs, (h1, h1, h2, h2))]             # :451-452
        if all(s is None for s in xs):
            nneed = setup['ineed1'] - setup['ineed0']
            Iu = np.zeros((len(setup['v_valid']), nneed, nneed))
        else:
            Iu = iunif(setup, xs, vl, Ynest, prof, prof_n, prof_d,
                       Mstar, Rc, pI, Ienv, voff, kchunk=kchunk)
        if 'mom0' in setup.get('scaling', 'uniform'):              # :520-532
            Io = regrid(setup, xoff, yoff, Iu)
            m0 = np.sum(Io, axis=0) * setup['dv']
            m0[m0 < 0] = np.nan
            Is = np.nan_to_num(Io * setup['mom0decon'] / m0)
            out = beam_convolve(setup, Is) if convolving else peaktounity(Is)
            Ic, Ir, s, gf, ff = Is, Io, np.nan, np.nan, np.nan
        else:
            Ic = beam_convolve(setup, Iu) if convolving else peaktounity(Iu)
            Ir = regrid(setup, xoff, yoff, Ic)
            s, gf, ff = scale_factor(setup, Ir)
            out = Ir * s
    if stages is not None:
        stages.update(prof=prof, prof_d=prof_d, xdisk=xs, vlos=vl, Iunif=Iu,
                      Iconv=Ic, Iregrid=Ir, scale=s, gf=gf, ff=ff)
    return out


def chi2(setup, model):
    """:601-602"""
    return (np.nansum((setup['data_valid'] - model)**2)
            / setup['sigma']**2 / setup['pixperbeam'])


def chi2_of(setup, p, kchunk=None):
    return chi2(setup, cubemodel(setup, p, kchunk=kchunk))


def lnprob_batch(setup, P):
    """-0.5 chi2 for each row of natural-unit parameters P[nw, 13]."""
    return np.array([-0.5 * chi2_of(setup, p) for p in np.atleast_2d(P)])
