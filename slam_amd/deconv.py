"""Deconvolved moment-0 maps for ChannelFit's `mom0*` scalings: host side, once
per fit, numpy/scipy (imported lazily: the 'uniform' scaling never needs scipy).

Three methods, selected by `ChannelFit(scaling=...)`; each returns a map on the
observed (y, x) grid in Jy/pixel that `makegrid` uploads with the other per-fit
constants.  They restate what the reference computes at
`channelfit/_channelfit.py:52-87` (CLEAN), `:89-178` (coarse-grid model, the
iterative branch `makegrid` uses) and `:180-210` (Fourier division); the maps are
bit-equal to the reference's on the golden configurations (tests/test_mom0.py),
which pins the order of the floating-point operations below.
"""
from __future__ import annotations

import numpy as np

from .utils import rot

_CLEAN_MAX_ITER = 10000000


def clean(data, beam, sigma, threshold: float = 2, gain: float = 0.01,
          weakestcomponent: float = 0.3, savetxt=None, loadtxt=None,
          verbose: bool = True):
    """Hogbom-style CLEAN of the moment-0 map.  Peak by peak, a component of
    max(gain * peak, weakestcomponent * sigma) / beam area moves from the
    residual to the model until the residual peak is under threshold * sigma;
    the last residual, per beam area, is added back."""
    from scipy.signal import convolve
    if loadtxt is not None:
        print(f'Load clean components of moment 0 from {loadtxt}.')
        return np.loadtxt(loadtxt)
    area = np.sum(beam)
    model = data * 0
    resid = data * 1
    for it in range(_CLEAN_MAX_ITER):
        peak = np.nanmax(resid)
        if peak < threshold * sigma or it == _CLEAN_MAX_ITER - 1:
            if verbose:
                rms = np.sqrt(np.nanmean(resid**2))
                print(f'\nCLEAN stopped after {it} components: peak {peak / sigma:.2f} sigma, '
                      f'rms {rms / sigma:.2f} sigma.')
            break
        if verbose:
            print(f'\rCLEAN: residual peak {peak / sigma:.2f} sigma  ', end='')
        at = np.unravel_index(np.nanargmax(resid), data.shape)
        comp = np.zeros_like(data)
        comp[at] = max(gain * resid[at], weakestcomponent * sigma) / area
        resid = resid - convolve(comp, beam, mode='same')
        model = model + comp
    out = model + resid / area
    if savetxt is not None:
        np.savetxt(savetxt, out)
    return out


def ftdeconvolve(data, x, y, bmaj, bmin, bpa, sigma, threshold: float = 3):
    """Division by the beam in the Fourier plane: an odd-sized crop of the map
    (the first row / column of an even axis is set aside and returned as 0),
    the beam's transform made real by re-centring and floored (set to 1) below
    1 % of its peak, pixels under threshold * sigma zeroed afterwards."""
    cx, cy = int(len(x) % 2 == 0), int(len(y) % 2 == 0)
    xd, yd, d = x[cx:], y[cy:], data[cy:, cx:]
    ny, nx = d.shape
    hx, hy = (nx - 1) // 2, (ny - 1) // 2
    dx, dy = x[1] - x[0], y[1] - y[0]
    xg = np.linspace(-hx * dx, hx * dx, nx)
    yg = np.linspace(-hy * dy, hy * dy, ny)
    s, t = rot(*np.meshgrid(xg, yg), np.radians(bpa))
    kernel = np.exp2(-4 * ((t / bmaj)**2 + (s / bmin)**2))
    u, w = np.meshgrid(np.fft.fftshift(np.fft.fftfreq(nx, d=dx)),
                       np.fft.fftshift(np.fft.fftfreq(ny, d=dy)))

    def centred(a, xa, ya):
        """Transform of `a` and the phase that moves its origin to the grid centre."""
        ph = 2 * np.pi * (u * (xa[-1] + dx) + w * (ya[-1] + dy))
        return np.fft.fftshift(np.fft.fft2(a)) * np.exp(-1j * ph), ph

    fk = np.real(centred(kernel, xg, yg)[0])
    div = fk + 1j * 0
    div[fk < np.max(fk) * 1e-2] = 1 + 1j * 0          # no division where the beam has no power
    fd, ph = centred(d, xd, yd)
    out = np.real(np.fft.ifft2(np.fft.ifftshift(fd / div * np.exp(1j * ph))))
    out[np.abs(d) < sigma * threshold] = 0
    if cx:
        out = np.concatenate((np.zeros((out.shape[0], 1)), out), axis=1)
    if cy:
        out = np.concatenate((np.zeros((1, out.shape[1])), out), axis=0)
    return out


def modeldeconvolve(data, x, y, bmaj, bmin, bpa, sigma, savetxt=None,
                    loadtxt=None, progressbar: bool = True, niter: int = 20):
    """Coarse-grid deconvolution.  The map is rotated into the beam frame and
    sampled every half beam; sweep after sweep (`niter`), each sample at or
    above 2 sigma is refitted so that the bilinear model times the beam centred
    on it reproduces the rotated map there, the others are zeroed.  Returns
    (map on (y, x), model x axis, model y axis, model samples)."""
    from scipy.interpolate import RegularGridInterpolator
    from scipy.optimize import curve_fit
    nx, ny = len(x), len(y)
    dx, dy = np.abs(x[1] - x[0]), np.abs(y[1] - y[0])
    sx, sy = int(np.floor(bmin / 2 / dx)), int(np.floor(bmaj / 2 / dy))
    hx = int(np.floor((nx - 1) / 2 / sx)) * sx
    hy = int(np.floor((ny - 1) / 2 / sy)) * sy
    mx, my = 2 * hx + 1, 2 * hy + 1
    ox, oy = int((nx - mx) / 2), int((ny - my) / 2)
    xi = x[::-1][ox:mx + ox]                          # ascending, an odd number of pixels
    yi = y[oy:my + oy]
    di = data[:, ::-1][oy:my + oy, ox:mx + ox]
    Xi, Yi = np.meshgrid(xi, yi)
    g = np.exp2(-((Xi - xi[hx]) / (bmin / 2))**2 - ((Yi - yi[hy]) / (bmaj / 2))**2)
    xm, ym = xi[::sx], yi[::sy]
    lin = dict(method='linear', bounds_error=False, fill_value=0)
    drot = RegularGridInterpolator((yi, xi), di, **lin)(
        tuple(rot(Xi, Yi, -np.radians(bpa)))[::-1])
    par = drot[::sy, ::sx].clip(0, None) / np.sum(g)
    if loadtxt is not None:
        flat = np.loadtxt(loadtxt)
        print(f'Load a deconvolved model of moment 0 from {loadtxt}.')
    else:
        bar = None
        if progressbar:
            from tqdm import tqdm
            bar = tqdm(total=niter * par.size)
            bar.set_description('Deconvolution')
        for _ in range(niter):
            for ip in range(len(ym)):
                for jp in range(len(xm)):
                    if bar is not None:
                        bar.update(1)
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
        if bar is not None:
            bar.close()
        flat = np.ravel(par)
    if savetxt is not None:
        np.savetxt(savetxt, flat)
    zm = np.reshape(flat, (len(ym), len(xm)))
    decon = RegularGridInterpolator((ym, xm), zm, **lin)(
        tuple(rot(*np.meshgrid(x, y), np.radians(bpa)))[::-1])
    return decon, xm, ym, zm
