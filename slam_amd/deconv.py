"""Deconvolved moment-0 maps for ChannelFit's `mom0*` scalings (host side, once
per fit; reference `channelfit/_channelfit.py:52-210, 343-371`).

These run in `makegrid`, not per walker, and stay in numpy/scipy like the
reference (scipy is imported lazily: the 'uniform' scaling never needs it).
The resulting map is uploaded to the GPU with the other per-fit constants.
"""
from __future__ import annotations

import numpy as np

from .utils import rot


def clean(data, beam, sigma, threshold: float = 2, gain: float = 0.01,
          weakestcomponent: float = 0.3, savetxt=None, loadtxt=None,
          verbose: bool = True):
    """CLEAN of the moment-0 map (:52-87): returns components [Jy/pixel] plus
    the final residual per beam area."""
    from scipy.signal import convolve
    if loadtxt is not None:
        print(f'Load clean components of moment 0 from {loadtxt}.')
        return np.loadtxt(loadtxt)
    beamarea = np.sum(beam)                       # pixel / beam
    components = data * 0
    residual = data * 1
    rms = 10000 * sigma
    last = 10000000 - 1
    for it in range(10000000):
        if it == last:
            print('\n10,000,000 iterations achived in CLEAN.')
            break
        peak = np.nanmax(residual)
        if peak < threshold * sigma:
            if verbose:
                print('\nThreshold achieved in CLEAN. '
                      f'(rms={rms / sigma:.2f}sigma, peak={peak / sigma:.2f}sigma)')
            break
        if verbose:
            print(f'\rCLEAN reached {peak / sigma:.2f}sigma in Moment 0.  ', end='')
        pos = np.unravel_index(np.nanargmax(residual), np.shape(data))
        one = np.zeros_like(data)
        one[pos] = max(gain * residual[pos], weakestcomponent * sigma) / beamarea
        residual = residual - convolve(one, beam, mode='same')
        rms = np.sqrt(np.nanmean(residual**2))
        components = components + one
    out = components + residual / beamarea
    if savetxt is not None:
        np.savetxt(savetxt, out)
    return out


def ftdeconvolve(data, x, y, bmaj, bmin, bpa, sigma, threshold: float = 3):
    """Beam division in the Fourier plane (:180-210)."""
    skipx, skipy = int(len(x) % 2 == 0), int(len(y) % 2 == 0)
    xd, yd = x[skipx:], y[skipy:]
    d = data[skipy:, skipx:]
    ny, nx = np.shape(d)
    nyh, nxh = (ny - 1) // 2, (nx - 1) // 2
    dx, dy = x[1] - x[0], y[1] - y[0]
    xg = np.linspace(-nxh * dx, nxh * dx, nx)
    yg = np.linspace(-nyh * dy, nyh * dy, ny)
    s, t = rot(*np.meshgrid(xg, yg), np.radians(bpa))
    g = np.exp2(-4 * ((t / bmaj)**2 + (s / bmin)**2))
    u = np.fft.fftshift(np.fft.fftfreq(nx, d=dx))
    v = np.fft.fftshift(np.fft.fftfreq(ny, d=dy))
    u, v = np.meshgrid(u, v)
    centre_g = 2 * np.pi * (u * (xg[-1] + dx) + v * (yg[-1] + dy))
    ft_beam = np.real(np.fft.fftshift(np.fft.fft2(g)) * np.exp(-1j * centre_g))
    divisor = ft_beam + 1j * 0
    divisor[ft_beam < np.max(ft_beam) * 1e-2] = 1 + 1j * 0     # no division in the noise
    centre_d = 2 * np.pi * (u * (xd[-1] + dx) + v * (yd[-1] + dy))
    ft_data = np.fft.fftshift(np.fft.fft2(d)) * np.exp(-1j * centre_d)
    ratio = ft_data / divisor
    dnew = np.real(np.fft.ifft2(np.fft.ifftshift(ratio * np.exp(1j * centre_d))))
    dnew[np.abs(d) < sigma * threshold] = 0
    if skipx:
        dnew = np.concatenate((np.zeros((np.shape(dnew)[0], 1)), dnew), axis=1)
    if skipy:
        dnew = np.concatenate((np.zeros((1, np.shape(dnew)[1])), dnew), axis=0)
    return dnew


def modeldeconvolve(data, x, y, bmaj, bmin, bpa, sigma, savetxt=None,
                    loadtxt=None, progressbar: bool = True, niter: int = 20):
    """Half-beam-sampled model whose beam-weighted sums reproduce the map
    (:89-178, iterative branch).  Returns (decon, xmodel, ymodel, zmodel)."""
    from scipy.interpolate import RegularGridInterpolator as RGI
    from scipy.optimize import curve_fit
    nx, ny = len(x), len(y)
    dx, dy = np.abs(x[1] - x[0]), np.abs(y[1] - y[0])
    xskip = int(np.floor(bmin / 2 / dx))
    yskip = int(np.floor(bmaj / 2 / dy))
    nxh = int(np.floor((nx - 1) / 2 / xskip)) * xskip
    nyh = int(np.floor((ny - 1) / 2 / yskip)) * yskip
    nxnew, nynew = 2 * nxh + 1, 2 * nyh + 1
    xoff, yoff = int((nx - nxnew) / 2), int((ny - nynew) / 2)
    xi = x[::-1][xoff:nxnew + xoff]
    yi = y[yoff:nynew + yoff]
    di = data[:, ::-1][yoff:nynew + yoff, xoff:nxnew + xoff]
    Xi, Yi = np.meshgrid(xi, yi)
    Xg = (Xi - xi[nxh]) / (bmin / 2)
    Yg = (Yi - yi[nyh]) / (bmaj / 2)
    g = np.exp2(-Xg**2 - Yg**2)
    gsum = np.sum(g)
    xmodel, ymodel = xi[::xskip], yi[::yskip]
    xnpar, ynpar = len(xmodel), len(ymodel)
    kw = dict(method='linear', bounds_error=False, fill_value=0)
    drot = RGI((yi, xi), di, **kw)(tuple(rot(Xi, Yi, -np.radians(bpa)))[::-1])
    Par0 = drot[::yskip, ::xskip].clip(0, None) / gsum
    if loadtxt is not None:
        popt = np.loadtxt(loadtxt)
        print(f'Load a deconvolved model of moment 0 from {loadtxt}.')
    else:
        bar = None
        if progressbar:
            from tqdm import tqdm
            bar = tqdm(total=niter * ynpar * xnpar)
            bar.set_description('Deconvolution')
        for _ in range(niter):
            for i_p in range(ynpar):
                i_d = i_p * yskip
                for j_p in range(xnpar):
                    j_d = j_p * xskip
                    if bar is not None:
                        bar.update(1)
                    dd = drot[i_d, j_d]
                    if dd < 2 * sigma:
                        Par0[i_p, j_p] = 0
                        continue
                    gg = np.roll(g, (i_d - nyh, j_d - nxh), axis=(0, 1))

                    def model(pts, par):
                        values = Par0 + 0
                        values[i_p, j_p] = par
                        return np.sum(RGI((ymodel, xmodel), values, **kw)(tuple(pts)) * gg)
                    fit, _ = curve_fit(model, [Yi, Xi], dd, p0=Par0[i_p, j_p],
                                       sigma=[sigma], absolute_sigma=True,
                                       bounds=[0, dd])
                    Par0[i_p, j_p] = fit[0]
        if bar is not None:
            bar.close()
        print('')
        popt = np.ravel(Par0)
    if savetxt is not None:
        np.savetxt(savetxt, popt)
    zmodel = np.reshape(popt, (ynpar, xnpar))
    decon = RGI((ymodel, xmodel), zmodel, **kw)(
        tuple(rot(*np.meshgrid(x, y), np.radians(bpa)))[::-1])
    return decon, xmodel, ymodel, zmodel
