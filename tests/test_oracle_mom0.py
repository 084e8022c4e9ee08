"""Pins the oracle's `mom0*` scalings (deconvolved moment-0 maps, forward model,
chi2) to golden vectors recorded from the reference (make_golden_mom0.py)."""
import os

import numpy as np
import pytest

from oracle import channelfit_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = [('A', 'mom0ft'), ('A', 'mom0clean'), ('A', 'mom0model'),
         ('B', 'mom0ft'), ('B', 'mom0clean')]


def load(cfg):
    return (np.load(os.path.join(GOLD, f'channelfit_{cfg}.npz')),
            np.load(os.path.join(GOLD, f'channelfit_mom0_{cfg}.npz')))


def setup_for(f, scaling):
    return O.make_setup(f['in_x'], f['in_y'], f['in_v'], f['in_data'],
                        bmaj=float(f['in_bmaj']), bmin=float(f['in_bmin']),
                        bpa=float(f['in_bpa']), sigma=float(f['in_sigma']),
                        pa=float(f['in_pa']), incl=float(f['in_incl']),
                        rmax=float(f['in_rmax']), vlim=tuple(f['in_vlim']),
                        nlayer=int(f['in_nlayer']), gaussmargin=float(f['in_gaussmargin']),
                        signmajor=int(f['setup_signmajor']),
                        signminor=int(f['setup_signminor']),
                        disk=bool(f['in_disk']), envelope=bool(f['in_envelope']),
                        scaling=scaling)


@pytest.fixture(scope='module')
def setups():
    cache = {}

    def get(cfg, sc):
        if (cfg, sc) not in cache:
            f, m = load(cfg)
            cache[cfg, sc] = (setup_for(f, sc), m)
        return cache[cfg, sc]
    return get


@pytest.mark.parametrize('cfg,sc', CASES)
def test_deconvolved_moment0_bitwise(setups, cfg, sc):
    s, m = setups(cfg, sc)
    assert np.array_equal(s['gaussbeam'], m[f'{sc}_gaussbeam'])
    assert np.array_equal(s['mom0decon'], m[f'{sc}_mom0decon'])
    f, _ = load(cfg)
    mom0, sig0 = O.moment0(s['data_valid'], s['v_valid'], float(f['in_sigma']))
    assert np.array_equal(mom0, m[f'{sc}_mom0']) and sig0 == m[f'{sc}_sigma_mom0']


@pytest.mark.parametrize('cfg,sc', CASES)
def test_model_and_chi2(setups, cfg, sc):
    s, m = setups(cfg, sc)
    for nm in m[f'{sc}_vector_names']:
        pre = f'{sc}_vec_{nm}_'
        st = {}
        model = O.cubemodel(s, m[pre + 'p'], stages=st)
        scale = np.max(np.abs(m[pre + 'model_chsum']))
        assert np.max(np.abs(model.sum(axis=(1, 2)) - m[pre + 'model_chsum'])) <= 1e-12 * scale
        assert O.chi2(s, model) == pytest.approx(float(m[pre + 'chi2']), rel=1e-12)
        if pre + 'model' in m:
            ref = m[pre + 'model']
            assert np.max(np.abs(model - ref)) <= 1e-13 * np.max(np.abs(ref))
            assert np.array_equal(st['Iregrid'], m[pre + 'Iregrid'])
        if pre + 'noconv_model' in m:
            nc = O.cubemodel(s, m[pre + 'p'], convolving=False)
            assert np.array_equal(nc, m[pre + 'noconv_model'])
