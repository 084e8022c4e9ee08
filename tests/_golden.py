"""Helpers shared by the tests: load golden fixtures and rebuild the oracle's
setup dict from a fixture's recorded inputs."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CONFIGS = ['A', 'B', 'C', 'D', 'E', 'F']


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, f'channelfit_{name}.npz'))


def inputs(g):
    """kwargs for oracle.make_setup / slam_amd ChannelFit from a fixture."""
    given = bool(g['in_signs_given'])
    return dict(
        x=g['in_x'], y=g['in_y'], v=g['in_v'], data=g['in_data'],
        bmaj=float(g['in_bmaj']), bmin=float(g['in_bmin']),
        bpa=float(g['in_bpa']), sigma=float(g['in_sigma']),
        pa=float(g['in_pa']), incl=float(g['in_incl']),
        rmax=float(g['in_rmax']), vlim=tuple(g['in_vlim']),
        nlayer=int(g['in_nlayer']), gaussmargin=float(g['in_gaussmargin']),
        signmajor=float(g['setup_signmajor']) if given else None,
        signminor=float(g['setup_signminor']) if given else None,
        disk=bool(g['in_disk']), envelope=bool(g['in_envelope']))


def vectors(g):
    return [str(n) for n in g['vector_names']]


def relmax(a, b):
    """max|a-b| / max|b| -- the cube parity measure."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    den = np.max(np.abs(b))
    if den == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / den)
