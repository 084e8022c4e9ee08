"""slam_amd: MI355X-native batched likelihood for SLAM's channelfit.

Host API mirrors the reference (`ChannelFit`, `emcee_corner`); the hot path is
hand-written HIP for gfx950 behind the C ABI in include/slamhip.h.
"""
from ._lib import SlamhipError  # noqa: F401
from .engine import HipLikelihood, PARAMKEYS  # noqa: F401

__all__ = ['HipLikelihood', 'PARAMKEYS', 'SlamhipError']
