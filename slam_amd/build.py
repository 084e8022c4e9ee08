"""Build libslamhip.so and libslampv.so (gfx950) in-tree with hipcc.
`python -m slam_amd.build`.

Every library carries the hash of the sources it was built from (`src_sha`,
sha256 over the .hip and its header, first 16 hex digits) in its version string
and in a sidecar `<lib>.build.json` next to it, together with the compile flags.
A library is rebuilt whenever its sidecar does not describe the current sources
and flags (not on modification times), and `slam_amd._lib.load()` refuses a
library whose embedded hash differs from the sources beside it.

Anything that is not the product is a VARIANT with its OWN output name,
`libslamhip_<variant>.so`, which is only ever loaded when the environment names it
(`SLAMHIP_VARIANT=<variant>`) and which says so in its version string: the
diagnostic build with in-kernel phase stamps (`--stamps` = variant `stamps`,
-DSLAMHIP_PHASE_STAMPS) and A/B builds of a kernel experiment
(`--variant NAME -DFLAG ...`).  A variant can never stand in for the product.
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# name -> (source, header, product output)
TARGETS = {
    'slamhip': (os.path.join(HERE, 'csrc', 'slamhip.hip'), os.path.join(ROOT, 'include', 'slamhip.h'),
                os.path.join(HERE, 'libslamhip.so')),
    'slampv': (os.path.join(HERE, 'csrc', 'slampv.hip'), os.path.join(ROOT, 'include', 'slampv.h'),
               os.path.join(HERE, 'libslampv.so')),
}


def variant_out(variant):
    return os.path.join(HERE, f'libslamhip_{variant}.so')


STAMPS_OUT = variant_out('stamps')
SRC, _, OUT = TARGETS['slamhip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off']


def src_sha(name):
    """Hash of the sources of library `name` ('slamhip' | 'slampv'); None when
    the sources are not there (a tree that ships binaries only)."""
    src, hdr, _ = TARGETS[name]
    h = hashlib.sha256()
    try:
        for p in (src, hdr):
            with open(p, 'rb') as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def sidecar(out):
    return out + '.build.json'


def read_sidecar(out):
    try:
        with open(sidecar(out)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _build_one(name, out, extra, force, verbose, variant='product'):
    src, hdr, _ = TARGETS[name]
    sha = src_sha(name)
    flags = FLAGS + extra
    if variant != 'product':
        flags = flags + [f'-DSLAM_VARIANT="{variant}"']
    want = {'src_sha': sha, 'flags': flags, 'stamps': '-DSLAMHIP_PHASE_STAMPS' in extra,
            'variant': variant}
    if not force and os.path.exists(out) and read_sidecar(out) == want:
        return out
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = ([hipcc] + flags + [f'-DSLAM_SRC_SHA="{sha}"', '-I', os.path.join(ROOT, 'include'), src, '-o', out])
    if verbose:
        print(' '.join(cmd), flush=True)
    if os.path.exists(sidecar(out)):
        os.remove(sidecar(out))           # a failed build leaves no sidecar that vouches for an old file
    subprocess.check_call(cmd)
    with open(sidecar(out), 'w') as f:
        json.dump(want, f)
    return out


def build(force=False, verbose=True, stamps=False, variant=None, defines=()):
    """Product libraries (always checked); with `stamps` ALSO the diagnostic
    libslamhip_stamps.so (k_conv_chi2 phase stamps, read by slamhip_group_counts
    on stderr); with `variant` ALSO libslamhip_<variant>.so compiled with `defines`.
    Variants are loaded only under SLAMHIP_VARIANT=<name>."""
    outs = [_build_one(n, t[2], [], force, verbose) for n, t in TARGETS.items()]
    if stamps:
        _build_one('slamhip', STAMPS_OUT, ['-DSLAMHIP_PHASE_STAMPS'], force, verbose, 'stamps')
    if variant:
        if variant in ('product', 'stamps') or not variant.isidentifier():
            raise ValueError(f'bad variant name {variant!r}')
        _build_one('slamhip', variant_out(variant), list(defines), force, verbose, variant)
    return outs[0]


if __name__ == '__main__':
    av = sys.argv[1:]
    var = av[av.index('--variant') + 1] if '--variant' in av else None
    build(force='--force' in av, stamps='--stamps' in av, variant=var,
          defines=[a for a in av if a.startswith('-D')])
