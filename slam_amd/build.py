"""Build libslamhip.so (gfx950) in-tree with hipcc.  `python -m slam_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc', 'slamhip.hip')
OUT = os.path.join(HERE, 'libslamhip.so')


def build(force=False, verbose=True):
    deps = [SRC, os.path.join(ROOT, 'include', 'slamhip.h')]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-ffp-contract=off', '-I', os.path.join(ROOT, 'include'), SRC, '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
