"""Build libslamhip.so and libslampv.so (gfx950) in-tree with hipcc.
`python -m slam_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# (source, header, output)
TARGETS = [
    (os.path.join(HERE, 'csrc', 'slamhip.hip'), os.path.join(ROOT, 'include', 'slamhip.h'),
     os.path.join(HERE, 'libslamhip.so')),
    (os.path.join(HERE, 'csrc', 'slampv.hip'), os.path.join(ROOT, 'include', 'slampv.h'),
     os.path.join(HERE, 'libslampv.so')),
]
SRC, _, OUT = TARGETS[0]


def build(force=False, verbose=True, stamps=False):
    """`stamps`: diagnostic build of libslamhip.so with in-kernel phase stamps
    (k_conv_chi2; read by slamhip_group_counts on stderr).  Never the product."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    outs = []
    for src, hdr, out in TARGETS:
        outs.append(out)
        if (not force and os.path.exists(out)
                and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in (src, hdr))):
            continue
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
               '-ffp-contract=off', '-I', os.path.join(ROOT, 'include'), src, '-o', out]
        if stamps and src.endswith('slamhip.hip'):
            cmd.insert(1, '-DSLAMHIP_PHASE_STAMPS')
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return outs[0]


if __name__ == '__main__':
    build(force='--force' in sys.argv or '--stamps' in sys.argv, stamps='--stamps' in sys.argv)
