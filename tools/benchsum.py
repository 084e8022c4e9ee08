"""Print the lines of one or more bench JSON files (last '{' line of each, or every
line of a .jsonl) as a table: evals/s, per-kernel us/eval, roofline fractions."""
import json
import sys


def show(d):
    r = d['roofline']
    print(f"{d['config']['workload']}")
    print(f"  evals/s {d['value']:.1f}  ms/step {d['ms_per_step']:.2f}  bound {r.get('bound')}"
          f"  frac {r.get('frac')}  AI {r.get('arithmetic_intensity_flop_per_byte')}  traffic_frac {r.get('traffic_frac')}"
          f"  dense-bytes x peak {r.get('algorithmic_x_peak')}")
    for k, v in r['per_kernel'].items():
        extra = ''
        if 'hbm_frac' in v:
            extra += f"  hbm {v['hbm_frac']:.3f}"
        if 'valu_frac' in v:
            extra += f"  valu {v['valu_frac']:.3f}"
        print(f"    {k:12s} {v['ms_per_launch']:9.4f} ms/launch x{v['launches']:<4d} "
              f"{v['ms_per_eval'] * 1e3:8.2f} us/eval  share {v['share']:.3f}{extra}")
    if 'cpu_baseline' in d:
        c = d['cpu_baseline']
        print('  cpu', c['value'], c['unit'], 'cores', c['cores'], 'err', c.get('max_rel_err_vs_gpu'),
              'all_cores', c.get('all_cores', {}).get('value'))


for path in sys.argv[1:]:
    lines = [l for l in open(path) if l.startswith('{')]
    if not path.endswith('.jsonl'):
        lines = lines[-1:]
    for l in lines:
        show(json.loads(l))
