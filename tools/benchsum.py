import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('evals/s', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 2),
      'dominant', r['kernel'], 'frac', round(r['frac'], 4),
      'pipeline frac', round(r['pipeline']['frac'], 4) if 'pipeline' in r else '-')
for k, v in r['per_kernel'].items():
    print(f"  {k:12s} {v['ms_per_launch']:.4f} ms/launch x{v['launches']}  {v['ms_per_eval']*1e3:8.2f} us/eval  share {v['share']:.3f}")
if 'cpu_baseline' in d:
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['unit'], 'err', d['cpu_baseline'].get('max_rel_err_vs_gpu'))
