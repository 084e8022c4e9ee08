import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('evals/s', round(d['value'], 1), 'pipeline frac', round(d['roofline']['pipeline']['frac'], 4),
      'dominant', d['roofline']['kernel'], round(d['roofline']['frac'], 4))
print({k: round(v['ms_per_eval'] * 1e3, 1) for k, v in d['roofline']['per_kernel'].items()}, 'us/eval')
if 'cpu_baseline' in d: print(d['cpu_baseline'])
