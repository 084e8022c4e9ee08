"""Turn the rocprofv3 outputs of tools/collect_profile.sh into profiles/<tag>_*.

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  MI355X_MICROARCH.md (HBM
section): on gfx950 FETCH_SIZE reads exactly half of a wide (16 B/lane)
coalesced stream; other widths are uncalibrated -> the script calibrates on
dispatches of known size in the same run (hipMemcpy D2D copy kernels read
and write `bytes` each; the fill kernel writes `bytes`) and reports both the
raw and the calibrated figure.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', name)


def counter_rows(d, counter):
    f = max(glob.glob(os.path.join(d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            per[short(r['Kernel_Name'])].append(float(r['Counter_Value']) * 1024.0)
    return per


def counter_rows_raw(d, counter):
    f = max(glob.glob(os.path.join(d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            per[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return per


def main():
    tag = sys.argv[1]
    src = f'gpurun_out/prof_{tag}'
    os.makedirs('profiles', exist_ok=True)
    stats = max(glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv')), key=os.path.getmtime)
    shutil.copy(stats, f'profiles/{tag}_kernel_stats.csv')
    bench = [l for l in open(os.path.join(src, 'bench.json')) if l.startswith('{')][-1]
    open(f'profiles/{tag}_bench.json', 'w').write(bench)
    fetch = counter_rows(os.path.join(src, 'fetch'), 'FETCH_SIZE')
    write = counter_rows(os.path.join(src, 'write'), 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        out[k] = {'launches': max(len(f), len(w)),
                  'fetch_bytes_per_launch_raw': sum(f) / max(len(f), 1),
                  'fetch_bytes_per_launch_max': max(f) if f else 0.0,
                  'write_bytes_per_launch': sum(w) / max(len(w), 1),
                  'write_bytes_per_launch_max': max(w) if w else 0.0}
    b = json.loads(bench)
    w = b['config']['workload']          # 'SYN-v1 C3 (sparse, f64): ... bpa=0 (beam rank 1)'
    m = re.match(r'SYN-v1 (\w+) \((\w+), (\w+)\).* bpa=([-\d.e+]+) ', w)
    scaling = re.search(r'scaling=(\w+)', b['metric'])
    out['_meta'] = {'config': m.group(1), 'workload': m.group(2), 'dtype': m.group(3),
                    'bpa': float(m.group(4)), 'scaling': scaling.group(1) if scaling else 'uniform',
                    'walkers_per_launch': b['config']['walkers_per_launch'],
                    # the build the counters were collected with: bench.py refuses the profile
                    # for any other (slam_amd/build.py src_sha)
                    'src_sha': b['config']['build'].split()[0],
                    'fetch_correction': 'FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request); '
                                        'Infinity-Cache hits are included (upper bound on HBM bytes)'}
    hm = os.path.join(src, 'l2')
    if os.path.isdir(hm):                # optional third pass: L2 hit / miss requests per kernel
        hit, miss = counter_rows_raw(hm, 'TCC_HIT_sum'), counter_rows_raw(hm, 'TCC_MISS_sum')
        for k in out:
            if k in hit and k in miss and not k.startswith('_'):
                h_, m_ = sum(hit[k]), sum(miss[k])
                out[k]['l2_hit_rate'] = h_ / max(h_ + m_, 1.0)
    json.dump(out, open(f'profiles/{tag}_traffic.json', 'w'), indent=1)
    for k, v in out.items():
        if k.startswith('_'):
            continue
        print(f"{k:28s} n={v['launches']:4d} fetch_raw={v['fetch_bytes_per_launch_max']/1e6:10.2f} MB"
              f"  write={v['write_bytes_per_launch_max']/1e6:10.2f} MB (max per launch)")


if __name__ == '__main__':
    main()
