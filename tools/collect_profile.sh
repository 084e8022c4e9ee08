#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel stats + HBM traffic
# counters (separate --pmc passes, as MI355X_MICROARCH.md's HBM section prescribes).
# usage: tools/collect_profile.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py "$@" > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --cpu-evals 0 "$@" > $out/trace.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --cpu-evals 0 --steps 1 --warmup 0 "$@" > $out/fetch.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --cpu-evals 0 --steps 1 --warmup 0 "$@" > $out/write.log 2>&1 || exit 4
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2 -- python bench.py --cpu-evals 0 --steps 1 --warmup 0 "$@" > $out/l2.log 2>&1 || exit 5
echo done
