#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass (counters given as $2...) over one
# launch group of the headline bench; summary to gpurun_out/pmc_<tag>.txt
# usage: tools/pmc.sh <tag> "<counters>" [bench args]
set -o pipefail
tag=$1; ctr=$2; shift 2
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python bench.py --cpu-evals 0 --cpu-procs 0 --steps 1 --warmup 0 --walkers 128 "$@" > $out/run.log 2>&1 || exit 3
f=$(ls -t $out/*/*_counter_collection.csv | head -1)
python tools/pmcsum.py $f > gpurun_out/pmc_$tag.txt
rm -rf $out/*/
