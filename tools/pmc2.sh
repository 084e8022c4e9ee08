#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass per counter group ("a b c" "d e" ...) over one
# 128-walker launch group of the headline bench; per-kernel summaries to gpurun_out/pmc_<tag>.txt
# usage: tools/pmc2.sh <tag> "<bench args>" "<counters pass 1>" ["<counters pass 2>" ...]
set -o pipefail
tag=$1; bargs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/pmc_$tag.txt
i=0
for ctr in "$@"; do
  i=$((i+1)); out=gpurun_out/pmc_${tag}_$i
  rm -rf $out; mkdir -p $out
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python bench.py --cpu-evals 0 --cpu-procs 0 --steps 1 --warmup 1 --walkers 128 $bargs > $out/run.log 2>&1 || { echo "pass $i failed"; tail -3 $out/run.log; exit 3; }
  f=$(ls -t $out/*/*_counter_collection.csv | head -1)
  python tools/pmcsum.py $f | grep -A12 "^k_conv\|^k_profile<\|^k_surface<" >> gpurun_out/pmc_$tag.txt
  rm -rf $out
done
cat gpurun_out/pmc_$tag.txt
