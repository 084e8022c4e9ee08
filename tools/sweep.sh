#!/bin/bash
# Run ON THE GPU BOX (via gpurun): secondary bench lines -- sampler-sized batches and
# dense-sky workloads -- into gpurun_out/sweep_<tag>.jsonl (one JSON line per run).
# usage: tools/sweep.sh <tag>
set -o pipefail
tag=$1
out=gpurun_out/sweep_$tag.jsonl
: > $out
run() { timeout -k 10 500 python bench.py --cpu-evals 0 --cpu-procs 0 "$@" 2>> gpurun_out/sweep_$tag.err | grep '^{' >> $out || exit 1; }
for w in 5 10 20 40 80 256 1024; do run --walkers $w --steps 10 --warmup 3; done
run --workload dense_disk --walkers 256 --steps 2 --warmup 1
run --workload envelope --walkers 256 --steps 2 --warmup 1
run --config C2 --walkers 64 --steps 10 --warmup 2
run --bpa 20 --walkers 1024 --steps 2 --warmup 1
run --bpa 20 --beam-tol 1e-5 --walkers 1024 --steps 2 --warmup 1
run --bpa 20 --beam-tol 1e-7 --walkers 1024 --steps 2 --warmup 1
run --dtype f32 --walkers 1024 --steps 3 --warmup 1
run --dtype f32g --walkers 1024 --steps 3 --warmup 1
run --dtype f32 --workload dense_disk --walkers 256 --steps 2 --warmup 1
run --scaling mom0ft --walkers 1024 --steps 2 --warmup 1
run --config P5 --walkers 1024 --steps 5 --warmup 2
run --config C4 --walkers 4096 --steps 1 --warmup 1
run --config C4 --dtype f32 --walkers 4096 --steps 1 --warmup 1
echo done
