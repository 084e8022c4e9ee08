#!/bin/bash
# Run ON THE GPU BOX (via gpurun): A/B of build variants on one box, back to back.
# usage: tools/ab.sh <tag> "<variant> <variant> ..." [bench args...]
#   variant 'product' = libslamhip.so; others = libslamhip_<variant>.so (python -m slam_amd.build --variant ...)
# Two rounds (A B A B) so that drift of the box shows; summary lines on stdout.
set -o pipefail
tag=$1; variants=$2; shift 2
out=gpurun_out/ab_$tag
mkdir -p $out
for round in 1 2; do
  for v in $variants; do
    f=$out/${v}_$round.json
    if [ "$v" = product ]; then unset SLAMHIP_VARIANT; else export SLAMHIP_VARIANT=$v; fi
    timeout -k 10 200 python bench.py --cpu-evals 0 --cpu-procs 0 --steps 5 --warmup 2 "$@" > $f 2> $out/${v}_$round.err || { echo "FAILED $v"; tail -5 $out/${v}_$round.err; exit 1; }
    echo "== $v (round $round)"; python tools/benchsum.py $f | grep -v "^  cpu" | cut -c1-150
  done
done
