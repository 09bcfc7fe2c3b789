#!/bin/bash
tag=${1:-r3h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 20 --warmup 3"
for rep in 1 2 3; do for sd in 0 1; do SAIS_TEMPORAL_SIDE=$sd $B > $O/b.json 2> $O/b.err; echo "side=$sd $(head -c 200 $O/b.json | tail -c 60)"; done; done
for sd in 0 1; do SAIS_TEMPORAL_SIDE=$sd $B --two-stream > $O/b.json 2> $O/b.err; echo "two-stream side=$sd $(head -c 200 $O/b.json | tail -c 60)"; done
