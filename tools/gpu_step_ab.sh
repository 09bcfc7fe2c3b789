#!/bin/bash
# A/B of whole training steps on one box: interleaved bench.py runs with different environment switches
#   tools/gpu_step_ab.sh <outdir> "<env A>" "<env B>" ...   (an env spec is e.g. "SAIS_TN_XL=0"; "-" = no switch)
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2; do
  i=0
  for spec in "$@"; do
    i=$((i+1))
    [ "$spec" = "-" ] && spec="SAIS_DUMMY=1"
    env $spec timeout 400 python bench.py --no-cpu-baseline --sustain-seconds 0 --steps 30 --parity-clips 0 > $out/bench_${i}_$rep.json 2> $out/bench_${i}_$rep.err
    echo "$spec rep $rep: $(python -c "import json,sys; d=json.loads(open('$out/bench_${i}_$rep.json').read().strip().splitlines()[-1]); ak=d['roofline']['all_kernels']; print(d['ms_per_step'], d['value'], ' '.join(f'{k}={v[chr(97)+chr(118)+chr(103)+chr(95)+chr(117)+chr(115)]}' for k,v in ak.items() if v['ms_per_step']>0.3))" 2>&1 | tail -1)" >> $out/summary.txt
  done
done
cat $out/summary.txt
