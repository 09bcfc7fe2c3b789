#!/bin/bash
# Board power / clocks while the training step replays, per environment spec, on one box:  tools/gpu_power_ab.sh <out> "<env A>" "<env B>" ...
out=gpurun_out/$1; shift
mkdir -p $out
i=0
for rep in 1 2; do
for spec in "$@"; do
  i=$((i+1))
  env $spec python bench.py --no-cpu-baseline --sustain-seconds 14 --steps 20 --warmup 3 --parity-clips 0 > $out/bench_$i.json 2> $out/bench_$i.err &
  BP=$!
  sleep 9
  for k in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Sensor junction" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 1
  done > $out/smi_$i.log 2>&1
  wait $BP
  echo "$spec rep $rep: $(python -c "import json; d=json.loads(open('$out/bench_$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['frames_per_s'])")" >> $out/summary.txt
  grep -o "Power[^;]*;\|sclk[^;]*;\|junction[^;]*;" $out/smi_$i.log | tr '\n' ' ' | cut -c1-700 >> $out/summary.txt
  echo >> $out/summary.txt
done
done
cat $out/summary.txt
