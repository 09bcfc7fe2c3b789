#!/bin/bash
# Clock / power of the GPU while bench.py replays the training step for 25 s (is the step power-bound?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-power}
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --sustain-seconds 25 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12
for i in $(seq 1 14); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)|GPU use" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1.5
done > $O/smi.log 2>&1
wait $BP
head -c 200 $O/bench.json; echo
cat $O/smi.log | cut -c1-400
/opt/rocm/bin/rocm-smi --showmaxpower --showclocks 2>/dev/null | grep -E "Max|sclk" | head -5
