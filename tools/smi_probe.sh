#!/bin/bash
# sample clocks/power while the dominant conv shape runs back to back
python tools/conv_bench.py --iters 20000 --shapes main > gpurun_out/smi_cb.txt 2>&1 &
PID=$!
sleep 10
for i in 1 2 3 4; do
  timeout 10 rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" >> gpurun_out/smi.txt
  echo --- >> gpurun_out/smi.txt
  sleep 0.5
done
wait $PID
echo idle >> gpurun_out/smi.txt
timeout 10 rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" >> gpurun_out/smi.txt
