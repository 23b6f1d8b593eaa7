# Shader clock and package power while bench.py runs (does the chip sit at a power limit under the split-16 kernels?)
mkdir -p gpurun_out/power_probe
( for i in $(seq 1 60); do date +%s.%N | cut -c1-14 | tr '\n' ' '; rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "GPU\[" | sed 's/GPU\[\([0-9]*\)\]\s*:\s*/g\1 /' | grep -i "sclk\|Power\|use" | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/power_probe/smi.txt &
SMI=$!
sleep 2
python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-legs > gpurun_out/power_probe/bench.json 2>/dev/null
date +%s.%N | cut -c1-14 > gpurun_out/power_probe/end.txt
wait $SMI
python -c "
import json;d=json.load(open('gpurun_out/power_probe/bench.json'));print(d['value'],d['ms_per_step'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i "GPU\[" | head -3
sed -n '1,60p' gpurun_out/power_probe/smi.txt | cut -c1-220 | awk 'NR%4==1'
