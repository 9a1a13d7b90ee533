#!/bin/bash
# samples rocm-smi power / clocks while the bench runs a long timed region: scripts/power_probe.sh <precision>
P=${1:-f16m8}
( for i in $(seq 1 14); do /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_$P.txt &
SMI=$!
python bench.py --precision $P --selfplay-seconds 0 --no-cpu-baseline --no-host-path --steps 1500 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$P', round(d['value']), round(d['ms_per_step'],3))"
wait $SMI
tail -8 gpurun_out/power_$P.txt | cut -c1-400
