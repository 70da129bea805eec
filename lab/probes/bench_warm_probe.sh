#!/bin/bash
# how much of the gap between the driver's --steps 20 --warmup 5 and a long run is clock ramp (warmup-dependent)
# and how much is the fixed cost of the timed region (steps-dependent)
for rep in 1 2 3; do
for cfg in "20 5" "20 200" "20 2000" "200 5" "200 200" "2000 20"; do
  set -- $cfg
  python bench.py --steps $1 --warmup $2 --no-extras --no-shares --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('steps %5d warmup %5d: value %.0f  ms_per_step %.4f  kernel_us %.2f' % (o['steps'], o['warmup'], o['value'], o['ms_per_step'], o['roofline']['kernel_us']))"
done
done
