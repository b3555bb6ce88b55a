#!/bin/bash
# Same-box A/B of several library builds on the train step (BASELINE configs[2]), arms alternating.
#   tools/ab_train_arms.sh <rounds> <tag|-> [<tag|-> ...]     ('-' = the shipped libidealnerf.so)
R=$1; shift
lib() { if [ "$1" = "-" ]; then echo "$PWD/ideal-nerf_amd/libidealnerf.so"; else echo "$PWD/ideal-nerf_amd/libidealnerf_$1.so"; fi; }
for rep in $(seq $R); do
  for arm in "$@"; do
    IDN_LIB=$(lib $arm) timeout -k 10 200 python bench.py --workload train --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('arm %-10s %.3f ms/step  frac %.4f ' % ('$arm', d['ms_per_step'], d['roofline']['frac']), {k:round(v['ms_per_step'],3) for k,v in d['roofline']['kernels'].items()}, flush=True)"
  done
done
