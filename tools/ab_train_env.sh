#!/bin/bash
# Same-box A/B of the train step (BASELINE configs[2]) under environment switches of ONE library build, arms alternating.
#   tools/ab_train_env.sh <rounds> "<VAR=val ...>" ["<VAR=val ...>" ...]      ('' = no switch)
R=$1; shift
for rep in $(seq $R); do
  for arm in "$@"; do
    env $arm timeout -k 10 200 python bench.py --workload train --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('arm %-44s %.3f ms/step  frac %.4f ' % ('[$arm]', d['ms_per_step'], d['roofline']['frac']), {k:round(v['ms_per_step'],3) for k,v in d['roofline']['kernels'].items()}, flush=True)"
  done
done
