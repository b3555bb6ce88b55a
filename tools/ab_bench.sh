#!/bin/bash
# A/B two builds of the library on the SAME GPU box (devices differ by several % in MFMA-bound
# wall time, so arms must never be compared across gpurun calls).  Usage:
#   tools/ab_bench.sh <tagA|-> <tagB> [precision ...]      ('-' = the shipped libidealnerf.so)
# Alternates A,B,A,B and prints samples/s per arm.
set -e
A=$1; B=$2; shift 2
PRECS=${@:-bf16x3}
mkdir -p gpurun_out
lib() { if [ "$1" = "-" ]; then echo "$PWD/ideal-nerf_amd/libidealnerf.so"; else echo "$PWD/ideal-nerf_amd/libidealnerf_$1.so"; fi; }
for p in $PRECS; do
  for rep in 1 2; do
    for arm in $A $B; do
      IDN_LIB=$(lib $arm) timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline --steps 8 > gpurun_out/ab_tmp.json 2>gpurun_out/ab_err.log
      python - "$p" "$arm" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:7s} arm {sys.argv[2]:10s} {d['value']:.4e} samples/s  frac {d['roofline']['frac']:.4f}  kernel avg {d['roofline']['avg_launch_ms']:.4f} ms", flush=True)
PY
    done
  done
done
