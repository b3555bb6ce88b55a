for v in "" _tnnopieces _tnnomfma; do
  IDN_LIB=$PWD/ideal-nerf_amd/libidealnerf$v.so python bench.py --workload train --steps 10 --warmup 3 > gpurun_out/tn$v.json 2>/dev/null
  python - "$v" <<PY
import json,sys
d=json.load(open("gpurun_out/tn%s.json"%sys.argv[1]))
print(sys.argv[1] or "base", round(d["ms_per_step"],2), {k:round(v["ms_per_step"],3) for k,v in d["roofline"]["kernels"].items()})
PY
done
