#!/bin/bash
# Copy what `tools/final_round.sh {tests|profiles}` left under gpurun_out/final_* into profiles/<round>_* (run here, after the gpurun call).
#   ROUND=r04 bash tools/stage_final.sh
R=${ROUND:-r04}
F=gpurun_out/final_profiles; T=gpurun_out/final_tests
if [ -d $F ]; then
  for P in f32 bf16x6; do cp $F/pmc_mlp_${P}_final.json profiles/${R}_pmc_mlp_${P}_final.json; cp $F/kernel_stats_${P}_final.csv profiles/${R}_kernel_stats_${P}_final.csv; done
  for P in render_fused render_split; do cp $F/pmc_${P}_final.json profiles/${R}_pmc_${P}_final.json; cp $F/kernel_stats_${P}_final.csv profiles/${R}_kernel_stats_${P}_final.csv; done
  cp $F/kernel_stats_train_final.csv profiles/${R}_kernel_stats_train_final.csv
  for j in $F/pmc_train_*.json; do cp $j profiles/${R}_$(basename $j); done
  for b in default bf16x6 train torso_bf16 n2_gloo_rehearsal train_n2_gloo_rehearsal; do cp $F/bench_$b.json profiles/${R}_bench_$b.json; done
  [ -f $F/ab_x6_stream_wrap.log ] && cp $F/ab_x6_stream_wrap.log profiles/${R}_ab_x6_stream_wrap.log
fi
if [ -d $T ]; then
  cp $T/pytest_gpu_final.log profiles/${R}_pytest_gpu_final.log
  cp $T/pytest_gpu_f32_as_default.log profiles/${R}_pytest_gpu_f32_as_default.log
  cp $T/pytest_gpu_fused_as_default.log profiles/${R}_pytest_gpu_fused_as_default.log
fi
python3 - <<'P'
import glob, json, os
r = os.environ.get("ROUND", "r04")
for f in sorted(glob.glob(f"profiles/{r}_pmc_*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), d.get("kernel_source_sha16"), d.get("commit"))
P
