#!/bin/bash
# End-of-round measurements on ONE GPU box (one gpurun call; ~12 min): the GPU test log (also with the fp32 kernel as the module
# default, and with the fused ray kernel as the default arrangement), the PMC / kernel-trace summaries of the fp32 and bf16x6 inference kernels and of the train step, the same-box
# stream-wrap A/B of the shipped bf16x6 kernel, and the bench lines -- all into gpurun_out/final_<part>/, from where they are
# copied to profiles/r<NN>_* by hand.
#   ROUND=r04 IDN_COMMIT=$(git rev-parse --short HEAD) bash tools/final_round.sh        (the GPU box has no .git)
#   (build the A/B arm first: python ideal-nerf_amd/build.py --variant wrap "-DIDN_TIMING_STREAM_WRAP=8" mlp_bf16x6.hip)
#   Two gpurun calls (a call is limited to 20 minutes): `bash tools/final_round.sh tests` (the three suites, ~14 min) and
#   `bash tools/final_round.sh profiles` (rocprofv3 summaries + bench lines, ~10 min; run it LAST: bench.py reports `traffic` only
#   from PMC summaries taken on the sources that are running).
set -e
export TMPDIR=/tmp
PART=${1:-tests}
F=gpurun_out/final_$PART
rm -rf $F; mkdir -p $F
if [ "$PART" = "tests" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $F/pytest_gpu_final.log 2>&1
tail -1 $F/pytest_gpu_final.log
# the shipped module default is bf16x6: the same suite with the fp32 MFMA kernel as the default
IDN_DEFAULT_PRECISION=f32 timeout -k 10 900 python -m pytest tests -m gpu -q > $F/pytest_gpu_f32_as_default.log 2>&1
tail -1 $F/pytest_gpu_f32_as_default.log
IDN_FUSED_MARCH=1 timeout -k 10 900 python -m pytest tests -m gpu -q > $F/pytest_gpu_fused_as_default.log 2>&1
tail -1 $F/pytest_gpu_fused_as_default.log
exit 0
fi
# the fused ray kernel (the opt-in arrangement of the fp32 path) first: its summaries are taken from the same output directory
IDN_FUSED_MARCH=1 IDN_PROFILE_KERNEL=render_fused_kernel bash tools/profile_round.sh f32 > $F/profile_render_fused.log 2>&1
cp gpurun_out/prof_f32/pmc_summary.json $F/pmc_render_fused_final.json
cp gpurun_out/prof_f32/kernel_stats.csv $F/kernel_stats_render_fused_final.csv
echo "profile render_fused done"
# ... and as two launches (coarse network + march | fine network + compositing): both instances of the kernel, mean per launch
IDN_FUSED_MARCH=2 IDN_PROFILE_KERNEL=render_fused_kernel bash tools/profile_round.sh f32 > $F/profile_render_split.log 2>&1
cp gpurun_out/prof_f32/pmc_summary.json $F/pmc_render_split_final.json
cp gpurun_out/prof_f32/kernel_stats.csv $F/kernel_stats_render_split_final.csv
echo "profile render_split done"
for P in f32 bf16x6; do
  bash tools/profile_round.sh $P > $F/profile_$P.log 2>&1
  cp gpurun_out/prof_$P/pmc_summary.json $F/pmc_mlp_${P}_final.json
  cp gpurun_out/prof_$P/kernel_stats.csv $F/kernel_stats_${P}_final.csv
  echo "profile $P done"
done
bash tools/profile_train.sh > $F/profile_train.log 2>&1
cp gpurun_out/prof_train/kernel_stats.csv $F/kernel_stats_train_final.csv
for j in gpurun_out/prof_train/pmc_*.json; do cp $j $F/pmc_train_$(basename $j | sed 's/^pmc_//'); done
echo "profile train done"
# does the (now L2-resident) 3.375 MiB stream still cost anything?  wrap = the stream wraps after 8 slices (384 KiB; wrong results)
if [ -f ideal-nerf_amd/libidealnerf_wrap.so ]; then bash tools/ab_bench.sh wrap - bf16x6 > $F/ab_x6_stream_wrap.log 2>&1; cat $F/ab_x6_stream_wrap.log; fi
# the PMC summaries must be in profiles/ for bench.py to report `traffic`: stage them where it looks
mkdir -p profiles
for P in f32 bf16x6; do cp $F/pmc_mlp_${P}_final.json profiles/${ROUND:-r04}_pmc_mlp_${P}_final.json; done
cp $F/pmc_render_fused_final.json profiles/${ROUND:-r04}_pmc_render_fused_final.json
cp $F/pmc_render_split_final.json profiles/${ROUND:-r04}_pmc_render_split_final.json
python bench.py > $F/bench_default.json 2> $F/bench_default.err
python bench.py --precision bf16x6 --no-cpu-baseline > $F/bench_bf16x6.json 2>> $F/bench_default.err
python bench.py --workload train --steps 12 --warmup 4 > $F/bench_train.json 2>> $F/bench_default.err
python bench.py --workload torso > $F/bench_torso_bf16.json 2>> $F/bench_default.err
IDN_DIST_BACKEND=gloo IDN_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $F/bench_n2_gloo_rehearsal.json 2>> $F/bench_default.err
IDN_DIST_BACKEND=gloo IDN_FORCE_DEVICE=0 python bench.py --workload train --gpus 2 --steps 6 --warmup 2 > $F/bench_train_n2_gloo_rehearsal.json 2>> $F/bench_default.err
echo "bench lines done"
