#!/bin/bash
# End-of-round measurements on ONE GPU box (one gpurun call; ~15 min): the GPU test log, the PMC / kernel-trace
# summaries of the three inference kernels and of the train step, and the bench lines -- all into gpurun_out/final/,
# from where they are copied to profiles/r<NN>_* by hand.
#   IDN_COMMIT=$(git rev-parse --short HEAD) bash tools/final_round.sh        (the GPU box has no .git)
set -e
export TMPDIR=/tmp
F=gpurun_out/final
rm -rf $F; mkdir -p $F
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $F/pytest_gpu_final.log 2>&1
tail -1 $F/pytest_gpu_final.log
for P in f32 bf16x6 bf16x3 fp16x3; do
  bash tools/profile_round.sh $P > $F/profile_$P.log 2>&1
  cp gpurun_out/prof_$P/pmc_summary.json $F/pmc_mlp_${P}_final.json
  cp gpurun_out/prof_$P/kernel_stats.csv $F/kernel_stats_${P}_final.csv
  echo "profile $P done"
done
bash tools/profile_train.sh > $F/profile_train.log 2>&1
cp gpurun_out/prof_train/kernel_stats.csv $F/kernel_stats_train_final.csv
for j in gpurun_out/prof_train/pmc_*.json; do cp $j $F/pmc_train_$(basename $j | sed 's/^pmc_//'); done
echo "profile train done"
# the PMC summaries must be in profiles/ for bench.py to report `traffic`: stage them where it looks
mkdir -p profiles
for P in f32 bf16x6 bf16x3 fp16x3; do cp $F/pmc_mlp_${P}_final.json profiles/${ROUND:-r02}_pmc_mlp_${P}_final.json; done
python bench.py > $F/bench_default.json 2> $F/bench_default.err
python bench.py --precision mixed --no-cpu-baseline > $F/bench_mixed.json 2>> $F/bench_default.err
python bench.py --precision bf16 --no-cpu-baseline > $F/bench_bf16_plain.json 2>> $F/bench_default.err
python bench.py --precision bf16x6 --no-cpu-baseline > $F/bench_bf16x6.json 2>> $F/bench_default.err
python bench.py --workload train > $F/bench_train.json 2>> $F/bench_default.err
python bench.py --workload torso > $F/bench_torso_bf16.json 2>> $F/bench_default.err
python bench.py --workload torso --precision mixed > $F/bench_torso_mixed.json 2>> $F/bench_default.err
IDN_DIST_BACKEND=gloo IDN_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $F/bench_n2_gloo_rehearsal.json 2>> $F/bench_default.err
echo "bench lines done"
