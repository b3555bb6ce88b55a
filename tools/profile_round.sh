#!/bin/bash
# One gpurun call: kernel-trace stats + the PMC passes of the default bench (each counter set in
# its own pass, --pmc never combined with trace domains).  Results under gpurun_out/prof_<precision>/.
#   IDN_COMMIT=<hash> bash tools/profile_round.sh [f32|bf16x3|fp16x3|bf16]      (the GPU box has no .git)
set -e
export TMPDIR=/tmp
P=${1:-f32}
O=gpurun_out/prof_$P
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mode --precision $P"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
echo "stats done"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- $B > $O/pmc_$tag.log 2>&1
  echo "pmc $tag done"
done
K=mlp_bf16x3_kernel; [ "$P" = "f32" ] && K=${IDN_PROFILE_KERNEL:-mlp_f32_kernel}; [ "$P" = "bf16x6" ] && K=mlp_bf16x6_kernel; [ "$P" = "bf16" ] && K=mlp_bf16_kernel; [ "$P" = "fp16x3" ] && K=mlp_fp16x3_kernel
python3 tools/pmc_summary.py $K $O/pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES $O/pmc_GRBM_GUI_ACTIVE > $O/pmc_summary.log
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# raw counter dumps are large: keep only the summaries
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES $O/pmc_GRBM_GUI_ACTIVE $O/stats
