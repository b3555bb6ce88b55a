#!/bin/bash
# One gpurun call: kernel-trace stats + the PMC passes of the train workload (BASELINE configs[2]),
# each counter set in its own pass, --pmc never combined with trace domains.  Results under
# gpurun_out/prof_train/: kernel_stats.csv and one pmc_<kernel>.json per MFMA kernel family.
#   IDN_COMMIT=<hash> bash tools/profile_train.sh
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_train
rm -rf $O; mkdir -p $O
B="python3 bench.py --workload train --steps 4 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
echo "stats done"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- $B > $O/pmc_$tag.log 2>&1
  echo "pmc $tag done"
done
for K in "gemm_tn_x6_kernel" "gemm_tn_kernel<4, 1," "delta_chain_x6_kernel" "mlp_bf16x6_kernel<0, true>"; do
  name=$(echo "$K" | tr -c 'a-zA-Z0-9' '_' | sed 's/__*/_/g; s/_$//')
  python3 tools/pmc_summary.py "$K" $O/pmc_$name.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES $O/pmc_GRBM_GUI_ACTIVE > $O/pmc_$name.log
done
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_WAVE_CYCLES $O/pmc_GRBM_GUI_ACTIVE $O/stats
