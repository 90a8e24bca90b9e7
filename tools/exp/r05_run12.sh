#!/bin/bash
# round 5, run 12: rocprofv3 evidence of the inference half after the retrieval-kernel changes (tag r05f)
bash tools/exp/r05_prof_infer.sh r05f > /dev/null 2>&1
bash tools/exp/r05_pmc_infer.sh r05f > /dev/null 2>&1
head -8 gpurun_out/prof_infer_r05f/retrieval_kernel_stats.csv | cut -c1-180
cat gpurun_out/prof_infer_r05f/retrieval_time.txt
head -8 gpurun_out/pmc_infer_r05f/rt_pmc_hbm_traffic.txt; head -6 gpurun_out/pmc_infer_r05f/rt_pmc_mfma_util.txt
