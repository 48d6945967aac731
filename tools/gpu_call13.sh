set -x
bash tools/pmc_kernel.sh "wgrad_mfma_bf16_kernel<3, 3, 3, 3, 4|wgrad_mfma_bf16_kernel<3, 3, 1, 3, 4" tools/bench_wgrad.py 2>&1 | grep -v simple_timer > gpurun_out/r03m_wgrad48_pmc.txt; cut -c1-420 gpurun_out/r03m_wgrad48_pmc.txt
