set -x
ZT_BENCH_ABL="64,64" python tools/bench_conv.py 2>&1 | grep "c64" | tee gpurun_out/r03k_conv_rs_ablation.txt
ZT_BENCH_ABL="48,48" python tools/bench_conv.py 2>&1 | grep "c48" | tee -a gpurun_out/r03k_conv_rs_ablation.txt
ZT_BENCH_ONLY="64,64,3" bash tools/pmc_kernel.sh "conv_rs_bf16_kernel" tools/bench_conv.py 2>&1 | grep -v simple_timer > gpurun_out/r03k_conv_rs_pmc.txt; cut -c1-400 gpurun_out/r03k_conv_rs_pmc.txt
