mkdir -p gpurun_out/r2o
(timeout 120 python tools/bench_gemm_ab.py 257 60 6912 phi3.gate > gpurun_out/r2o/clk_gemm.txt 2>&1) &
P=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" ; echo ---; sleep 1; done > gpurun_out/r2o/clk_smi.txt
wait $P
cat gpurun_out/r2o/clk_smi.txt | head -40
echo IDLE; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power"
tail -2 gpurun_out/r2o/clk_gemm.txt
