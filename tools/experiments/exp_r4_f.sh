#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the bench command
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest_f.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_gputest_f.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke_f.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r4_smoke_f.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_f.json 2> gpurun_out/r4_bench_f.err ) 2> gpurun_out/r4_bench_f.time
tail -3 gpurun_out/r4_gputest_f.txt; tail -2 gpurun_out/r4_smoke_f.txt; cat gpurun_out/r4_bench_f.time; head -c 600 gpurun_out/r4_bench_f.json
