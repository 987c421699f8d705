#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest_g.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_gputest_g.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke_g.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r4_smoke_g.txt
timeout 300 python tools/bench_render.py > gpurun_out/r4_render_ab.txt 2>&1
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_g.json 2> gpurun_out/r4_bench_g.err ) 2> gpurun_out/r4_bench_g.time
tail -3 gpurun_out/r4_gputest_g.txt; tail -2 gpurun_out/r4_smoke_g.txt; cat gpurun_out/r4_render_ab.txt | head; cat gpurun_out/r4_bench_g.time; head -c 500 gpurun_out/r4_bench_g.json
