#!/bin/bash
# final tree of round 6: the once-flaky full-step test 8 x in fresh processes, smoke, the whole GPU suite, the driver's bench command
ulimit -c 0
mkdir -p gpurun_out/final gpurun_out/r06_bench
T=tests/test_gpu_full_step.py::test_full_config_step_prune_determinism_packed_vs_single
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest $T -x -q -s -p no:cacheprovider > gpurun_out/final/fullstep_$i.log 2>&1
  echo "full-step run $i rc=$? $(grep -h 'packed vs per-prompt' gpurun_out/final/fullstep_$i.log | grep -v print | sed 's/.*worst of 8) //' | tr '\n' ' ')"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Extension" | tail -3
python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/final/gpu_tests.log 2>&1; echo "gpu suite rc=$?"
grep -v "^Extension modules\|^  File" gpurun_out/final/gpu_tests.log | tail -4 | cut -c1-200
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench/bench_default.json 2> gpurun_out/r06_bench/bench_default.err; echo "bench rc=$?"
