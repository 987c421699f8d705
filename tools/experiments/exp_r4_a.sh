#!/bin/bash
# round 4, experiment batch A: ViT GEMM tile sweep, stream priorities for the llava tower, baseline A/B/A/B inside one call
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 --cpu-baseline off > /dev/null 2>&1     # warm-up process (first process on a fresh box runs cold)
python tools/vit_gemm_sweep.py > gpurun_out/r4_vit_sweep.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --cpu-baseline off"
for rep in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('base', d['ms_per_step'])" >> gpurun_out/r4_prio.txt
  D3D_BENCH_HP_STREAM=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hp_main', d['ms_per_step'])" >> gpurun_out/r4_prio.txt
  for blk in 0 8 16; do
    D3D_BENCH_HP_STREAM=1 D3D_LLAVA_AFTER_CLIP_BLOCK=$blk $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hp_main+llava_after_block_$blk', d['ms_per_step'])" >> gpurun_out/r4_prio.txt
  done
  D3D_LLAVA_AFTER_CLIP_BLOCK=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('llava_after_block_0 (no priority)', d['ms_per_step'])" >> gpurun_out/r4_prio.txt
done
cat gpurun_out/r4_prio.txt
