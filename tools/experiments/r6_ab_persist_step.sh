cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_persist_step; rm -rf $out; mkdir -p $out
for i in 1 2 3; do
  D3D_GEMM_PERSIST=0 python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-decode --parity-golden off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('turnover  ', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> $out/ab.txt
  D3D_GEMM_PERSIST=1 python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-decode --parity-golden off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persistent', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> $out/ab.txt
done
cat $out/ab.txt
