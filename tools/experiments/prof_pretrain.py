import cProfile, pstats, sys, os, io
sys.argv = ["bench_pretrain.py", "--steps", "4", "--warmup", "2"]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "bench_pretrain.py"), run_name="__main__")
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
    print(s.getvalue()[:14000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
    print(s.getvalue()[:9000])
