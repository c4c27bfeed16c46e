import cProfile, pstats, sys, io, os
sys.argv = ["bench.py", "--steps", "300", "--warmup", "20", "--no-cpu-baseline"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "bench.py"), run_name="__main__")
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
    open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/hostprof.txt"), "w").write(s.getvalue())
