"""cProfile of the host side of bench.py's step loop (development aid): python tools/host_profile.py [bench args]"""
import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-timing"] + sys.argv[1:]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
