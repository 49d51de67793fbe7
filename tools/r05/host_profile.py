"""cProfile of the host side of one bench configuration (the SGDet step is bound by the host's enqueue time: DESIGN.md section 5).
usage: python tools/r05/host_profile.py cfg3 [steps]   -> top functions by own time and by cumulative time"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.argv = ['bench.py', '--config', sys.argv[1], '--steps', sys.argv[2] if len(sys.argv) > 2 else '10', '--warmup', '3', '--no-cpu-baseline', '--meter-every', '1000']
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    for key in ('tottime', 'cumtime'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])
