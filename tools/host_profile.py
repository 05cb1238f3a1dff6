import cProfile, pstats, sys, io, time
sys.path.insert(0, '/root/repo')
sys.argv = ['bench.py', '--workload', 'c3', '--steps', '30', '--warmup', '2', '--no-cpu-baseline', '--no-other-workloads']
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
print(s.getvalue()[:6000])
