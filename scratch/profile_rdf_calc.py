import cProfile, pstats, runpy, sys, io
sys.argv = ["time_calculators.py", "4000"]
import builtins
pr = cProfile.Profile()
ns = {}
src = open("scratch/time_calculators.py").read()
head = src[:src.index('timed("RadialDistributionFunction(500')]
exec(compile(head, "head", "exec"), ns)
exp = ns["exp"]
exp.run.RadialDistributionFunction(number_of_configurations=500, plot=False)   # warm-up
pr.enable()
exp.run.RadialDistributionFunction(number_of_configurations=-1, plot=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
