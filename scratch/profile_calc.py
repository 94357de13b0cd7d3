import cProfile, pstats, sys, io
sys.argv = ["time_calculators.py", "4000"]
ns = {}
src = open("scratch/time_calculators.py").read()
head = src[:src.index('timed("RadialDistributionFunction(500')]
exec(compile(head, "head", "exec"), ns)
exp = ns["exp"]
exp.run.GreenKuboDiffusionCoefficients(data_range=400, plot=False)   # warm-up
for label, fn in (("GK self", lambda: exp.run.GreenKuboDiffusionCoefficients(data_range=500, plot=False)),
                  ("Einstein", lambda: exp.run.EinsteinDiffusionCoefficients(data_range=100, plot=False)),
                  ("ADF", lambda: exp.run.AngularDistributionFunction(number_of_configurations=50, cutoff=6.0, plot=False))):
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8)
    print("=====", label); print("\n".join(s.getvalue().splitlines()[0:20]))
