import cProfile, os, pstats, sys, io
from_dir = os.path.dirname(os.path.abspath(__file__))
frames = sys.argv[1] if len(sys.argv) > 1 else "4000"
sys.argv = ["time_calculators.py", frames]
ns = {}
src = open(os.path.join(from_dir, "time_calculators.py")).read().replace("os.path.abspath(__file__)", repr(os.path.join(from_dir, "time_calculators.py")))
head = src[:src.index('timed("RadialDistributionFunction(500')]
exec(compile(head, "head", "exec"), ns)
exp = ns["exp"]
exp.run.GreenKuboDiffusionCoefficients(data_range=400, plot=False)   # warm-up
for label, fn in (("GK self", lambda: exp.run.GreenKuboDiffusionCoefficients(data_range=500, plot=False)),
                  ("Einstein", lambda: exp.run.EinsteinDiffusionCoefficients(data_range=100, plot=False)),
                  ("ADF", lambda: exp.run.AngularDistributionFunction(number_of_configurations=50, cutoff=6.0, plot=False))):
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
    print("=====", label); print("\n".join(s.getvalue().splitlines()[0:26]))
