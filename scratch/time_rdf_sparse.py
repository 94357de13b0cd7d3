import sys, time
sys.argv = ["time_calculators.py", "4000"]
ns = {}
src = open("scratch/time_calculators.py").read()
exec(compile(src[:src.index('timed("RadialDistributionFunction(500')], "head", "exec"), ns)
exp = ns["exp"]
exp.run.RadialDistributionFunction(number_of_configurations=100, plot=False)
for n in (500, 499, 900, 1100, 2000):
    t0 = time.perf_counter(); exp.run.RadialDistributionFunction(number_of_configurations=n, plot=False)
    print(f"RDF {n} configurations of 4000: {(time.perf_counter()-t0)*1e3:.0f} ms", flush=True)
