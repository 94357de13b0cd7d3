"""One warm-up and three timed launches of the default RDF kernel (counter passes)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_analysis_amd import engine
nf = 2048
engine.set_device(0)
na = engine.DeviceArray((5000, nf, 3), np.float32); na.fill_uniform(1, 0.0, 65.0)
cl = engine.DeviceArray((5000, nf, 3), np.float32); cl.fill_uniform(2, 0.0, 65.0)
rdf = engine.RdfEngine([5000, 5000], [65.0] * 3, np.float32(32.4), 3240,
                       table_kernel=bool(os.environ.get("TABLE")))
for _ in range(4):
    rdf.accumulate([na, cl])
engine.synchronize()
print(rdf.read().sum())
