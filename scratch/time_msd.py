"""MSD band kernels: matrix-core against vector version, time and agreement."""
import os, subprocess, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(S, T, R):
    from lammps_analysis_amd import engine
    engine.set_device(0)
    x = engine.DeviceArray((S, T, 3), np.float32)
    x.fill_langevin(3, 0.05, 1.0) if hasattr(x, "fill_langevin") else x.fill_uniform(3, 0., 1.)
    tau = np.arange(R)
    engine.profile_enable(True)
    out = None
    for it in range(3):
        engine.profile_reset()
        out = engine.msd_windows(x, 1, 0, T - R, tau)
        ms, n = engine.profile_query("msd")
    return out, ms

if __name__ == "__main__":
    if len(sys.argv) > 1:
        S, T, R = map(int, sys.argv[1:4])
        out, ms = run(S, T, R)
        print(json.dumps({"ms": ms, "out": out.tolist()}))
    else:
        for S, T, R in [(5000, 16384, 100), (5000, 16384, 500), (5000, 100000, 100)]:
            res = {}
            for mode in ("mfma", "valu"):
                env = dict(os.environ)
                if mode == "valu":
                    env["MDX_MSD_BAND_VALU"] = "1"
                r = subprocess.run([sys.executable, __file__, str(S), str(T), str(R)], env=env,
                                   capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    print(mode, "failed", r.stderr[-2000:]); sys.exit(1)
                res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
            a, b = np.array(res["mfma"]["out"]), np.array(res["valu"]["out"])
            print(f"S={S} T={T} R={R}: mfma {res['mfma']['ms']:.3f} ms, valu {res['valu']['ms']:.3f} ms, "
                  f"max |diff|/max = {np.max(np.abs(a - b)) / np.max(np.abs(b)):.3e}, "
                  f"max rel (m>0) = {np.max(np.abs(a[1:] - b[1:]) / np.abs(b[1:])):.3e}, msd[0] = {a[0]}", flush=True)
