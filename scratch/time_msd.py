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


def run_acf(S, T, R):
    from lammps_analysis_amd import engine
    engine.set_device(0)
    x = engine.DeviceArray((S, T, 3), np.float32)
    x.fill_langevin(3, 0.05, 1.0)
    tau = np.arange(R)
    engine.profile_enable(True)
    for it in range(3):
        engine.profile_reset()
        res = engine.acf_windows(x, 1, 0, T - R, tau)
        ms, n = engine.profile_query("acf_partial")
    out = res[0] if isinstance(res, (tuple, list)) else getattr(res, "acf_sum", res)
    return np.asarray(out), ms

if __name__ == "__main__":
    if len(sys.argv) > 1:
        S, T, R = map(int, sys.argv[1:4])
        out, ms = (run_acf if sys.argv[4] == "acf" else run)(S, T, R)
        print(json.dumps({"ms": ms, "out": out.tolist()}))
    else:
        for S, T, R, what in [(5000, 16384, 100, "msd"), (5000, 16384, 500, "msd"),
                              (5000, 100000, 100, "msd"), (5000, 16384, 500, "acf"),
                              (5000, 100000, 500, "acf")]:
            res = {}
            for mode in ("mfma", "mfma16", "valu"):
                env = dict(os.environ)
                if mode == "valu":
                    env["MDX_MSD_BAND_VALU"] = "1"
                    env["MDX_ACF_BAND_VALU"] = "1"
                if mode == "mfma16":
                    env["MDX_BAND_MFMA16"] = "1"
                r = subprocess.run([sys.executable, __file__, str(S), str(T), str(R), what], env=env,
                                   capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    print(mode, "failed", r.stderr[-2000:]); sys.exit(1)
                res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
            a, b = np.array(res["mfma"]["out"]), np.array(res["valu"]["out"])
            print(f"{what} S={S} T={T} R={R}: mfma4 {res['mfma']['ms']:.3f} ms, mfma16 {res['mfma16']['ms']:.3f} ms, valu {res['valu']['ms']:.3f} ms, "
                  f"max |diff|/max = {np.max(np.abs(a - b)) / np.max(np.abs(b)):.3e}, "
                  f"max rel (m>0) = {np.max(np.abs(a[1:] - b[1:]) / np.abs(b[1:])):.3e}, msd[0] = {a[0]}", flush=True)
