#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line.

Workload (BASELINE.json configs[1]): RadialDistributionFunction on a synthetic
10 000-atom NaCl melt (5 000 Na + 5 000 Cl, cubic box L = 65, cutoff L/2 - 0.1 =
32.4, 3 240 bins, reference-strict species masks).  The trajectory is generated
on the device and stays resident in HBM in the reference layout
``[atoms][frames][3]`` float32 per species; one *step* histograms one batch of
``--frames-per-step`` frames (pack kernel + pair-tile kernel).  ``value`` is frames
per second over all ranks.  With N > 1 every rank owns its own frames (weak
scaling, no collective inside a step); the per-rank histograms are summed once at
the end with one all-reduce, outside the timed steps (reported separately).

Also measured in the same run, outside the timed steps and reported as extra
keys: the Green-Kubo chain of configs[2] (atom reduction of a 10 000-atom velocity
block to the ionic current, then the windowed autocorrelation, data_range 500) and
the CPU baseline (C restatement of the reference algorithm, OpenMP) on a bounded
sample of the same trajectory, whose histogram is also compared bit for bit with
the GPU result for those frames.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

N_NA = 5000
N_CL = 5000
BOX = 65.0
CUTOFF = np.float32(BOX / 2 - 0.1)
NBINS = int(CUTOFF / 0.01)
TRAJECTORY_FRAMES = 100000       # BASELINE.json configs[1]: 10k atoms x 100k frames
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak
F32_VALU_PEAK_TFLOPS = 157.3     # MI355X vector fp32 spec peak (FMA counted as 2)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames per step; 0 = the configuration's 100 000 frames divided "
                         "over the timed steps")
    ap.add_argument("--cpu-frames", type=int, default=24,
                    help="frames of the CPU-baseline sample (0 disables it)")
    ap.add_argument("--gk-frames", type=int, default=100000,
                    help="frames of the Green-Kubo side measurement (0 disables it)")
    ap.add_argument("--skip-host-input", action="store_true",
                    help="leave out the host-memory side measurement (counter passes: "
                         "its 256-frame blocks share a grid with the full-size launch)")
    ap.add_argument("--literal-kernel", action="store_true",
                    help="time the literal-arithmetic kernel instead of the table kernel")
    return ap.parse_args()


def recorded_traffic(kernel_key: str):
    """HBM-side bytes per launch of the kernel whose summary key starts with
    ``kernel_key``, from the committed PMC summary of this command with default sizes
    (the counters cannot be read from inside this process); None when not recorded."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        entries = [v for k, v in table.items() if k.startswith(kernel_key)]
        if not entries:
            return None
        # several grids of one kernel: the full-size launches are the longest ones
        entry = max(entries, key=lambda v: v["mean_duration_us_under_pmc"])
        return entry["fetch_bytes_x2"] + entry["write_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def recorded_counter(kernel_key: str, counter: str):
    """Mean of a hardware counter per launch of the longest-running grid of a kernel, from
    the same committed summary as ``recorded_traffic``; None when not recorded."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        entries = [v for k, v in table.items() if k.startswith(kernel_key) and counter in v]
        if not entries:
            return None
        return max(entries, key=lambda v: v["mean_duration_us_under_pmc"])[counter]
    except (OSError, KeyError, ValueError):
        return None


def main() -> None:
    args = parse_args()
    env_world = os.environ.get("MDX_WORLD_SIZE") or os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        # started as a plain command: this process becomes the launcher of N ranks (one
        # per GPU) and never touches a GPU itself; rank 0 prints the result line
        from lammps_analysis_amd import parallel as launcher
        raise SystemExit(launcher.launch([os.path.abspath(__file__), *sys.argv[1:]], args.gpus))
    # Native libraries write banners to file descriptor 1 (RCCL prints its version block
    # when the first communicator is created).  Everything except the result line goes to
    # stderr: descriptor 1 points there until the JSON line is written to the saved copy.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    from lammps_analysis_amd import _lib, engine, parallel

    _lib.load()
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: no HIP device is visible to libmdx")
    # BENCH_FORCE_COMM=1 creates the RCCL communicator even for one rank (rehearsal of
    # the collective on a single GPU)
    force = os.environ.get("BENCH_FORCE_COMM") == "1"
    rank, world = parallel.init(transport="rccl" if (force and env_world is None) else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    local_rank = int(os.environ.get("MDX_LOCAL_RANK") or os.environ.get("LOCAL_RANK") or rank)
    engine.set_device(local_rank % _lib.device_count())

    def barrier_sync():
        engine.synchronize()
        parallel.barrier()

    steps, warmup = args.steps, args.warmup
    fps = args.frames_per_step if args.frames_per_step > 0 else max(1, TRAJECTORY_FRAMES // steps)
    # the timed steps walk once through the resident trajectory (100 000 frames per rank
    # with the default step size); warm-up steps re-read its first blocks
    n_frames = fps * steps

    # ---- resident synthetic trajectory, reference layout [atoms][frames][3]
    na = engine.DeviceArray((N_NA, n_frames, 3), np.float32)
    cl = engine.DeviceArray((N_CL, n_frames, 3), np.float32)
    na.fill_uniform(seed=1000 + 2 * rank, lo=0.0, hi=BOX)
    cl.fill_uniform(seed=1001 + 2 * rank, lo=0.0, hi=BOX)

    rdf = engine.RdfEngine([N_NA, N_CL], [BOX, BOX, BOX], CUTOFF, NBINS,
                           strict_reference_masks=True,
                           literal_kernel=args.literal_kernel)
    species = [na, cl]

    def step(i: int) -> None:
        rdf.accumulate(species, frames=np.arange(i * fps, (i + 1) * fps, dtype=np.int64))

    for i in range(warmup):
        step(i % steps)
    rdf.reset()
    engine.profile_enable(True)
    engine.profile_reset()
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    engine.profile_enable(False)
    tile_ms, tile_launches = engine.profile_query("rdf_tile")
    pack_ms, pack_launches = engine.profile_query("rdf_pack")

    elapsed = float(parallel.all_reduce_max([elapsed])[0])  # the slowest rank's time

    # the one exchange step of the run: per-rank histograms summed over the ranks
    allreduce_ms = None
    if world > 1 or parallel.communicator() is not None:
        barrier_sync()
        ta = time.perf_counter()
        if parallel.communicator() is not None:
            rdf.reduce(parallel.communicator())  # in place on the devices, RCCL
            hist_total = rdf.read()
        else:
            hist_total = parallel.all_reduce_sum(rdf.read())
        allreduce_ms = (time.perf_counter() - ta) * 1e3
    else:
        hist_total = rdf.read()

    total_frames = steps * fps * world
    value = total_frames / elapsed
    ms_per_step = elapsed / steps * 1e3

    n_atoms = N_NA + N_CL
    pairs_per_frame = n_atoms * (n_atoms - 1) // 2
    bytes_per_frame = n_atoms * 3 * 4
    launches = max(tile_launches, 1)
    tile_avg_ms = tile_ms / launches
    frames_per_launch = steps * fps / launches
    achieved_gbps = (bytes_per_frame * frames_per_launch) / (tile_avg_ms * 1e-3) / 1e9
    roofline = {
        "kernel": "rdf_tile_kernel",
        "bound": "hbm",
        "achieved": achieved_gbps,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved_gbps / HBM_PEAK_GBPS,
        "traffic": recorded_traffic("rdf_tile_kernel<false>")
        if (fps == 1024 and not args.literal_kernel) else None,
        "traffic_source": "profiles/r01_pmc.json: separate rocprofv3 --pmc FETCH_SIZE and "
                          "WRITE_SIZE passes over this command; bytes per launch = "
                          "2*FETCH_SIZE (gfx950 counts 128-B reads as 64 B) + WRITE_SIZE",
        "avg_launch_ms": tile_avg_ms,
        "launches": tile_launches,
        "algorithmic_bytes_per_launch": bytes_per_frame * frames_per_launch,
        "note": ("BASELINE.json asks for the HBM fraction; the pair loop reuses every "
                 "12-byte atom ~N/2 times from LDS/registers, so the physical limiter "
                 "is f32 VALU issue + LDS atomics, see valu"),
    }
    pairs_per_s_kernel = pairs_per_frame * frames_per_launch / (tile_avg_ms * 1e-3)
    # VALU issue bound of the table kernel's wrapped-frame loop: 17.5 issue slots per
    # pair (ISA count: 1.5 packed subtract, 6 minimum image, 2.5 packed square/add,
    # 2 for the 8-cycle v_sqrt_f32, 0.5 packed fma, 5 cutoff/bin/edge/address), one
    # slot = 4 cycles of a SIMD16 for a 64-lane wave, 1024 SIMDs at 2.4 GHz
    slots_per_pair = 17.5
    issue_bound = 1024 * 2.4e9 / 4.0 * 64 / slots_per_pair
    valu = {
        "pairs_per_s_kernel": pairs_per_s_kernel,
        "issue_slots_per_pair": slots_per_pair,
        "issue_bound_pairs_per_s": issue_bound,
        "frac_of_issue_bound": pairs_per_s_kernel / issue_bound,
        "pairs_per_s_end_to_end": pairs_per_frame * value / world,
        "pack_kernel_ms_total": pack_ms,
        "tile_kernel_ms_total": tile_ms,
    }
    # measured side of the same statement: wave-level VALU instructions per launch from
    # the SQ counter pass, 4 cycles each on one of 1024 SIMDs (v_sqrt_f32 takes 8, so
    # the true busy fraction is higher), against the launch time measured here
    valu_insts = recorded_counter("rdf_tile_kernel<false>", "SQ_INSTS_VALU") \
        if (fps == 1024 and not args.literal_kernel) else None
    if valu_insts:
        valu["valu_insts_per_launch_pmc"] = valu_insts
        valu["valu_insts_per_pair_slot"] = valu_insts * 64 / (pairs_per_frame * frames_per_launch)
        valu["simd_issue_busy_frac_at_2p4GHz"] = (valu_insts * 4.0 / 1024
                                                  / (tile_avg_ms * 1e-3 * 2.4e9))

    result = {
        "metric": "frames/sec (RDF + GK-ACF) on 10k-atom synthetic trajectory; %HBM-roofline",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "RadialDistributionFunction, 10k-atom NaCl melt (5000 Na + 5000 Cl), "
                        "L=65, cutoff=32.4, 3240 bins, strict reference masks "
                        f"(BASELINE.json configs[1]); {steps * fps} frames per rank resident "
                        "in HBM, each histogrammed once in the timed steps",
            "frames_per_step": fps,
            "frames_per_rank": steps * fps,
            "parallelism": f"frames sharded over {world} rank(s), one histogram all-reduce",
            "kernel": "literal" if args.literal_kernel else "table",
        },
        "roofline": roofline,
        "valu": valu,
        "pairs_counted": int(hist_total.sum()),
        "allreduce_ms": allreduce_ms,
    }

    if rank == 0:
        if args.cpu_frames > 0:
            result["cpu_baseline"] = cpu_baseline(engine, species, n_frames, args.cpu_frames,
                                                  rank)
        if args.gk_frames > 0:
            result["gk_acf"] = gk_side_measurement(engine, args.gk_frames)
        if not args.skip_host_input:
            result["host_input"] = host_input_measurement(engine, fps)
    parallel.barrier()
    transport = parallel.transport()
    parallel.shutdown()
    if rank == 0:
        result["config"]["transport"] = transport
        sys.stdout.flush()
        with os.fdopen(result_fd, "w") as result_stream:
            result_stream.write(json.dumps(result) + "\n")


def host_input_measurement(engine, fps: int) -> dict:
    """One step's worth of frames handed over as pageable host arrays (what the
    calculators do with database batches): staging, PCIe copies and kernels."""
    host = []
    for s_idx, n in enumerate((N_NA, N_CL)):
        dev = engine.DeviceArray((n, fps, 3), np.float32)
        dev.fill_uniform(seed=500 + s_idx, lo=0.0, hi=BOX)
        host.append(dev.to_numpy())
        dev.free()
    rdf = engine.RdfEngine([N_NA, N_CL], [BOX] * 3, CUTOFF, NBINS)
    rdf.accumulate(host)  # warm-up: page-locked staging is allocated once per engine
    engine.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        rdf.accumulate(host)
    engine.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rdf.close()
    return {"frames_per_s": fps / dt, "ms_per_step": dt * 1e3,
            "note": "same step from pageable host memory, PCIe copies overlapped with "
                    "the kernels; never the headline value"}


def cpu_baseline(engine, species, n_frames: int, sample: int, rank: int) -> dict:
    """C restatement of the reference algorithm (oracle, OpenMP) on the first
    ``sample`` frames of rank 0's trajectory; also a bit-exact check of the GPU."""
    from oracle import c_oracle

    sample = min(sample, n_frames)
    host = []
    for s_idx, n in enumerate((N_NA, N_CL)):
        seed = 1000 + s_idx + 2 * rank
        # element (a, f, c) of the device array sits at flat index (a*n_frames + f)*3 + c
        idx = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(n_frames * 3)
               + np.arange(sample * 3, dtype=np.uint64)[None, :])
        host.append(engine.uniform_fill_at(idx, seed, 0.0, BOX).reshape(n, sample, 3))
    pos = np.concatenate(host, axis=0)
    threads = c_oracle.num_threads()
    t0 = time.perf_counter()
    ref = c_oracle.rdf_hist(pos, [N_NA, N_CL], [BOX] * 3, CUTOFF, NBINS, strict=True)
    dt = time.perf_counter() - t0
    check = engine.RdfEngine([N_NA, N_CL], [BOX] * 3, CUTOFF, NBINS)
    check.accumulate(species, frames=np.arange(sample, dtype=np.int64))
    got = check.read()
    check.close()
    return {
        "value": sample / dt,
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {sample} frames of the same 10k-atom trajectory, "
                  f"{dt:.2f} s wall, oracle/mdx_oracle.c with OpenMP",
        "gpu_matches_bit_exact": bool(np.array_equal(got, ref)),
    }


def gk_side_measurement(engine, n_frames: int) -> dict:
    """BASELINE.json configs[2] in small: ionic current of a 10k-atom velocity block
    (HBM-bound atom reduction) followed by the windowed autocorrelation."""
    data_range = 500
    vel_na = engine.DeviceArray((N_NA, n_frames, 3), np.float32)
    vel_cl = engine.DeviceArray((N_CL, n_frames, 3), np.float32)
    vel_na.fill_uniform(seed=31, lo=-1.0, hi=1.0)
    vel_cl.fill_uniform(seed=32, lo=-1.0, hi=1.0)
    tau = np.arange(data_range)
    time_axis = tau * 0.002
    n_win = n_frames - data_range

    def chain_via_host():
        cur = engine.weighted_atom_sum(vel_na, scalar_weight=1.0)
        cur = engine.weighted_atom_sum(vel_cl, scalar_weight=-1.0, out=cur, accumulate=True)
        series = cur.astype(np.float32)[None]  # what a calculator loads from the store
        acf, mom = engine.acf_windows(series, 1, 0, n_win, tau, time=time_axis)
        return cur, acf

    cur_dev = engine.DeviceArray((n_frames, 3), np.float64)
    series_dev = engine.DeviceArray((1, n_frames, 3), np.float32)

    def chain():
        # the series never leaves the device: float64 sums, rounded to the store's
        # float32 by the scale map, correlated in place
        engine.weighted_atom_sum(vel_na, scalar_weight=1.0, out=cur_dev)
        engine.weighted_atom_sum(vel_cl, scalar_weight=-1.0, out=cur_dev, accumulate=True)
        engine.map_coordinates(engine.MAP_SCALE, cur_dev.reshape((1, n_frames, 3)),
                               [1.0, 1.0, 1.0], out=series_dev)
        acf, mom = engine.acf_windows(series_dev, 1, 0, n_win, tau, time=time_axis)
        return cur_dev, acf

    _, acf_host = chain_via_host()  # warm-up of both variants; they must agree
    _, acf_dev = chain()
    if not np.array_equal(acf_host, acf_dev):
        raise SystemExit("GK chain: device-resident and host-staged series disagree")
    engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        chain_via_host()
    engine.synchronize()
    dt_host = (time.perf_counter() - t0) / 3
    engine.profile_enable(True)
    engine.profile_reset()
    engine.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        cur, acf = chain()
    engine.synchronize()
    dt = (time.perf_counter() - t0) / reps
    engine.profile_enable(False)
    sum_ms, sum_n = engine.profile_query("atom_sum")
    slide_ms, slide_n = engine.profile_query("acf_finish")
    bytes_per_launch = N_NA * n_frames * 3 * 4
    avg_ms = sum_ms / max(sum_n, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    out = {
        "workload": f"GreenKuboIonicConductivity chain, 10k atoms x {n_frames} frames, "
                    f"data_range {data_range}, correlation_time 1",
        "frames_per_s": n_frames / dt,
        "ms_per_pass": dt * 1e3,
        "ms_per_pass_series_via_host": dt_host * 1e3,
        "roofline_atom_sum": {
            "kernel": "weighted_sum_kernel<float,4>", "bound": "hbm", "achieved": achieved,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": recorded_traffic("weighted_sum_kernel<float, 4>")
            if n_frames == 100000 else None,
            "avg_launch_ms": avg_ms, "launches": sum_n,
            "algorithmic_bytes_per_launch": bytes_per_launch},
        "acf_slide_kernel_ms": slide_ms / max(slide_n, 1),
    }

    # per-atom velocity autocorrelation of one species (GreenKuboDiffusionCoefficients):
    # lag products summed over atoms into a band table, then sliding windows
    def vacf():
        return engine.acf_windows(vel_na, 1, 0, n_win, tau, time=time_axis)

    vacf()  # warm-up
    engine.profile_enable(True)
    engine.profile_reset()
    engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        vacf()
    engine.synchronize()
    dt_v = (time.perf_counter() - t0) / reps
    engine.profile_enable(False)
    band_ms, band_n = engine.profile_query("acf_partial")
    slide_ms, slide_n = engine.profile_query("acf_finish")
    band_flops = 2.0 * n_frames * data_range * N_NA * 3
    band_avg = band_ms / max(band_n, 1)
    out["vacf"] = {
        "workload": f"per-atom VACF, {N_NA} atoms x {n_frames} frames, data_range "
                    f"{data_range}, correlation_time 1",
        "frames_per_s": n_frames / dt_v,
        "ms_per_pass": dt_v * 1e3,
        "band_kernel_ms": band_avg,
        "slide_kernel_ms": slide_ms / max(slide_n, 1),
        "band_kernel_tflops_f64": band_flops / (band_avg * 1e-3) / 1e12 if band_avg else None,
        "f64_vector_peak_tflops": 78.6,
    }
    # per-atom mean-square displacement of one species (EinsteinDiffusionCoefficients,
    # default data_range 100), on the same array read as positions
    msd_range = 100
    tau_msd = np.arange(msd_range)

    def msd():
        return engine.msd_windows(vel_na, 1, 0, n_frames - msd_range, tau_msd)

    msd()  # warm-up
    engine.profile_enable(True)
    engine.profile_reset()
    engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        msd()
    engine.synchronize()
    dt_m = (time.perf_counter() - t0) / reps
    engine.profile_enable(False)
    msd_ms, msd_n = engine.profile_query("msd")
    msd_avg = msd_ms / max(msd_n, 1)
    out["einstein_msd"] = {
        "workload": f"per-atom MSD, {N_NA} atoms x {n_frames} frames, data_range {msd_range}, "
                    "correlation_time 1",
        "frames_per_s": n_frames / dt_m,
        "ms_per_pass": dt_m * 1e3,
        "band_kernel_ms": msd_avg,
        # subtract + fma per origin, lag, atom and component
        "band_kernel_tflops_f64": (3.0 * n_frames * msd_range * N_NA * 3
                                   / (msd_avg * 1e-3) / 1e12) if msd_avg else None,
    }
    vel_na.free()
    vel_cl.free()
    cur_dev.free()
    series_dev.free()
    return out


if __name__ == "__main__":
    main()
