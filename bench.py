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

Also measured in the same run (one rank only), outside the timed steps and reported as
extra keys: the Green-Kubo chain of configs[2] (atom reduction of a 10 000-atom velocity
block to the ionic current, then the windowed autocorrelation, data_range 500) with its
own CPU baseline; the flux reductions that feed configs[4]; the CPU baselines of the RDF
(C restatement of the reference algorithm with OpenMP, and the literal NumPy restatement)
on bounded samples of the same trajectory, whose histograms are also compared bit for bit
with the GPU's; and the same two calculators end to end through the product API on a
trajectory written to the on-disk store (``--api-frames``).

With N > 1 the sums of the exchange step must go through RCCL: a run whose ranks fell
back to the host channel exits non-zero unless MDX_COMM_TRANSPORT=host asked for that.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak
TRAJECTORY_FRAMES = 100000       # BASELINE.json configs[1]: 10k atoms x 100k frames

# workload -> species sizes, box length, cutoff, bins (cutoff = L/2 - 0.1, bins of 0.01)
WORKLOADS = {
    # BASELINE.json configs[1] (and [2] for the Green-Kubo side measurement)
    "nacl": {"species": (5000, 5000), "box": 65.0,
             "label": "10k-atom NaCl melt (5000 Na + 5000 Cl), L=65, cutoff=32.4, 3240 bins"},
    # BASELINE.json configs[3]: atom-pair tiles of every frame sharded over the ranks
    "water": {"species": (33334, 66666), "box": 100.0,
              "label": "100k-atom water box (33334 O + 66666 H), L=100, cutoff=49.9, 4990 bins"},
}
N_NA, N_CL = WORKLOADS["nacl"]["species"]
BOX = WORKLOADS["nacl"]["box"]
CUTOFF = np.float32(BOX / 2 - 0.1)
NBINS = int(CUTOFF / 0.01)

# Cycles one SIMD needs per wave64 instruction at the sustained rate measured by
# scratch/valu_probe_f32.hip (profiles/r02_valu_probe.txt; 1024 SIMDs, counted at 2.4 GHz),
# and the instructions of one loop iteration (4 pairs per lane) of the fast RDF kernel,
# read off its ISA: 6 packed subtractions, 24 folds, 2 packed multiplies and 6 packed fma,
# 4 square roots, 4 unsigned minima (pairs beyond the cutoff go to a dump word), 4 compares
# (near a bin edge), 4 shifts and 8 masks.  The probe prices every instruction alone, so the
# sum is a model of the loop's issue time, not a strict lower bound.
VALU_CYCLES = {"packed": 4.6, "sub_abs": 2.75, "sqrt": 8.4, "compare": 5.1, "min": 4.5,
               "shift": 2.3, "mask": 2.4}
FAST_LOOP_MIX = {"packed": 14, "sub_abs": 24, "sqrt": 4, "compare": 4, "min": 4, "shift": 4,
                 "mask": 8}
PAIRS_PER_LOOP_ITERATION = 4


def _hilbert_rank(q: np.ndarray, bits: int) -> np.ndarray:
    """Rank of the cells q[:, 0..2] of a 2^bits grid along the three-dimensional Hilbert
    curve (Skilling's transpose form), as the library's sort kernel orders them."""
    x = [q[:, 0].astype(np.int64), q[:, 1].astype(np.int64), q[:, 2].astype(np.int64)]
    top = 1 << (bits - 1)
    level = top
    while level > 1:
        low = level - 1
        for i in range(3):
            hit = (x[i] & level) != 0
            x[0] = np.where(hit, x[0] ^ low, x[0])
            t = np.where(hit, 0, (x[0] ^ x[i]) & low)
            x[0] ^= t
            x[i] ^= t
        level >>= 1
    for i in range(1, 3):
        x[i] ^= x[i - 1]
    t = np.zeros_like(x[0])
    level = top
    while level > 1:
        t = np.where((x[2] & level) != 0, t ^ (level - 1), t)
        level >>= 1
    rank = np.zeros_like(x[0])
    for j in range(bits - 1, -1, -1):
        for i in range(3):
            rank = (rank << 1) | (((x[i] ^ t) >> j) & 1)
    return rank


def fold_free_share(frame: list, box: float) -> dict:
    """Host estimate, on one frame, of how many minimum-image folds the fast kernel leaves
    out: the library sorts every species along a Hilbert curve of cells and takes a
    component of (64 i-atoms, 64 j-atoms) as it is when the two groups' boxes keep every
    difference within half a box length.  frame: float32 [n, 3] per species, the atoms that
    take part.  Returns the share of (group pair, component) combinations without a fold
    (all group pairs of different tiles; the few pairs inside one tile are left out)."""
    boxes = []
    for pos in frame:
        used = len(pos)
        bits = 1
        while bits < 4 and (1 << (3 * (bits + 1))) * 4 <= used:
            bits += 1
        per_axis = 1 << bits
        lo = pos.min(axis=0)
        scale = np.float32(per_axis) / np.float32(box)
        cell = np.minimum(np.maximum((pos - lo) * scale, 0).astype(np.int64), per_axis - 1)
        order = np.argsort(_hilbert_rank(cell, bits), kind="stable")
        pad = -len(pos) % 64
        sorted_pos = np.concatenate([pos[order], np.full((pad, 3), np.nan, np.float32)])
        groups = sorted_pos.reshape(-1, 64, 3)
        boxes.append((np.nanmin(groups, axis=1), np.nanmax(groups, axis=1)))
    half = np.float32(box) / 2
    free = total = 0
    for a in range(len(boxes)):
        for b in range(a, len(boxes)):
            (lo_a, hi_a), (lo_b, hi_b) = boxes[a], boxes[b]
            inside = (hi_b[None] - lo_a[:, None] <= half) & (hi_a[:, None] - lo_b[None] <= half)
            tile_a = np.arange(len(lo_a)) // 4
            tile_b = np.arange(len(lo_b)) // 4
            counted = tile_a[:, None] < tile_b[None] if a == b else np.ones(inside.shape[:2], bool)
            free += int(inside[counted].sum())
            total += 3 * int(counted.sum())
    return {"share": free / total, "group_pairs": total // 3}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="nacl",
                    help="nacl: frames sharded over the ranks (weak scaling, the headline); "
                         "water: every rank sees all frames and takes its share of the "
                         "atom-pair tiles (strong scaling, configs[3])")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames per step; 0 = the configuration's 100 000 frames divided "
                         "over the timed steps (nacl) or 64 (water)")
    ap.add_argument("--cpu-frames", type=int, default=240,
                    help="frames of the CPU-baseline sample (0 disables it)")
    ap.add_argument("--gk-frames", type=int, default=100000,
                    help="frames of the Green-Kubo side measurement (0 disables it)")
    ap.add_argument("--skip-host-input", action="store_true",
                    help="leave out the host-memory side measurement")
    ap.add_argument("--kernel", choices=("fast", "table", "literal"), default="fast",
                    help="RDF pair kernel to time (all three give identical counts)")
    ap.add_argument("--literal-frames", type=int, default=1,
                    help="frames of the literal NumPy baseline (0 disables it)")
    ap.add_argument("--flux-atoms", type=int, default=5000,
                    help="atoms of the flux-reduction side measurement over 100 000 frames "
                         "(0 disables it)")
    ap.add_argument("--api-frames", type=int, default=20000,
                    help="frames of the trajectory written to the on-disk store for the "
                         "end-to-end measurement through the calculators (0 disables it)")
    ap.add_argument("--cutoff", type=float, default=0.0,
                    help="RDF cutoff of the timed steps; 0 = the reference's default L/2 - "
                         "0.1 (the headline).  Any other value is a side experiment.")
    return ap.parse_args()


PMC_SUMMARY = "profiles/r03_pmc.json"


def recorded(kernel_key: str, units_per_launch: float = None, unit: str = None):
    """Per-launch counter means of a kernel from the committed PMC summary (counters
    cannot be read from inside the process), scaled to this run's launch shape.

    The summary's ``__meta__`` says how many units (frames) one recorded launch of the
    kernel processed; counters are proportional to them, so a launch of another size gets
    the recorded values times the ratio.  {} when the kernel, or the size of its recorded
    launch, is not on record."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_SUMMARY)
    try:
        with open(path) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return {}
    meta = table.get("__meta__", {})
    entries = [v for k, v in table.items() if k.startswith(kernel_key)]
    if not entries:
        return {}
    # several grids of one kernel: the full-size launches are the longest ones
    entry = dict(max(entries, key=lambda v: v.get("mean_duration_us_under_pmc", 0.0)))
    if unit is None:
        return entry
    on_record = meta.get(unit)
    if not on_record or not units_per_launch:
        return {}
    ratio = float(units_per_launch) / float(on_record)
    scaled = {k: (v * ratio if isinstance(v, (int, float)) and k != "launches_seen" else v)
              for k, v in entry.items()}
    scaled["scaled_from_units_per_launch"] = on_record
    return scaled


def traffic_of(entry: dict):
    """HBM bytes of one launch: FETCH_SIZE doubled (gfx950 tallies a 128-byte read as
    64) plus WRITE_SIZE."""
    if "fetch_bytes_x2" in entry and "write_bytes" in entry:
        return entry["fetch_bytes_x2"] + entry["write_bytes"]
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
    # parallel.init raises when ranks that each own a GPU cannot set up RCCL; ranks that
    # share a GPU land on the host channel by design, which this benchmark accepts only
    # when it was asked for by name
    host_asked = os.environ.get("MDX_COMM_TRANSPORT", "").lower() == "host"
    if world > 1 and parallel.transport() != "rccl" and not host_asked:
        raise SystemExit(f"--gpus {world}: the exchange step would run over the "
                         f"'{parallel.transport()}' transport, not RCCL (ranks sharing a "
                         "GPU?); set MDX_COMM_TRANSPORT=host to measure that on purpose")
    local_rank = int(os.environ.get("MDX_LOCAL_RANK") or os.environ.get("LOCAL_RANK") or rank)
    engine.set_device(local_rank % _lib.device_count())

    def barrier_sync():
        engine.synchronize()
        parallel.barrier()

    work = WORKLOADS[args.workload]
    n_per_species = list(work["species"])
    box = work["box"]
    cutoff = np.float32(args.cutoff if args.cutoff > 0 else box / 2 - 0.1)
    nbins = int(cutoff / 0.01)
    shard_tiles = args.workload == "water"
    steps, warmup = args.steps, args.warmup
    if args.frames_per_step > 0:
        fps = args.frames_per_step
    else:
        fps = 64 if shard_tiles else max(1, TRAJECTORY_FRAMES // steps)
    # the timed steps walk once through the resident trajectory (100 000 frames per rank
    # with the default step size); warm-up steps re-read its first blocks
    n_frames = fps * steps

    # ---- resident synthetic trajectory, reference layout [atoms][frames][3]
    seed_shift = 0 if shard_tiles else 2 * rank  # tile shards see the same frames
    species = []
    for s_idx, n in enumerate(n_per_species):
        arr = engine.DeviceArray((n, n_frames, 3), np.float32)
        arr.fill_uniform(seed=1000 + s_idx + seed_shift, lo=0.0, hi=box)
        species.append(arr)

    rdf = engine.RdfEngine(n_per_species, [box] * 3, cutoff, nbins,
                           strict_reference_masks=True,
                           literal_kernel=args.kernel == "literal",
                           table_kernel=args.kernel == "table")
    shard = (rank, world) if shard_tiles else (0, 1)

    def step(i: int) -> None:
        rdf.accumulate(species, frames=np.arange(i * fps, (i + 1) * fps, dtype=np.int64),
                       shard_rank=shard[0], shard_count=shard[1])

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

    total_frames = steps * fps * (1 if shard_tiles else world)
    value = total_frames / elapsed
    ms_per_step = elapsed / steps * 1e3

    n_atoms = sum(n_per_species)
    pairs_per_frame = n_atoms * (n_atoms - 1) // 2
    bytes_per_frame = n_atoms * 3 * 4
    launches = max(tile_launches, 1)
    tile_avg_ms = tile_ms / launches
    frames_per_launch = steps * fps / launches
    kernel_name = {"fast": "rdf_fast_kernel", "table": "rdf_tile_kernel<false>",
                   "literal": "rdf_tile_kernel<true>"}[args.kernel]
    # counters on record belong to the headline geometry; they scale with the frames of a
    # launch, whatever --steps / --frames-per-step make of it
    on_record = args.workload == "nacl" and args.kernel == "fast" and args.cutoff <= 0 \
        and world == 1
    pmc = recorded(kernel_name, frames_per_launch, "rdf_frames_per_launch") if on_record else {}
    achieved_gbps = (bytes_per_frame * frames_per_launch) / (tile_avg_ms * 1e-3) / 1e9
    roofline = {
        "kernel": kernel_name,
        "bound": "hbm",
        "achieved": achieved_gbps,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved_gbps / HBM_PEAK_GBPS,
        "traffic": traffic_of(pmc),
        "traffic_source": f"{PMC_SUMMARY}: separate rocprofv3 --pmc FETCH_SIZE and "
                          "WRITE_SIZE passes over this command; bytes per launch = "
                          "2*FETCH_SIZE (gfx950 counts 128-B reads as 64 B) + WRITE_SIZE, "
                          "scaled by frames per launch "
                          f"({pmc.get('scaled_from_units_per_launch')} on record)",
        "traffic_over_algorithmic": (traffic_of(pmc) / (bytes_per_frame * frames_per_launch))
        if traffic_of(pmc) else None,
        "avg_launch_ms": tile_avg_ms,
        "launches": tile_launches,
        "algorithmic_bytes_per_launch": bytes_per_frame * frames_per_launch,
        "note": ("BASELINE.json asks for the HBM fraction; the pair loop reuses every "
                 "12-byte atom ~N/2 times from registers, so the physical limiter is "
                 "float32 vector issue, see valu"),
    }
    # float32 vector issue: cycles per pair from the fast kernel's instruction mix at the
    # measured per-instruction rates, against the kernel time measured in this run
    share = (1.0 / world) if shard_tiles else 1.0
    pairs_per_s_kernel = pairs_per_frame * share * frames_per_launch / (tile_avg_ms * 1e-3)
    cycles_per_pair_folded = sum(VALU_CYCLES[k] * n for k, n in FAST_LOOP_MIX.items()) \
        / PAIRS_PER_LOOP_ITERATION
    # sorted frames: a component whose two 64-atom groups lie within half a box of each other
    # is taken without its two fold instructions; the share is estimated on frame 0
    unfolded = None
    if args.kernel == "fast" and os.environ.get("MDX_RDF_UNFOLD") != "0":
        first = []
        for s_idx, n in enumerate(n_per_species):
            idx = (np.arange(1, n, dtype=np.uint64)[:, None] * np.uint64(n_frames * 3)
                   + np.arange(3, dtype=np.uint64)[None, :])   # strict masks: atom 0 left out
            first.append(engine.uniform_fill_at(idx, 1000 + s_idx + seed_shift, 0.0, box)
                         .reshape(n - 1, 3))
        group_edge = min(box * (64.0 / (n - 1)) ** (1 / 3) for n in n_per_species)
        if group_edge < 0.3 * box and max(n_per_species) > 7 * 256:
            unfolded = fold_free_share(first, box)
    cycles_per_pair = cycles_per_pair_folded - \
        (6 * VALU_CYCLES["sub_abs"] * unfolded["share"] if unfolded else 0.0)
    issue_bound = 1024 * 2.4e9 * 64 / cycles_per_pair
    valu = {
        "pairs_per_s_kernel": pairs_per_s_kernel,
        "pairs_per_s_end_to_end": pairs_per_frame * value / (1 if shard_tiles else world),
        "pack_kernel_ms_total": pack_ms,
        "tile_kernel_ms_total": tile_ms,
    }
    if args.kernel == "fast":
        valu.update({
            "model": "instruction mix of the hot loop x sustained cycles per wave64 "
                     "instruction per SIMD (profiles/r02_valu_probe.txt), 1024 SIMDs, 2.4 GHz",
            "valu_cycles_per_pair_model": cycles_per_pair,
            "valu_cycles_per_pair_every_component_folded": cycles_per_pair_folded,
            "fold_free_components": unfolded,
            "issue_bound_pairs_per_s": issue_bound,
            "frac_of_issue_bound": pairs_per_s_kernel / issue_bound,
        })
    if "SQ_INSTS_VALU" in pmc:
        # wave-level pair evaluations of a frame: species are padded to tiles of 256 slots
        # (first atom left out under the strict masks); a pair of different tiles costs
        # 256 j-atoms for each of its four waves, a tile paired with itself 256 + 192 + 128
        # + 64 (a wave skips the j-atoms below its first lane)
        tiles = [-(-(n - 1) // 256) for n in n_per_species]
        slots_per_frame = sum(t * (t - 1) // 2 * 1024 + t * 640 for t in tiles) + sum(
            tiles[a] * tiles[b] * 1024 for a in range(len(tiles)) for b in range(a + 1, len(tiles)))
        slots = slots_per_frame * frames_per_launch * share
        valu["valu_insts_per_launch_pmc"] = pmc["SQ_INSTS_VALU"]
        valu["valu_insts_per_wave_pair_slot_pmc"] = pmc["SQ_INSTS_VALU"] / slots
        valu["cycles_per_wave_pair_slot_at_2p4GHz"] = (tile_avg_ms * 1e-3 * 2.4e9 * 1024) / slots

    if shard_tiles:
        parallelism = (f"atom-pair tiles of every frame dealt cyclically to {world} rank(s), "
                       "one histogram all-reduce")
    else:
        parallelism = f"frames sharded over {world} rank(s), one histogram all-reduce"
    result = {
        "metric": "frames/sec (RDF + GK-ACF) on 10k-atom synthetic trajectory; %HBM-roofline",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if shard_tiles else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"RadialDistributionFunction, {work['label']}, strict reference masks "
                        f"(BASELINE.json configs[{3 if shard_tiles else 1}]); "
                        f"{steps * fps} frames per rank resident in HBM, each histogrammed "
                        "once in the timed steps",
            "frames_per_step": fps,
            "frames_per_rank": steps * fps,
            "parallelism": parallelism,
            "kernel": args.kernel,
            "fold_free_loop_variants": unfolded is not None,
        },
        "roofline": roofline,
        "valu": valu,
        "pairs_counted": int(hist_total.sum()),
        # the exchange step: which transport carried it, how many ranks the RCCL
        # communicator itself reports, and what the one histogram all-reduce took
        "transport": parallel.transport(),
        "rccl_ranks": parallel.comm_ranks(),
        "allreduce_ms": allreduce_ms,
        "gk_acf_frames_per_s": None,
    }
    if world > 1 and result["transport"] == "rccl" and result["rccl_ranks"] != world:
        raise SystemExit(f"RCCL communicator spans {result['rccl_ranks']} rank(s), the job "
                         f"has {world}")

    # side measurements: one-rank runs only (with N > 1 the other ranks would sit idle,
    # and the CPU baselines belong to the N = 1 line)
    side = rank == 0 and world == 1 and not shard_tiles and args.cutoff <= 0
    if side and args.cpu_frames > 0:
        result["cpu_baseline"] = cpu_baseline(engine, species, n_frames, args.cpu_frames, rank)
    if side and args.literal_frames > 0:
        result["cpu_baseline_literal"] = cpu_baseline_literal(engine, species, n_frames,
                                                              args.literal_frames, rank)
    rdf.close()
    for arr in species:
        arr.free()
    if side:
        if args.gk_frames > 0:
            result["gk_acf"] = gk_side_measurement(engine, args.gk_frames,
                                                   with_cpu=args.cpu_frames > 0)
            # the second half of the metric: frames/s of the GreenKuboIonicConductivity
            # chain (configs[2]) on this GPU, measured outside the timed steps
            result["gk_acf_frames_per_s"] = result["gk_acf"]["frames_per_s"]
        if args.flux_atoms > 0:
            result["flux_reductions"] = flux_side_measurement(engine, args.flux_atoms, 100000)
            result["transformations"] = transformation_side_measurement(
                engine, args.flux_atoms, 100000)
            result["adf"] = adf_side_measurement(engine)
        if not args.skip_host_input:
            result["host_input"] = host_input_measurement(engine, min(fps, 1024))
        if args.api_frames > 0:
            result["api_end_to_end"] = api_measurement(args.api_frames)
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


def cpu_baseline_literal(engine, species, n_frames: int, sample: int, rank: int) -> dict:
    """The NumPy restatement that follows the reference line by line (index lists of the
    upper triangle, gathers, minimum image, boolean masks, the fixed-width histogram rule;
    oracle/rdf.py) on the first frame(s) of the same trajectory: the closest thing to the
    reference's own CPU path that can run here, one host thread plus NumPy's."""
    from oracle import rdf as literal

    sample = min(sample, n_frames)
    host = []
    for s_idx, n in enumerate((N_NA, N_CL)):
        seed = 1000 + s_idx + 2 * rank
        idx = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(n_frames * 3)
               + np.arange(sample * 3, dtype=np.uint64)[None, :])
        host.append(engine.uniform_fill_at(idx, seed, 0.0, BOX).reshape(n, sample, 3))
    t0 = time.perf_counter()
    ref = literal.rdf_histograms(host, ["Na", "Cl"], [BOX] * 3, CUTOFF, NBINS, rdf_minibatch=500)
    dt = time.perf_counter() - t0
    check = engine.RdfEngine([N_NA, N_CL], [BOX] * 3, CUTOFF, NBINS)
    check.accumulate(species, frames=np.arange(sample, dtype=np.int64))
    got = check.read()
    check.close()
    same = all(np.array_equal(got[k], ref[name].astype(np.uint64))
               for k, name in enumerate(literal.key_names(["Na", "Cl"])))
    return {
        "value": sample / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {sample} frame(s) of the same 10k-atom trajectory, {dt:.2f} s wall, "
                  "oracle/rdf.py (NumPy, rows of 500 atoms per tile)",
        "gpu_matches_bit_exact": bool(same),
    }


def gk_cpu_baseline(engine, series_f32: np.ndarray, acf_gpu: np.ndarray, data_range: int,
                    decay: float, kick: float) -> dict:
    """CPU side of the Green-Kubo chain (C restatement, OpenMP): the atom reduction on a
    2 000-frame block of 10 000 atoms, and the windowed autocorrelation of the full ionic
    current series the GPU chain produced.  Both stages are also compared with the GPU."""
    from oracle import c_oracle

    threads = c_oracle.num_threads()
    sample_frames = 2000
    block = []
    for seed, n in ((3, N_NA), (4, N_CL)):
        dev = engine.DeviceArray((n, sample_frames, 3), np.float32)
        dev.fill_langevin(seed=seed, decay=decay, kick=kick)
        block.append((dev, dev.to_numpy()))
    t0 = time.perf_counter()
    cur = c_oracle.weighted_atom_sum(block[0][1], 1.0) - c_oracle.weighted_atom_sum(block[1][1], 1.0)
    dt_sum = time.perf_counter() - t0
    got = engine.weighted_atom_sum(block[0][0], scalar_weight=1.0)
    got = engine.weighted_atom_sum(block[1][0], scalar_weight=-1.0, out=got, accumulate=True)
    for dev, _ in block:
        dev.free()
    sum_err = float(np.max(np.abs(got - cur)) / np.max(np.abs(cur)))

    n_frames = series_f32.shape[1]
    n_win = n_frames - data_range
    t0 = time.perf_counter()
    acf = c_oracle.acf_windows(series_f32.astype(np.float64), 1, 0, n_win,
                               np.arange(data_range, dtype=np.int32))
    dt_acf = time.perf_counter() - t0
    acf_err = float(np.max(np.abs(acf_gpu - acf)) / np.max(np.abs(acf)))
    reduce_rate = sample_frames / dt_sum
    acf_rate = n_frames / dt_acf
    return {
        # frames/s of the chain when both stages run at their measured rates
        "value": 1.0 / (1.0 / reduce_rate + 1.0 / acf_rate),
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"atom reduction: {sample_frames} frames x 10k atoms, {dt_sum:.3f} s wall; "
                  f"windowed autocorrelation: all {n_win} windows of the {n_frames}-frame "
                  f"series, {dt_acf:.2f} s wall; oracle/mdx_oracle.c with OpenMP; value "
                  "composes the two measured stage rates",
        "reduce_frames_per_s": reduce_rate,
        "acf_frames_per_s": acf_rate,
        "gpu_reduce_max_rel_diff": sum_err,
        "gpu_acf_max_rel_diff": acf_err,
        "gpu_matches_1e-6": bool(sum_err <= 1e-6 and acf_err <= 1e-6),
    }


def flux_side_measurement(engine, n_atoms: int, n_frames: int) -> dict:
    """The reductions that turn per-atom stress / velocity / energy arrays into the series
    GreenKuboViscosity and GreenKuboThermalConductivity correlate (configs[4] stage A):
    HBM-bound streams, timed with device-resident input."""
    stress = engine.DeviceArray((n_atoms, n_frames, 6), np.float32)
    vel = engine.DeviceArray((n_atoms, n_frames, 3), np.float32)
    ke = engine.DeviceArray((n_atoms, n_frames, 1), np.float32)
    pe = engine.DeviceArray((n_atoms, n_frames, 1), np.float32)
    for seed, arr in enumerate((stress, vel, ke, pe)):
        arr.fill_uniform(seed=700 + seed, lo=-1.0, hi=1.0)
    cell = n_atoms * n_frames * 4
    cases = {
        "momentum_flux": (lambda: engine.momentum_flux(stress), 6 * cell,
                          "weighted_sum_kernel<float, 4, true>",
                          "whole [t][6] rows: the kept components 3..5 share their sectors "
                          "with the other three (needed bytes are half of this)"),
        "thermal_flux": (lambda: engine.thermal_flux(stress, vel, ke, pe), 11 * cell,
                         "flux_stream_kernel<true>", "stress 6 + velocity 3 + two energies"),
        "energy_weighted_sum": (lambda: engine.energy_weighted_sum(vel, ke, pe), 5 * cell,
                                "flux_stream_kernel<false>", "vector 3 + two energies"),
    }
    out = {"workload": f"{n_atoms} atoms x {n_frames} frames, float32, device resident"}
    reps = 3
    for name, (fn, nbytes, kernel, what) in cases.items():
        fn()  # warm-up
        engine.profile_enable(True)
        engine.profile_reset()
        for _ in range(reps):
            fn()
        engine.synchronize()
        engine.profile_enable(False)
        ms, launches = engine.profile_query("flux")
        avg = ms / max(launches, 1)
        achieved = nbytes / (avg * 1e-3) / 1e9
        pmc = recorded(kernel) if (n_atoms, n_frames) == (5000, 100000) else {}
        out[name] = {"kernel": kernel, "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic_of(pmc), "avg_launch_ms": avg, "launches": launches,
                     "algorithmic_bytes_per_launch": nbytes, "algorithmic_bytes": what}
    for arr in (stress, vel, ke, pe):
        arr.free()
    return out


def transformation_side_measurement(engine, n_atoms: int, n_frames: int) -> dict:
    """The per-atom coordinate transformations and the Kinaci heat current: single passes
    over ``[atoms][frames][3]`` float32 arrays (read + write, or read only for the heat
    current), timed with device-resident input.  The unwrapped input is a slow
    Ornstein-Uhlenbeck walk (0.09 per frame, box 65), wrapped into the box on the device,
    so that an atom crosses a face about once in a thousand frames, as in a trajectory."""
    box = [BOX] * 3
    walk = engine.DeviceArray((n_atoms, n_frames, 3), np.float32)
    walk.fill_langevin(seed=900, decay=1.0 - 1e-5, kick=20.0 * (2e-5) ** 0.5)
    wrapped = engine.DeviceArray((n_atoms, n_frames, 3), np.float32)
    engine.map_coordinates(engine.MAP_WRAP, walk, box, out=wrapped)
    out = engine.DeviceArray((n_atoms, n_frames, 3), np.float32)
    cell = n_atoms * n_frames * 4
    cases = {
        "unwrap_coordinates": (lambda: engine.unwrap_coordinates(wrapped, box, out=out),
                               6 * cell, "transform", "unwrap_stream_kernel<4>",
                               "3 words read + 3 written per atom and frame"),
        "wrap_coordinates": (lambda: engine.map_coordinates(engine.MAP_WRAP, walk, box, out=out),
                             6 * cell, "transform", "map_kernel<0, float, float>",
                             "3 words read + 3 written"),
        "scale_coordinates": (lambda: engine.map_coordinates(engine.MAP_SCALE, walk, box,
                                                             out=out),
                              6 * cell, "transform", "map_kernel<2, float, float>",
                              "3 words read + 3 written"),
        "velocity_from_positions": (lambda: engine.map_coordinates(
            engine.MAP_VELOCITY, walk, [0.002, 0, 0], out=out), 6 * cell, "transform",
            "map_kernel<4, float, float>", "3 words read + 3 written"),
    }
    result = {"workload": f"{n_atoms} atoms x {n_frames} frames, float32 in and out, device "
                          "resident"}
    reps = 3

    def timed(fn, slot):
        fn()  # warm-up
        engine.profile_enable(True)
        engine.profile_reset()
        for _ in range(reps):
            fn()
        engine.synchronize()
        engine.profile_enable(False)
        ms, launches = engine.profile_query(slot)
        return ms / max(launches, 1), launches

    on_record = (n_atoms, n_frames) == (5000, 100000)

    def entry(kernel, avg, launches, nbytes, what, kernels_on_record=None):
        achieved = nbytes / (avg * 1e-3) / 1e9
        traffic = None
        if on_record:
            parts = [traffic_of(recorded(k)) for k in (kernels_on_record or [kernel])]
            traffic = sum(parts) if all(p is not None for p in parts) else None
        return {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "avg_launch_ms": avg, "launches": launches,
                "algorithmic_bytes_per_launch": nbytes, "algorithmic_bytes": what}

    for name, (fn, nbytes, slot, kernel, what) in cases.items():
        avg, launches = timed(fn, slot)
        result[name] = entry(kernel, avg, launches, nbytes, what)
    walk.free()
    out.free()
    # Kinaci heat current: position, velocity, force (3 words each) and the potential
    # energy (1) per atom and frame; the streaming pass reads them once and a weighted column
    # sum reads the positions again (13 words fetched for these 10)
    vel = engine.DeviceArray((n_atoms, n_frames, 3), np.float32)
    force = engine.DeviceArray((n_atoms, n_frames, 3), np.float32)
    pe = engine.DeviceArray((n_atoms, n_frames, 1), np.float32)
    for seed, arr in enumerate((vel, force, pe)):
        arr.fill_uniform(seed=910 + seed, lo=-1.0, hi=1.0)
    avg, launches = timed(lambda: engine.kinaci_heat_current(wrapped, vel, force, pe, 0.002),
                          "flux")
    result["kinaci_heat_current"] = entry(
        "kinaci_stream_kernel + kinaci_offsets_kernel + sliced_weighted_sum_kernel", avg,
        launches, 10 * cell, "position 3 + velocity 3 + force 3 + potential energy 1; the "
                             "positions are read a second time (13 words fetched)",
        kernels_on_record=["kinaci_stream_kernel", "kinaci_offsets_kernel",
                           "sliced_weighted_sum_kernel"])
    for arr in (wrapped, vel, force, pe):
        arr.free()
    return result


def adf_side_measurement(engine) -> dict:
    """AngularDistributionFunction's batch body (neighbour search within the cutoff, all
    triples of every atom's neighbourhood, weighted angle histograms) on the 10k-atom melt:
    not part of the metric, reported so that the kernel has a number next to the others."""
    n_frames = 16
    species = []
    for s_idx, n in enumerate((N_NA, N_CL)):
        arr = engine.DeviceArray((n, n_frames, 3), np.float32)
        arr.fill_uniform(seed=60 + s_idx, lo=0.0, hi=BOX)
        species.append(arr)
    edges = np.histogram_bin_edges(np.zeros(1, np.float32), bins=500, range=(0.0, 3.15))
    out = {"workload": f"{N_NA} + {N_CL} atoms uniform in a box of {BOX}, {n_frames} frames per "
                       "call, 500 angle bins, device resident"}
    for cutoff in (6.0, 10.0):
        engine.adf_accumulate(species, [BOX] * 3, cutoff, edges)
        engine.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            _, triples = engine.adf_accumulate(species, [BOX] * 3, cutoff, edges)
        engine.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[f"cutoff_{cutoff:g}"] = {"frames_per_s": n_frames / dt, "ms_per_call": dt * 1e3,
                                     "triples_per_frame": float(triples.sum()) / n_frames,
                                     "triples_per_s": float(triples.sum()) / dt}
    for arr in species:
        arr.free()
    return out


def api_measurement(n_frames: int) -> dict:
    """The same two calculators as a user runs them: a 10k-atom trajectory written to the
    on-disk store, then ``project.run.RadialDistributionFunction(number_of_configurations
    =-1)`` and ``project.run.GreenKuboIonicConductivity()`` (store reads, batching, PCIe,
    kernels, result files)."""
    import shutil
    import tempfile

    import lammps_analysis_amd as mds
    from lammps_analysis_amd.database.mdsuite_properties import mdsuite_properties
    from lammps_analysis_amd.database.simulation_database import (SpeciesInfo,
                                                                  TrajectoryChunkData,
                                                                  TrajectoryMetadata)
    from lammps_analysis_amd.file_io.file_read import FileProcessor

    props = [mdsuite_properties.positions, mdsuite_properties.velocities]
    species = [SpeciesInfo(name=name, n_particles=n, properties=props, charge=q)
               for name, n, q in (("Na", N_NA, 1.0), ("Cl", N_CL, -1.0))]
    meta = TrajectoryMetadata(species_list=species, n_configurations=n_frames, sample_rate=1,
                              box_l=[BOX] * 3)
    rng = np.random.default_rng(7)
    chunk_frames = 500

    class Synthetic(FileProcessor):
        def __str__(self):
            return "synthetic"

        def _get_metadata(self):
            return meta

        def get_configurations_generator(self):
            for f0 in range(0, n_frames, chunk_frames):
                nf = min(chunk_frames, n_frames - f0)
                chunk = TrajectoryChunkData(species_list=species, chunk_size=nf)
                for sp in species:
                    chunk.add_data(rng.random((nf, sp.n_particles, 3), dtype=np.float32)
                                   * np.float32(BOX), 0, sp.name, "Positions")
                    chunk.add_data(rng.standard_normal((nf, sp.n_particles, 3),
                                                       dtype=np.float32), 0, sp.name,
                                   "Velocities")
                yield chunk

    workdir = tempfile.mkdtemp(prefix="mdx_bench_", dir=os.environ.get("TMPDIR") or None)
    try:
        project = mds.Project(name="bench", storage_path=workdir)
        t0 = time.perf_counter()
        project.add_experiment("melt", timestep=0.002, temperature=1400.0, units="metal",
                               simulation_data=Synthetic())
        ingest_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        rdf = project.run.RadialDistributionFunction(number_of_configurations=-1, plot=False)
        rdf_first_s = time.perf_counter() - t0
        # the first calculator call of a process also imports the calculators package
        # (SciPy: 0.37 s) and loads the kernels' code object; a second run with another
        # bin count (no cached result) shows the rate of the path itself
        t0 = time.perf_counter()
        project.run.RadialDistributionFunction(number_of_configurations=-1, plot=False,
                                               number_of_bins=NBINS - 240)
        rdf_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        project.run.GreenKuboIonicConductivity(data_range=500, plot=False)
        gk_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        project.run.GreenKuboIonicConductivity(data_range=500, correlation_time=2, plot=False)
        gk_again_s = time.perf_counter() - t0
        keys = sorted(rdf["melt"].data_dict)
    finally:
        from lammps_analysis_amd.database.simulation_database import release_read_maps
        release_read_maps()  # the store's files are only freed once nothing maps them
        shutil.rmtree(workdir, ignore_errors=True)
    rdf_frames = n_frames - 1   # number_of_configurations=-1 samples n - 1 frames
    return {
        "workload": f"10k-atom NaCl trajectory of {n_frames} frames in the on-disk store "
                    f"({2 * n_frames * 120000 / 1e9:.1f} GB of float32 positions and "
                    "velocities under $TMPDIR, page cache warm: just written)",
        "ingest_s": ingest_s,
        "rdf_frames_per_s": rdf_frames / rdf_s,
        "rdf_s": rdf_s,
        "rdf_first_call_s": rdf_first_s,
        "rdf_keys": keys,
        "gk_ionic_frames_per_s": n_frames / gk_s,
        "gk_ionic_s": gk_s,
        "gk_ionic_series_stored_s": gk_again_s,
        "note": "project.run.RadialDistributionFunction(number_of_configurations=-1) and "
                "project.run.GreenKuboIonicConductivity(data_range=500), the latter including "
                "the IonicCurrent transformation the first time; wall time of the call, "
                "result files included; rdf_s is a second run with another bin count, "
                "rdf_first_call_s the process's first calculator call (package import and "
                "code-object load included, profiles/r03_api_profile.txt)",
    }


def gk_side_measurement(engine, n_frames: int, with_cpu: bool = True) -> dict:
    """BASELINE.json configs[2] in small: ionic current of a 10k-atom velocity block
    (HBM-bound atom reduction) followed by the windowed autocorrelation."""
    data_range = 500
    vel_na = engine.DeviceArray((N_NA, n_frames, 3), np.float32)
    vel_cl = engine.DeviceArray((N_CL, n_frames, 3), np.float32)
    # per-atom Langevin velocities with the parameters of the reference's own test
    # (kT = 0.7, m = 1.5, gamma = kT / 1.2345, dt = 0.1)
    kt, mass, dt = 0.7, 1.5, 0.1
    gamma = kt / 1.2345
    decay, kick = 1.0 - gamma * dt / mass, np.sqrt(2.0 * kt * gamma * dt) / mass
    vel_na.fill_langevin(seed=3, decay=decay, kick=kick)
    vel_cl.fill_langevin(seed=4, decay=decay, kick=kick)
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

    cur_host, acf_host = chain_via_host()  # warm-up of both variants; they must agree
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
        "workload": f"GreenKuboIonicConductivity chain, 10k atoms x {n_frames} frames of "
                    f"Langevin velocities, charges +1/-1, data_range {data_range}, "
                    "correlation_time 1 (BASELINE.json configs[2])",
        "frames_per_s": n_frames / dt,
        "ms_per_pass": dt * 1e3,
        "ms_per_pass_series_via_host": dt_host * 1e3,
        "roofline_atom_sum": {
            "kernel": "weighted_sum_kernel<float, 4, false>", "bound": "hbm", "achieved": achieved,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic_of(recorded("weighted_sum_kernel<float, 4, false>"))
            if n_frames == 100000 else None,
            "avg_launch_ms": avg_ms, "launches": sum_n,
            "algorithmic_bytes_per_launch": bytes_per_launch},
        "acf_slide_kernel_ms": slide_ms / max(slide_n, 1),
    }
    if with_cpu:
        out["cpu_baseline"] = gk_cpu_baseline(engine, cur_host.astype(np.float32)[None],
                                              acf_host, data_range, decay, kick)

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
        "kernel": "band4_kernel<float, 33, false>",
        "mfma_f64_sustained_tflops": 74.0,
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
        "kernel": "band4_kernel<float, 27, true>",
        # algorithmic count: subtract, multiply, add per origin, lag, atom and component
        "band_kernel_tflops_f64": (3.0 * n_frames * msd_range * N_NA * 3
                                   / (msd_avg * 1e-3) / 1e12) if msd_avg else None,
        # what the matrix cores execute: 128-frame tiles x 27 lag blocks of 4, 2 flops per
        # element, row and component
        "mfma_tflops_issued": (2.0 * 128 * math.ceil(n_frames / 128) * 108 * N_NA * 3
                               / (msd_avg * 1e-3) / 1e12) if msd_avg else None,
        # v_mfma_f64_4x4x4_4b_f64 back to back (scratch/mfma/clock.hip); the 16x16x4
        # instruction sustains 48
        "mfma_f64_sustained_tflops": 74.0,
    }
    vel_na.free()
    vel_cl.free()
    cur_dev.free()
    series_dev.free()
    return out


if __name__ == "__main__":
    main()
