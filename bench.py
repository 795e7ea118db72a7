#!/usr/bin/env python3
"""Benchmark of the scan-matching hot path on MI355X.

Metric (BASELINE.json): candidate poses scored per second (and loop-closure constraints,
i.e. completed Match* calls, per second).

  N = 1 (default)   BASELINE config[1] "C2": one loop-closure search per step --
                    FastCorrelativeScanMatcher2D::MatchFullSubmap of a 1000-point scan against
                    one 400x400 submap, depth 7, full-angle.  Stack and point cloud are
                    resident in HBM before the timed region; only the result returns.
                    The other BASELINE configs (C1 real-time 2D single + batched, one GPU's
                    share of C3, C4 real-time 3D, C5 fast 3D) are measured after the timed
                    region and reported under config.other (skip with --no-other).
  N > 1             BASELINE config[2] "C3": one scan against 64 N DISTINCT submaps (seeds
                    42 ..., N = 8: the 512 submaps of the config), submap-sharded with
                    sharding.shard_range, every rank searching its own block.  Per step the
                    ranks exchange what the reference's ConstraintBuilder collects -- one
                    optional constraint per submap (all-gather, 48 B per submap) -- and agree on
                    the node-wide best match (all-reduce(max) of one packed 8-byte key), both on
                    RCCL over xGMI.  Per-GPU work is fixed as N grows: weak scaling.
  --config c1|c2|c3|c4|c5 selects the timed workload explicitly (single GPU for c1/c4/c5).

Timed regions run the library as a production caller does: WITHOUT the HIP-event brackets behind
cmx_match_stats' *_ms fields (they are packets the chain of launches waits behind: 24 of a single
search's 156 us).  The kernel / device times of the roofline blocks come from a few untimed
"instrumented" passes of the same workload right behind each region, with the brackets switched
on (set_timing, instrumented; the headline: one extra step after the K timed ones).

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
# searches issued from sixteen host threads share four queues, and kernels of different streams on
# one queue wait for each other (16 threads: 21 000 matches/s on 4 queues, 27 000 on 16).  A
# deployment setting of the runtime, read when it initialises -- so before torch is imported; the
# library sets the same default when it is loaded first (cmx_common.hip).  An explicit value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0        # HBM3E spec (6.3 TB/s achievable by a copy)
L2_PEAK_GBS = 34500.0        # aggregate L2 -> L1
LDS_PEAK_GBS = 150000.0      # ds_read_b64/b128, every CU streaming
LDS_B32_PEAK_GBS = 75000.0   # ds_read_b32: half of it (the bound kernel's aligned dword reads)

C3_SUBMAPS_PER_GPU = 64      # 512 submaps over the 8 GPUs of BASELINE config[2]
C5_SUBMAPS_PER_GPU = 32      # 256 submaps over the 8 GPUs of BASELINE config[4]
C3_POSITIVE = 137            # the scan is drawn from submap #137


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=200,
                    help="untimed steps; a C2 step is ~0.2 ms, so the default also lets clocks settle")
    ap.add_argument("--config", choices=["auto", "c1", "c2", "c3", "c4", "c5"], default="auto",
                    help="timed workload; auto = c2 on one GPU, c3 (sharded) on several")
    ap.add_argument("--submaps", type=int, default=0,
                    help="submaps searched per step per GPU (default: 1 for c2, 64 for c3)")
    ap.add_argument("--matches", type=int, default=128,
                    help="c1: independent real-time matches per step (one per trajectory / robot)")
    ap.add_argument("--grid", type=int, default=400)
    ap.add_argument("--depth", type=int, default=7)
    ap.add_argument("--beams", type=int, default=1000)
    ap.add_argument("--min-score", type=float, default=0.6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-submaps", type=int, default=4,
                    help="c2 / c3: submaps of this rank's block the parity gate searches with the "
                         "reference (0.4 s of one host core each; 512 = every submap of config[2])")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the device-vs-reference gate in front of the timed regions "
                         "(profiling runs only: the line then carries no `parity`)")
    ap.add_argument("--no-other", action="store_true",
                    help="skip the C1 / C3-share / C4 / C5 measurements reported under config.other")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--concurrency", type=int, default=0,
                    help="host threads issuing the passes of a step concurrently (the reference's "
                         "thread-pool fan-out over independent searches, "
                         "constraint_builder_2d.cc:97-111); the C ABI is re-entrant: every call "
                         "leases its own stream + scratch.  0 = auto: 16 for c2 on one GPU (a single "
                         "search is a latency chain that fills a fraction of the chip; round 6: "
                         "16 threads on 16 hardware queues, until then 8 on the runtime's 4), else 1")
    ap.add_argument("--scans", type=int, default=8,
                    help="c2 / c3: distinct scans (poses of the same world) the passes of a step "
                         "cycle through.  Default 8 since round 6: the one scan rounds 1 - 5 timed "
                         "(--scans 1, still measured as the `c2_easy` leg) is an easy search -- its "
                         "dive finds a tight bound at once, 3 000 nodes are expanded -- while other "
                         "poses of the same world expand 17 000 - 93 000")
    ap.add_argument("--c1-distinct", type=int, default=0,
                    help="c1: distinct (grid, scan, pose) triples of a batch (0 = one per match up "
                         "to 128 matches, 256 for larger batches)")
    ap.add_argument("--passes-per-step", type=int, default=0,
                    help="passes of the hot path that make one step (a step is one pass over a "
                         "BATCH of searches).  0 = auto: calibrated during warmup so that a step "
                         "lasts >= 30 ms -- the driver's 20 timed steps are then >= 0.6 s, not 3 ms")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the collectives even with one rank (plumbing test)")
    ap.add_argument("--pmc-dir", default=os.path.join(ROOT, "profiles"),
                    help="directory with <tag>_pmc_{fetch,write}_size.csv of THIS command "
                         "(tools/profile_all.sh writes them); roofline.traffic is null without")
    ap.add_argument("--pmc-tag", default="r06")
    ap.add_argument("--details", default=os.path.join(ROOT, "gpurun_out", "bench_details.json"),
                    help="file the FULL record goes to (per-config blocks, notes, nested "
                         "rooflines); the line on stdout is the compact headline (< 4 KB); "
                         "'' = do not write it")
    return ap.parse_args()


def _cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 32))


def pmc_traffic_bytes(pmc_dir, tag, kernel, config=""):
    """HBM-side bytes per launch of `kernel` from the rocprofv3 PMC passes of ONE bench command:
    tools/profile_all.sh writes <tag>_pmc_{fetch,write}_size.csv for the default (C2) command
    and <tag>_<config>_pmc_... for `--config c1` (128 matches), `c3 --submaps 16`, `c4` and
    `c5 --submaps 32`.  FETCH_SIZE and WRITE_SIZE need separate passes (MI355X_MICROARCH.md
    "rocprofv3 PMC slots"); values are KiB; FETCH_SIZE is doubled per the guide's gfx950
    correction.  None when that pass does not exist or does not contain the kernel (a rename,
    another batch size): a number of some other command is worse than none."""
    import csv
    stem = f"{tag}_{config}" if config else tag
    total = 0.0
    for suffix, factor in (("_pmc_fetch_size.csv", 2.0), ("_pmc_write_size.csv", 1.0)):
        path = os.path.join(pmc_dir, stem + suffix)
        if not os.path.exists(path):
            return None
        with open(path) as f:
            rows = [r for r in csv.DictReader(f) if kernel in r["Kernel"]]
        if not rows:
            return None
        total += float(rows[0]["MeanValue"]) * 1024.0 * factor
    return total


# --------------------------------------------------------------------------------------
# CPU baseline (C2): the reference's own source where oracle/_ref is built
# --------------------------------------------------------------------------------------
def cpu_baseline_reference(cells, lim, depth, scans, min_score, seconds):
    """The reference's OWN fast_correlative_scan_matcher_2d.cc (oracle/_ref, built in place from
    /root/reference by __graft_entry__.build(); the prebuilt .so travels to the GPU box) timed
    on this box's host cores: one MatchFullSubmap per thread, like the reference's thread pool
    runs them (const methods, concurrent calls on one matcher), call i with scan i mod len(scans)
    -- the scans the device's passes cycle through.  Candidates are counted with the oracle port,
    whose search is bit-identical (tests/test_reference_ref.py).  Returns None when oracle/_ref
    is not available."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        return None
    cores = _cores()
    args = (cells, lim["resolution"], lim["max_x"], lim["max_y"], depth)
    port = orc.FastCorrelativeScanMatcher2D(*args)
    matcher = orc.ReferenceFastCorrelativeScanMatcher2D(*args)
    with ThreadPoolExecutor(max_workers=cores) as pool:
        counted = list(pool.map(lambda sc: port.match_full_submap(sc, min_score), scans))
    t0 = time.perf_counter()
    one = matcher.match_full_submap(scans[0], min_score)
    t_one = time.perf_counter() - t0
    assert one["found"] == counted[0]["found"]
    if one["found"]:
        assert np.float32(one["score"]) == np.float32(counted[0]["score"])
    per_match = [c["candidates_scored"] for c in counted]
    # Bounded sample: rounds of `cores` concurrent matches (ctypes releases the GIL) until
    # `seconds` have elapsed.
    t0 = time.perf_counter()
    matches = 0
    candidates = 0
    with ThreadPoolExecutor(max_workers=cores) as pool:
        while True:
            ks = [(matches + j) % len(scans) for j in range(cores)]
            list(pool.map(lambda k: matcher.match_full_submap(scans[k], min_score), ks))
            matches += cores
            candidates += sum(per_match[k] for k in ks)
            dt = time.perf_counter() - t0
            if dt >= seconds or matches >= 64 * cores:
                break
    return {
        "value": candidates / dt, "unit": "candidates/s", "cores": cores,
        "kind": "reference",
        "sample": f"{matches} MatchFullSubmap calls of the bench workload ({len(scans)} scans in "
                  f"turn) by the reference's own "
                  f"fast_correlative_scan_matcher_2d.cc (oracle/_ref: compiled in place with the reference's -O3 -DNDEBUG, "
                  f"stand-in Eigen value types; {min(per_match)} - {max(per_match)} candidates each -- the reference's "
                  f"depth-first search scores more candidates per match than the device schedule, "
                  f"each side counts its own -- {t_one * 1e3:.0f} ms single-thread for scan 0), {cores} threads, {dt:.1f} s",
        "single_thread_candidates_per_s": per_match[0] / t_one,
        "matches_per_s": matches / dt,
    }


def cpu_baseline(cells, lim, depth, scans, min_score, seconds):
    """CPU baseline on this box's host cores: the reference's own code when oracle/_ref is
    available, else the oracle (CPU restatement, "port"): one MatchFullSubmap per thread, the
    reference's thread-pool fan-out, over the scans the device's passes cycle through."""
    try:
        ref = cpu_baseline_reference(cells, lim, depth, scans, min_score, seconds)
        if ref is not None:
            return ref
    except Exception as e:   # never let the baseline leg break the bench line
        sys.stderr.write(f"reference baseline unavailable ({e}); timing the port instead\n")
    from oracle import pyoracle as orc
    cores = _cores()
    matcher = orc.FastCorrelativeScanMatcher2D(cells, lim["resolution"], lim["max_x"],
                                               lim["max_y"], depth)
    t0 = time.perf_counter()
    one = matcher.match_full_submap(scans[0], min_score)
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    total = 0
    rounds = 0
    while True:
        r = orc.fast2d_match_batch([matcher] * cores, scans[rounds % len(scans)], min_score, cores)
        total += r["candidates_scored"]
        rounds += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or rounds >= 64:
            break
    return {
        "value": total / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
        "sample": f"{rounds * cores} MatchFullSubmap calls of the bench workload ({len(scans)} scans "
                  f"in turn; {one['candidates_scored']} candidates for scan 0, {t_one * 1e3:.0f} ms "
                  f"single-thread), {cores} threads, {dt:.1f} s",
        "single_thread_candidates_per_s": one["candidates_scored"] / t_one,
        "matches_per_s": rounds * cores / dt,
    }


def _timed_threads(fn, cores, seconds, max_rounds=64):
    """Rounds of `cores` concurrent calls of fn() (ctypes releases the GIL) until `seconds` have
    elapsed; returns (calls, elapsed)."""
    from concurrent.futures import ThreadPoolExecutor
    calls = 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        while True:
            list(pool.map(lambda _: fn(), range(cores)))
            calls += cores
            dt = time.perf_counter() - t0
            if dt >= seconds or calls >= max_rounds * cores:
                return calls, dt


def cpu_baseline_c1(w, seconds):
    """BASELINE config[0] is the reference's own CPU case: its real_time_correlative_scan_matcher_2d.cc
    (oracle/_ref) on the scans and grids the device batch matches, one Match per host thread."""
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        return None
    cores = _cores()
    g, lim, scan, init = w.host_cells[0], w.host_lims[0], w.S[0], w.I[0]
    call = lambda: orc.ref_rt2d_match(g, lim["resolution"], lim["max_x"], lim["max_y"],  # noqa: E731
                                      [init.x, init.y, init.theta], scan, 0.3, math.radians(7.0),
                                      0.1, 0.1)
    t0 = time.perf_counter()
    call()
    t_one = time.perf_counter() - t0
    calls, dt = _timed_threads(call, cores, seconds)
    per_match = w.candidates_per_match
    return {"value": calls * per_match / dt, "unit": "candidates/s", "cores": cores,
            "kind": "reference", "matches_per_s": calls / dt,
            "single_thread_candidates_per_s": per_match / t_one,
            "sample": f"{calls} Match calls of the C1 workload by the reference's own "
                      f"real_time_correlative_scan_matcher_2d.cc (oracle/_ref), {per_match} "
                      f"candidates each, {cores} threads, {dt:.1f} s"}


def cpu_baseline_c4(w, seconds):
    """C4: the reference's real_time_correlative_scan_matcher_3d.cc over a contiguous RANGE of
    the 1 771 561 candidates (the whole search space is ten minutes on 8 cores: bounded sample,
    candidates/s extrapolates linearly -- every candidate costs the same 65 536 lookups)."""
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        return None
    cores = _cores()
    init = list(w.init.translation) + list(w.init.rotation)
    sample = 64 * cores
    t0 = time.perf_counter()
    total = 0
    while True:
        r = orc.ref_rt3d_match_mt(0.1, w.vox, init, w.cloud, 0.5, math.radians(2.0), 0.1, 0.1,
                                  cores, first_candidate=total, max_candidates=sample)
        total += sample
        dt = time.perf_counter() - t0
        if dt >= seconds or total >= r["num_candidates"]:
            break
    return {"value": total / dt, "unit": "candidates/s", "cores": cores, "kind": "reference",
            "matches_per_s": total / dt / r["num_candidates"],
            "sample": f"{total} of the {r['num_candidates']} candidates of the C4 search (generation "
                      f"order, from the first) scored by the reference's own "
                      f"real_time_correlative_scan_matcher_3d.cc (oracle/_ref: "
                      f"TransformPointCloud + ScoreCandidate per candidate), {cores} threads, "
                      f"{dt:.1f} s; a whole match would take {r['num_candidates'] / (total / dt):.0f} s"}


def cpu_baseline_c5(w, seconds):
    """C5: the reference's fast_correlative_scan_matcher_3d.cc (stack built once, outside the timed
    sample) matching the bench node against submap #0, one Match per host thread."""
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        return None
    cores = _cores()
    call = w.reference_match
    t0 = time.perf_counter()
    one = call()
    t_one = time.perf_counter() - t0
    calls, dt = _timed_threads(call, cores, seconds, max_rounds=16)
    return {"value": calls / dt, "unit": "matches/s", "cores": cores, "kind": "reference",
            "matches_per_s": calls / dt, "single_thread_match_s": t_one,
            "found": bool(one["found"]),
            "sample": f"{calls} Match calls of the C5 node against submap #0 by the reference's own "
                      f"fast_correlative_scan_matcher_3d.cc (oracle/_ref; precomputation stack built "
                      f"beforehand), {cores} threads, {dt:.1f} s"}


# --------------------------------------------------------------------------------------
# Parity gate ("parity gate before any timing is reported").  Every
# workload compares what the DEVICE returned with what the REFERENCE returns on the same bytes
# -- the reference's own sources (oracle/_ref) where the prebuilt library travelled with the
# repo, else the oracle port (bit-identical to them: tests/test_reference_ref*.py), for C4 the
# committed output of the reference (tests/golden/rt3d_c4_reference.json: ten CPU-minutes) --
# BEFORE its timed region; a mismatch beyond the north star's 1e-4 aborts the run.  The checker
# is never inside a timed region.
# --------------------------------------------------------------------------------------
PARITY_TOL = 1e-4


class ParityError(RuntimeError):
    pass


def _reference_kind():
    from oracle import pyoracle as orc
    try:
        return "reference" if orc.ref_lib() is not None else "port"
    except Exception:       # noqa: BLE001
        return "port"


def parity_record(kind, pairs):
    """pairs: (device_found, device_score, device_pose, ref_found, ref_score, ref_pose) per
    checked match; poses as flat sequences (None when not found)."""
    dscore = dpose = 0.0
    exact = True
    for i, (df, ds, dp, rf, rs, rp) in enumerate(pairs):
        if bool(df) != bool(rf):
            raise ParityError(f"parity: match {i}: device found={bool(df)} vs {kind} found={bool(rf)}")
        if not rf:
            continue
        a, b = float(np.float32(ds)), float(np.float32(rs))
        dscore = max(dscore, abs(a - b))
        d = float(np.max(np.abs(np.asarray(dp, np.float64) - np.asarray(rp, np.float64))))
        dpose = max(dpose, d)
        exact = exact and a == b and d == 0.0
    out = {"vs": kind, "checked": len(pairs), "max_abs_dscore": dscore, "max_abs_dpose": dpose,
           "bit_exact": exact, "tol": PARITY_TOL}
    if dscore > PARITY_TOL or dpose > PARITY_TOL:
        raise ParityError(f"parity: device differs from the {kind}: {out}")
    return out


def _parity_word(record):
    """One word for the summary: "exact", the largest difference, or None (not checked)."""
    if not record or not record.get("checked"):
        return None
    if record.get("bit_exact"):
        return "exact"
    return f"{max(record['max_abs_dscore'], record['max_abs_dpose']):.1e}"


def parity_gate(workload, result=None):
    """workload.parity(result) with the library's event brackets off; `result` = a device result
    of workload.search() (one is taken when None).  Raises ParityError; returns the record, or
    {"vs": None} for a leg that has no reference of its own (it repeats a checked workload)."""
    check = getattr(workload, "parity", None)
    if check is None:
        return {"vs": None, "checked": 0}
    if result is None:
        result = workload.search(0)
    return check(result)


# --------------------------------------------------------------------------------------
# Workloads.  Each exposes step() -> stats dict (candidates_scored, coarse_candidates,
# dominant_kernel_ms, device_ms, num_scans, nodes_expanded + workload keys), describe(),
# roofline(acc, steps) and `matches_per_step`.
# --------------------------------------------------------------------------------------
def _submap(synth, sm, seed, grid, depth, device):
    cells, lim, world = synth.make_submap(seed, grid, grid, 0.05, 30, 1000, 30.0, 0.01)
    g = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
    return sm.FastCorrelativeScanMatcher2D(g, depth, device=device), cells, lim, world


class Fast2DWorkload:
    """C2 (one submap per GPU) and C3 (a block of distinct submaps per GPU)."""

    def __init__(self, args, device, rank, world_size, sharded):
        from cartographer_amd import scan_matching as sm, sharding, synth
        self.sm, self.sharding, self.args = sm, sharding, args
        self.rank, self.world_size, self.sharded = rank, world_size, sharded
        if sharded:
            per_gpu = args.submaps or C3_SUBMAPS_PER_GPU
            self.total = per_gpu * world_size
            self.begin, self.end = sharding.shard_range(self.total, rank, world_size)
            positive = C3_POSITIVE % self.total
        else:
            self.total = args.submaps or 1
            self.begin, self.end = 0, self.total
            positive = 0
        self.matchers = []
        self.cells0 = self.lim0 = None
        self.host_submaps = []
        for gid in range(self.begin, self.end):
            m, cells, lim, world = _submap(synth, sm, 42 + gid, args.grid, args.depth, device)
            self.matchers.append(m)
            if self.cells0 is None:
                self.cells0, self.lim0 = cells, lim
            if len(self.host_submaps) < getattr(args, "parity_submaps", 4):   # (the parity gate's share of a block)
                self.host_submaps.append((cells, lim))
        # Every rank draws the same scan, from the world of the one true-positive submap.
        truth = synth.make_submap(42 + positive, args.grid, args.grid, 0.05, 30, 1000, 30.0,
                                  0.01)[2]
        pose = truth.free_pose(1234, 0.5)
        self.scan = truth.scan(pose, args.beams, 30.0, 0.01, 7)
        self.cloud = sm.PointCloudOnDevice(self.scan, device=device)
        # A step is a batch of passes: pass k searches with scan k mod `--scans` (scans taken at
        # different poses of the same world, resident in HBM like the first).
        self.clouds = [self.cloud]
        self.host_scans = [self.scan]
        for k in range(1, max(1, getattr(args, "scans", 1))):
            sk = truth.scan(truth.free_pose(1234 + k, 0.5), args.beams, 30.0, 0.01, 7 + k)
            self.host_scans.append(sk)
            self.clouds.append(sm.PointCloudOnDevice(sk, device=device))
        self.n_points = self.scan.shape[0]
        self.matches_per_step = len(self.matchers)
        self.positive = positive
        self.gathered = None
        self.best = None
        self._next = 0

    def search(self, k=None):
        """Pass k searches with scan k mod `--scans`; without k the scans are taken in turn."""
        if k is None:
            k = self._next
            self._next += 1
        return self.sm.match_full_submap_batch(self.matchers, self.clouds[k % len(self.clouds)],
                                               self.args.min_score)

    def parity(self, result):
        """EVERY scan the passes cycle through against the first submaps of this rank's block (at
        most `--parity-submaps`; 0.2 - 0.6 s of one host core per search), searched by the
        reference; found / score / pose against the device's result of the same (scan, submap)."""
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle as orc
        kind = _reference_kind()
        cls = (orc.ReferenceFastCorrelativeScanMatcher2D if kind == "reference"
               else orc.FastCorrelativeScanMatcher2D)
        a = self.args
        pairs = [(k, i) for k in range(len(self.host_scans)) for i in range(len(self.host_submaps))]

        def one(pair):
            k, i = pair
            cells, lim = self.host_submaps[i]
            return cls(cells, lim["resolution"], lim["max_x"], lim["max_y"],
                       a.depth).match_full_submap(self.host_scans[k], a.min_score)
        with ThreadPoolExecutor(min(len(pairs), _cores())) as pool:
            refs = list(pool.map(one, pairs))
        self.reference_result = refs[0]
        device = {0: result}        # (the gate's own search is pass 0: scan 0)
        for k in range(1, len(self.host_scans)):
            device[k] = Fast2DWorkload.search(self, k)
        return parity_record(kind, [(device[k][0][i], device[k][1][i], device[k][2][i], r["found"],
                                     r["score"], r["pose"]) for (k, i), r in zip(pairs, refs)])

    def exchange(self, found, scores, poses, torch_device):
        """What crosses xGMI per step: every submap's optional constraint to every rank (the
        reference's semantics) and the node-wide best match."""
        sh = self.sharding
        self.gathered = sh.all_gather_results(found, scores, poses, self.total, self.rank,
                                              self.world_size, device=torch_device)
        self.best = sh.unpack_best_key(sh.all_reduce_best(
            sh.pack_best_key(found, scores, self.begin), device=torch_device))

    def describe(self, stats, found):
        a = self.args
        if self.sharded:
            what = (f"C3: 2D ConstraintBuilder loop-closure batch, one {self.n_points}-point scan vs "
                    f"{self.total} distinct {a.grid}x{a.grid} submaps (seeds 42..{41 + self.total}, "
                    f"true positive #{self.positive}) submap-sharded over {self.world_size} GPU(s), "
                    f"{self.end - self.begin} per GPU; MatchFullSubmap depth {a.depth}, min_score "
                    f"{a.min_score}; per step one all-gather of the per-submap results + one "
                    f"all-reduce(max) of the best-match key")
        else:
            what = (f"C2: 2D FastCorrelativeScanMatcher MatchFullSubmap (branch and bound): "
                    f"{self.n_points}-point scan vs {self.total} {a.grid}x{a.grid} submap(s) per GPU, "
                    f"depth {a.depth}, full-angle search, min_score {a.min_score}")
        out = {"workload": what, "submaps_per_gpu": self.end - self.begin,
               "rotations": stats["num_scans"] // max(self.matches_per_step, 1),
               "nodes_expanded_per_step": stats["nodes_expanded"], "found": int(np.sum(found))}
        if self.sharded and self.gathered is not None:
            out["constraints_found_node_wide"] = int(np.sum(self.gathered[0]))
            out["best_match"] = {"score": self.best[0], "submap": self.best[1]}
        return out

    def roofline(self, acc, steps, pmc):
        """Front-end kernel (prep + lowest-resolution scoring of every rotation).
        Algorithmic bytes: N x 1 B per lowest-resolution candidate + 4 B per point
        per rotation.  The bound is not HBM: the 256 KB phase-plane set is L2-resident and every
        point gathers ONE 64-byte plane through L2 -> L1; `frac` is that gather traffic against
        the L2 peak.  The HBM-side ratios SURVEY 8d asks for are reported next to it."""
        launches = steps
        coarse = acc["coarse_candidates"] / launches
        scans = acc["num_scans"] / launches
        k_ms = acc["dominant_kernel_ms"] / launches
        alg = coarse * self.n_points * 1.0 + scans * self.n_points * 4.0
        gathered = scans * self.n_points * 64.0         # <= one plane per point per rotation
        secs = max(k_ms, 1e-9) * 1e-3
        # PMC passes exist for the single-submap command and for 16 submaps per GPU
        per_gpu = self.end - self.begin
        pmc_config = "" if per_gpu == 1 else ("c3" if per_gpu == 16 else None)
        traffic = None if pmc_config is None else pmc("PrepScoreFused", pmc_config)
        out = {
            "kernel": "PrepScoreFusedKernel (prep + bucketing + lowest-resolution scoring)",
            "bound": "l2-gather", "achieved": gathered / secs / 1e9, "peak": L2_PEAK_GBS,
            "unit": "GB/s", "frac": gathered / secs / 1e9 / L2_PEAK_GBS, "traffic": traffic,
            "kernel_ms": k_ms, "algorithmic_bytes": alg, "gathered_bytes": gathered,
            "algorithmic_GBps": alg / secs / 1e9,
            "hbm_frac_algorithmic": alg / secs / 1e9 / HBM_PEAK_GBS,
            "hbm_frac_traffic": None if traffic is None else traffic / secs / 1e9 / HBM_PEAK_GBS,
            "kernel_share_of_step_device_time": acc["dominant_kernel_ms"] / max(acc["device_ms"], 1e-9),
            # the same kernel against the measured rate of ITS OWN access pattern (four random
            # 64-byte rows of a 256 KB table per wave-wide gather: tools/row_gather_ceiling.hip,
            # profiles/r04_row_gather_ceiling.txt: 10.83 cycles per instruction and CU)
            "gather_instructions": scans * self.n_points / 4.0,
            "frac_of_gather_pattern_rate": (scans * self.n_points / 4.0) / secs /
                                           (256 * 2.4e9 / 10.83),
            "note": "frac = bytes the kernel gathers through L2 (one 64-byte phase plane per point "
                    "per rotation, an upper bound: points no candidate can reach are skipped) / "
                    "kernel time / L2 peak 34.5 TB/s.  hbm_frac_algorithmic may exceed 1: the "
                    "working set is on-chip (SURVEY 8d); hbm_frac_traffic = rocprofv3 FETCH_SIZE x2 "
                    "+ WRITE_SIZE of the same command / kernel time / 8 TB/s (null without the PMC "
                    "passes under --pmc-dir).  kernel_ms: HIP events on the kernel's own stream.",
        }
        # The tree search behind the front end.  Single searches (fewer than four problems per
        # call): ONE launch since round 6, TreeQueueKernel -- chains of best children per wavefront
        # over a work queue; batches: the level-synchronous ExpandWaveKernel launches.  Priced
        # against the chip's measured gather-issue ceiling, with SURVEY 8d's algorithmic bytes
        # (N x 1 B per candidate scored below the lowest resolution) next to it.
        # (round 6: batches of 32 and more problems run as independent single searches over the
        # library's host pool -- the work queue again; their kernel times are SUMS over searches
        # that run concurrently)
        queue = per_gpu < 4 or per_gpu >= 32
        tree = expansion_roofline(
            acc, steps,
            "TreeQueueKernel (branch and bound through a work queue: one quad gather per point and "
            "expansion, chains of best children per wavefront)" if queue else
            "ExpandWaveKernel (one wavefront per node: one quad gather per point)",
            None if pmc_config is None else pmc("TreeQueue" if queue else "ExpandWave", pmc_config),
            "achieved = lookups of the tree search (64 per gather instruction issued, counted by "
            "the kernel: an expansion stops once no child can reach the bound) / HIP-event time of "
            "the launch(es) / the measured gather-issue ceiling of the chip (coherent byte gathers; "
            "a quad gather touches up to 64 distinct 128-byte lines, so the fraction is a lower "
            "bound of how close the stage is to what such gathers allow).  algorithmic_bytes: "
            "SURVEY 8d's N x 1 B per candidate scored below the lowest resolution"
            + (" (the launch also takes and hands on nodes, selects the best leaf and publishes "
               "the results)" if queue else " (all levels; the wave launches score the top ones)"),
            algorithmic=(acc["candidates_scored"] - acc["coarse_candidates"]) / steps * self.n_points)
        if per_gpu >= 32:
            out["kernel_ms_is_sum_of_concurrent_launches"] = True
            out["gathered_bytes"] = gathered / 3.0      # (group bounds: one sum per three rotations)
            if tree is not None:
                tree["kernel_ms_is_sum_of_concurrent_launches"] = True
        if tree is not None and acc["expansion_ms"] > acc["dominant_kernel_ms"]:
            tree["front_end"] = out      # the tree search is the dominant kernel
            return tree
        if tree is not None:
            out["expansion"] = tree
        return out


class Fast2DConcurrentWorkload(Fast2DWorkload):
    """C2 from `threads` host threads, one search each per step, over `scans` different scans
    (poses of the same world).  scans = 1 is the `c2_easy` leg: the ONE scan rounds 1 - 5 timed
    as the headline, which happens to be an easy search (its dive finds a tight bound at once);
    the headline itself cycles through `--scans` = 8 since round 6."""

    def __init__(self, args, device, threads=8, scans=None, per_thread=16):
        sub = argparse.Namespace(**vars(args))
        sub.scans, sub.submaps = scans or threads, 0
        super().__init__(sub, device, 0, 1, sharded=False)
        from concurrent.futures import ThreadPoolExecutor
        self.threads, self.per_thread = threads, per_thread
        self.pool = ThreadPoolExecutor(threads)
        self.matches_per_step = threads * per_thread

    def search(self, k=0):
        """A step: every thread issues `per_thread` searches back to back (thread t: scans t,
        t + 1, ...), as the headline's passes are issued -- one search per thread and step would
        time the barrier between the steps."""
        def worker(t):
            return [Fast2DWorkload.search(self, t + j) for j in range(self.per_thread)]
        per = list(self.pool.map(worker, range(self.threads)))
        self.last_results = [r[0] for r in per]
        flat = [r for rs in per for r in rs]
        found = np.concatenate([r[0] for r in flat])
        scores = np.concatenate([r[1] for r in flat])
        stats = dict(flat[0][3])
        for r in flat[1:]:
            for key, v in r[3].items():
                stats[key] = stats.get(key, 0) + v
        return found, scores, flat[-1][2], stats

    def parity(self, result):
        """Every one of the scans searched by the reference (one host thread each) against the
        device result of the same scan."""
        from oracle import pyoracle as orc
        kind = _reference_kind()
        cls = (orc.ReferenceFastCorrelativeScanMatcher2D if kind == "reference"
               else orc.FastCorrelativeScanMatcher2D)
        a, lim = self.args, self.lim0
        matcher = cls(self.cells0, lim["resolution"], lim["max_x"], lim["max_y"], a.depth)
        refs = list(self.pool.map(lambda sc: matcher.match_full_submap(sc, a.min_score),
                                  self.host_scans[:self.threads]))
        return parity_record(kind, [(d[0][0], d[1][0], d[2][0], r["found"], r["score"], r["pose"])
                                    for d, r in zip(self.last_results, refs)])

    def describe(self, stats, found):
        out = super().describe(stats, found)
        out["workload"] = (f"C2 over {len(self.clouds)} scan(s): {self.matches_per_step} searches per "
                           f"step, issued from {self.threads} host threads; " + out["workload"])
        return out


def c1_scan(world, pose, beams, max_range, seed):
    """A scan of EXACTLY `beams` points within `max_range` (C1: 1000 beams, r_max < 5 m, hence
    the 27 rotations x 13 x 13 = 4563 candidates BASELINE.md prices): the lidar is given as many
    more beams per revolution as its returns beyond max_range take away, and `beams` of the
    returns, evenly spread in angular order, are kept.  (Rounds 1 - 5 asked for `beams` beams and
    matched the ~850 that came back.)"""
    asked = beams
    for _ in range(8):
        pts = world.scan(pose, asked, max_range, 0.01, seed)
        if len(pts) >= beams:
            break
        asked = int(math.ceil(asked * beams / max(len(pts), 1) * 1.02)) + 8
    if len(pts) <= beams:
        return pts
    keep = np.unique(np.floor(np.arange(beams) * (len(pts) / beams)).astype(np.int64))
    return np.ascontiguousarray(pts[keep])


class Rt2DWorkload:
    """C1 as a throughput workload: `--matches` independent real-time matches per step (scan i
    against resident grid i around pose i: one per trajectory / robot), 1000 beams vs 200x200,
    window 0.3 m / 7 deg, weights 0.1 / 0.1.  `grid`: side of the grids (the reference's active
    grid doubles 100 -> 200 -> 400, mapping/2d/grid_2d.cc:130-164); `dirty`: every grid has one
    more (small) scan inserted before every step, as the real caller does after every match
    (2d/local_trajectory_builder_2d.cc:78-80, :289) -- the match then never meets a cached
    image of its grid; the insertions are timed apart and not part of the step."""

    def __init__(self, args, device, matches=None, grid=200, dirty=False, distinct=None):
        from cartographer_amd import grid_2d, scan_matching as sm, synth
        self.sm = sm
        self.m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1,
                                                     device=device)
        batch = matches or args.matches
        # DISTINCT (grid, scan, initial pose) triples: one world per match up to `distinct`
        # (default: every match of a batch up to 128 its own, 256 for larger batches -- 256 grids
        # with their derived images are ~45 MB, far beyond what the L2s hold), reused round-robin
        # beyond that.  Rounds 2 - 5 cycled through 8 triples (every grid image L2-hot).
        distinct = distinct or getattr(args, "c1_distinct", 0) or (batch if batch <= 128 else 256)
        distinct = min(batch, distinct)
        grids, inits, scans = [], [], []
        self.host_cells, self.host_lims = [], []
        self.grid_side, self.dirty, self.insert_s = grid, dirty, 0.0
        self.dirty_inputs = []
        for k in range(distinct):
            cells, lim, world = synth.make_submap(42 + k, grid, grid, 0.05, 30, 1000, 5.0, 0.01)
            pose = world.free_pose(1234, 0.5)
            grids.append(grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), grid,
                                                         grid, cells=cells))
            scans.append(c1_scan(world, pose, args.beams, 5.0, 7))
            inits.append(sm.Rigid2d(pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)))
            self.host_cells.append(cells)
            self.host_lims.append(lim)
            c, s_ = math.cos(pose[2]), math.sin(pose[2])
            pts = np.zeros((8, 3), np.float32)              # eight returns: enough to bump the version
            pts[:, 0] = pose[0] + c * scans[-1][:8, 0] - s_ * scans[-1][:8, 1]
            pts[:, 1] = pose[1] + s_ * scans[-1][:8, 0] + c * scans[-1][:8, 1]
            self.dirty_inputs.append((np.asarray(pose[:2], np.float32), pts))
        self.distinct_grids = grids
        self.distinct = distinct
        self.candidates_per_match = 27 * 13 * 13      # re-read from the first search's stats
        self.G = [grids[i % len(grids)] for i in range(batch)]
        self.I = [inits[i % len(grids)] for i in range(batch)]
        self.S = [scans[i % len(grids)] for i in range(batch)]
        # Argument arrays built once, as a C++ caller holds them (the python marshalling of
        # 128 poses and pointers per call cost more than the device work).
        self.batch = sm.Rt2DBatch(self.m, self.G, self.S, resident=True)   # scans uploaded once
        self.init = np.array([[p.x, p.y, p.theta] for p in self.I], np.float64)
        self.points = float(np.mean([len(s) for s in self.S]))
        self.matches_per_step = batch
        self.n_points = int(self.points)

    def search(self, k=0):
        if self.dirty:
            t0 = time.perf_counter()
            for g, (origin, pts) in zip(self.distinct_grids, self.dirty_inputs):
                g.insert(origin, pts)
            self.insert_s += time.perf_counter() - t0
        scores, poses, stats = self.batch.match(self.init)
        self.candidates_per_match = stats["candidates_scored"] // self.matches_per_step
        return np.ones(len(scores), np.int32), scores, poses, stats

    def parity(self, result):
        """EVERY distinct (grid, scan, initial pose) of the batch matched by the reference's
        real_time_correlative_scan_matcher_2d.cc (a few ms each, on the host's threads)."""
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle as orc
        kind = _reference_kind()
        fn = orc.ref_rt2d_match if kind == "reference" else orc.rt2d_match
        if self.dirty:      # the device grids have had scans inserted: read them back
            grids = [(g.cells, g.limits) for g in self.distinct_grids]
        else:
            grids = list(zip(self.host_cells, self.host_lims))
        scores, poses = result[1], result[2]
        count = min(self.matches_per_step, len(grids))

        def one(i):
            cells, lim = grids[i]
            init = self.I[i]
            return fn(cells, lim["resolution"], lim["max_x"], lim["max_y"],
                      [init.x, init.y, init.theta], self.S[i], 0.3, math.radians(7.0), 0.1, 0.1)
        with ThreadPoolExecutor(_cores()) as pool:
            refs = list(pool.map(one, range(count)))
        pairs = [(True, scores[i], np.asarray(poses[i], np.float64).reshape(-1)[:3], True,
                  r["score"], r["pose"]) for i, r in enumerate(refs)]
        return parity_record(kind, pairs)

    def describe(self, stats, found):
        out = {"workload": f"C1: 2D RealTimeCorrelativeScanMatcher, {self.matches_per_step} "
                           f"independent matches per step ({self.distinct} distinct grid / scan / pose "
                           f"triples), {self.n_points}-point scans vs "
                           f"{self.grid_side}x{self.grid_side} resident probability grids"
                           f"{' (a scan inserted into every grid before every step)' if self.dirty else ''}"
                           f", window 0.3 m / 7 deg "
                           f"({stats['candidates_scored'] // self.matches_per_step} candidates per match)",
               "matches_per_step": self.matches_per_step,
               "search_space_candidates_per_match": stats["candidates_scored"] / self.matches_per_step,
               "summed_candidates_per_match": stats.get("coarse_candidates", 0) / self.matches_per_step,
               "refined_candidates_per_match": stats.get("refined_candidates", 0) / self.matches_per_step,
               "f32_finalists_per_match": stats.get("finalists", 0) / self.matches_per_step}
        return out

    def roofline(self, acc, steps, pmc):
        """Tile kernel (round 4): a tile of the quantised grid image sits in LDS; a HALF-wavefront
        is one stream of points, a lane fetches aligned 4-cell blocks (8 B) of two window rows per
        point.  C1: 13 x 13 window -> 8 row slots x 4 blocks x 2 rows = 512 B of ds_read_b64 per
        (rotation, point) serving 169 candidates.  Algorithmic bytes: 2 B per
        candidate per point.  kernel_ms: HIP events around the tile kernel(s) of the call (the
        parts of a batch run on streams of their own: their spans are added)."""
        k_ms = acc["dominant_kernel_ms"] / steps
        cand = acc["candidates_scored"] / steps
        secs = max(k_ms, 1e-9) * 1e-3
        alg = cand * self.points * 2.0
        scans = acc["num_scans"] / steps
        side = int(round(math.sqrt(cand / max(scans, 1))))
        summed = acc["coarse_candidates"] / steps
        if summed < cand:
            # From 192 matches per call on: block bounds first (rt_2d_bounds.h).  Round 6: blocks of
            # 4 x 4 translations -- per (rotation, point) the bound kernel reads (side + 3) / 4
            # block rows of two aligned dwords each from the sixteen phase planes of the 4 x 4
            # max-pooled byte image in LDS, and the point's two coordinates; a second kernel
            # (Rt2DBoundTail4Kernel) sums the candidates of the few blocks that reach the bound out
            # of a byte image of the match's box in LDS and finishes the match.  kernel_ms is the
            # BOUND kernel alone (HIP events around it; the tail kernel is in device_ms).
            nb = (side + 3) // 4
            lds = scans * self.points * (8.0 * nb + 8.0)
            return {"kernel": "Rt2DBoundKernel<NB4, 2> (4x4 block bounds from sixteen pooled byte planes in LDS)",
                    "bound": "lds", "achieved": lds / secs / 1e9, "peak": LDS_B32_PEAK_GBS, "unit": "GB/s",
                    "frac": lds / secs / 1e9 / LDS_B32_PEAK_GBS,
                    "traffic": (pmc("Rt2DBoundKernel", "c1" if self.matches_per_step == 128 else "c1b1024")
                                if self.matches_per_step in (128, 1024) and self.grid_side == 200
                                and not self.dirty else None),
                    "kernel_ms": k_ms, "algorithmic_bytes": alg, "lds_bytes": lds,
                    # (calls of 256 matches and more go out in three PARTS on streams of their own:
                    # kernel_ms is the SUM of their launches' durations, which overlap)
                    "kernel_ms_is_sum_of_concurrent_parts": self.matches_per_step >= 256,
                    "algorithmic_GBps": alg / secs / 1e9,
                    "hbm_frac_algorithmic": alg / secs / 1e9 / HBM_PEAK_GBS,
                    "candidates_per_s_kernel": cand / secs,
                    "summed_candidates_per_s_kernel": summed / secs,
                    "note": "block bounds: frac = LDS bytes of the bound kernel's row reads / its "
                            "HIP-event time (staging of the planes and the cloud included) / 75 TB/s "
                            "(the LDS peak of dword reads); the kernel is co-limited by its vector "
                            "instructions (37 k wave-instructions per match, 0.73 of the VALU issue "
                            "rate: profiles/r06b_c1dev_*) and by bank conflicts of its random dword "
                            "pairs; algorithmic bytes stay SURVEY 8d's 2 B per candidate of the "
                            "SEARCH SPACE per point (what the reference reads), of which the device "
                            "reads a fraction: summed_candidates = block bounds + candidates of "
                            "surviving blocks"}
        lds = cand / (side * side) * self.points * 512.0
        return {"kernel": "Rt2DTileKernel (LDS tiles of the quantised grid image, packed 16-bit sums)",
                "bound": "lds", "achieved": lds / secs / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                "frac": lds / secs / 1e9 / LDS_PEAK_GBS,
                "traffic": pmc("Rt2DTile", "c1") if self.matches_per_step == 128 and
                self.grid_side == 200 and not self.dirty else None,
                "kernel_ms": k_ms, "algorithmic_bytes": alg, "lds_bytes": lds,
                "algorithmic_GBps": alg / secs / 1e9,
                "hbm_frac_algorithmic": alg / secs / 1e9 / HBM_PEAK_GBS,
                "candidates_per_s_kernel": cand / secs,
                "note": "frac = LDS bytes read by the tile kernel's window reads / its HIP-event "
                        "time (image staging, list copies and task building of every work item "
                        "included; discretisation is the prep kernel's) / 150 TB/s (ds_read_b64 "
                        "aggregate); hbm_frac_algorithmic > 1 means on-chip residency"}


class Rt2DTsdfWorkload:
    """C1 on a TSDF2D (SURVEY 8 a8', ComputeCandidateScore(TSDF2D),
    real_time_correlative_scan_matcher_2d.cc:38-59): the same world, scan and window as C1, the
    grid a truncated signed distance field of it (tsd and weight planes through the reference's
    TSDValueConverter expressions, tsd_value_converter.h:39-67).  A score is a ratio of two f32
    sums, not an integer sum: this branch runs on the one-thread-per-candidate kernels
    (rt_2d.hip), every candidate with the reference's sequential f32 chains -- the leg exists so
    that the slow path has a number."""
    TRUNCATION, MAX_WEIGHT = 0.3, 10.0

    def __init__(self, args, device):
        from scipy import ndimage
        from cartographer_amd import scan_matching as sm, synth
        self.sm = sm
        self.m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1,
                                                     device=device)
        cells, lim, world = synth.make_submap(42, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
        pose = world.free_pose(1234, 0.5)
        self.scan = c1_scan(world, pose, args.beams, 5.0, 7)
        self.init = sm.Rigid2d(pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0))
        # walls = cells whose correspondence cost lies in the lower half; distance to the nearest
        value = cells & 32767
        wall = (value > 0) & (value < 16384)
        d = ndimage.distance_transform_edt(~wall) * 0.05
        rng = np.random.default_rng(42)
        weight = rng.uniform(1.0, self.MAX_WEIGHT, cells.shape)
        known = d < self.TRUNCATION

        def to_value(x, lo, hi):          # BoundedFloatToValue (probability_values.h:32-44)
            return (np.round((np.clip(x, lo, hi) - lo) * (32766.0 / (hi - lo))).astype(np.int64) + 1)
        self.tsd = np.where(known, to_value(d, -self.TRUNCATION, self.TRUNCATION), 0).astype(np.uint16)
        self.wgt = np.where(known, to_value(weight, 0.0, self.MAX_WEIGHT), 0).astype(np.uint16)
        self.lim = lim
        self.grid = sm.TSDF2D(self.tsd, self.wgt, 0.05, lim["max_x"], lim["max_y"], self.TRUNCATION,
                              self.MAX_WEIGHT)
        self.matches_per_step = 1
        self.n_points = len(self.scan)

    def search(self, k=0):
        score, est = self.m.match(self.init, self.scan, self.grid)
        return (np.ones(1, np.int32), np.array([score], np.float64),
                np.array([[est.x, est.y, est.theta]], np.float64), self.m.last_stats)

    def parity(self, result):
        from oracle import pyoracle as orc
        kind = _reference_kind()
        lim, i = self.lim, self.init
        if kind == "reference":
            r = orc.ref_rt2d_match(self.tsd, lim["resolution"], lim["max_x"], lim["max_y"],
                                   [i.x, i.y, i.theta], self.scan, 0.3, math.radians(7.0), 0.1, 0.1,
                                   weight_cells=self.wgt, truncation_distance=self.TRUNCATION,
                                   max_weight=self.MAX_WEIGHT)
        else:
            r = orc.rt2d_match_tsdf(self.tsd, self.wgt, lim["resolution"], lim["max_x"], lim["max_y"],
                                    self.TRUNCATION, self.MAX_WEIGHT, [i.x, i.y, i.theta], self.scan,
                                    0.3, math.radians(7.0), 0.1, 0.1)
        return parity_record(kind, [(True, result[1][0], result[2][0], True, r["score"], r["pose"])])

    def describe(self, stats, found):
        return {"workload": f"C1 on a TSDF2D: 2D RealTimeCorrelativeScanMatcher, one match per step, "
                            f"{self.n_points}-point scan vs a 200x200 TSDF (tsd + weight planes), "
                            f"window 0.3 m / 7 deg ({stats['candidates_scored']} candidates)",
                "matches_per_step": 1}

    def roofline(self, acc, steps, pmc):
        """Per-candidate kernels: two 2-byte gathers (tsd, weight) per candidate and point from the
        80 KB + 80 KB planes (L2 / L1 resident): algorithmic bytes 4 B per candidate and point,
        priced against the chip's gather-issue ceiling like the other gather kernels."""
        k_ms = acc["dominant_kernel_ms"] / steps
        cand = acc["candidates_scored"] / steps
        secs = max(k_ms, 1e-9) * 1e-3
        lookups = cand * self.n_points * 2.0
        return {"kernel": "Rt2DScoreKernel<tsdf> (one thread per candidate, sequential f32 chains)",
                "bound": "gather-issue", "achieved": lookups / secs / 1e9, "peak": GATHER_PEAK_GLOOKUPS,
                "unit": "Glookup/s", "frac": lookups / secs / 1e9 / GATHER_PEAK_GLOOKUPS,
                "traffic": None, "kernel_ms": k_ms, "algorithmic_bytes": cand * self.n_points * 4.0,
                "algorithmic_GBps": cand * self.n_points * 4.0 / secs / 1e9,
                "hbm_frac_algorithmic": cand * self.n_points * 4.0 / secs / 1e9 / HBM_PEAK_GBS,
                "note": "frac = lookups / kernel time / the builder-measured gather-issue ceiling "
                        "(tools/gather_ceiling.hip); SURVEY 8d's ratio next to it: "
                        "hbm_frac_algorithmic = 4 B per candidate and point / kernel time / 8 TB/s "
                        "(the planes are on-chip)"}


class Rt2DPipelinedWorkload(Rt2DWorkload):
    """C1 as the throughput a fleet sees: `threads` host threads (one per group of trajectories)
    each issue `calls` batches of `matches` resident matches per step, every thread on its own
    argument arrays and, inside the library, its own workspaces and streams -- how C2's headline
    is issued, and what hides the host's share of a call (plan, descriptor fill, launch, wait)
    behind the device work of the other threads' calls.  Same entry point, same results."""

    def __init__(self, args, device, matches=128, threads=4, calls=4):
        super().__init__(args, device, matches=matches)
        from concurrent.futures import ThreadPoolExecutor
        clouds = self.batch._clouds                          # uploaded once, shared
        self.batches = [self.batch] + [self.sm.Rt2DBatch(self.m, self.G, clouds, resident=True)
                                       for _ in range(threads - 1)]
        self.threads, self.calls = threads, calls
        self.batch_matches = matches
        self.matches_per_step = matches * threads * calls
        self.pool = ThreadPoolExecutor(threads)

    def _worker(self, batch):
        total = None
        for _ in range(self.calls):
            scores, poses, stats = batch.match(self.init)
            if total is None:
                total = dict(stats)
            else:
                for k, v in stats.items():
                    total[k] = total.get(k, 0) + v
        return scores.copy(), poses.copy(), total

    def search(self, k=0):
        results = list(self.pool.map(self._worker, self.batches))
        stats = dict(results[0][2])
        for r in results[1:]:
            for key, v in r[2].items():
                stats[key] = stats.get(key, 0) + v
        self.candidates_per_match = stats["candidates_scored"] // self.matches_per_step
        scores, poses = results[0][0], results[0][1]
        return np.ones(len(scores), np.int32), scores, poses, stats

    def describe(self, stats, found):
        d = super().describe(stats, found)
        d["workload"] = (f"C1, pipelined ({self.threads} host threads x {self.calls} calls of "
                         f"{self.batch_matches} resident matches per step): " + d["workload"])
        d["host_threads"] = self.threads
        d["calls_per_thread_per_step"] = self.calls
        d["matches_per_call"] = self.batch_matches
        return d

    def roofline(self, acc, steps, pmc):
        r = super().roofline(acc, steps, pmc)
        r["traffic"] = None
        r["note"] += ("  Pipelined leg: kernel_ms is the SUM of the bulk kernels' durations per "
                      "step (HIP events on each call's own stream); launches of different calls "
                      "share the chip, so the per-launch figure of the single-call leg is the one "
                      "to read as a kernel roofline.")
        return r


class Rt3DWorkload:
    """C4: 64 rings x 1024 azimuths vs a 150^3 HybridGrid, window 0.5 m / 2 deg."""

    def __init__(self, args, device):
        from cartographer_amd import scan_matching_3d as sm3, synth
        grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
        self.vox = grid.voxels()
        pos = world.free_position(77, 0.5)
        self.cloud = world.scan(pos, 0.3, 64, 1024, seed=9)
        c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
        self.init = sm3.Rigid3d(tuple(pos + np.array([0.07, -0.04, 0.02])), (c, 0.0, 0.0, s))
        self.m = sm3.RealTimeCorrelativeScanMatcher3D(0.5, math.radians(2.0), 0.1, 0.1)
        self.matches_per_step = 1
        self.n_points = len(self.cloud)

    def search(self, k=0):
        score, est = self.m.match(self.init, self.cloud, 0.1, self.vox)
        return np.ones(1, np.int32), np.array([score], np.float32), [est], self.m.last_stats

    def parity(self, result):
        """The reference's real_time_correlative_scan_matcher_3d.cc on this workload is ten
        minutes of eight host cores: its committed output (tests/golden/rt3d_c4_reference.json,
        made by tests/golden/make_rt3d_c4_golden.py from the same seeded inputs,
        tests/golden/workloads.py rt3d_c4)."""
        path = os.path.join(ROOT, "tests", "golden", "rt3d_c4_reference.json")
        with open(path) as f:
            g = json.load(f)["rt3d_c4"]
        assert g["num_points"] == self.n_points, "the golden file is of another cloud"
        est = result[2][0]
        pose = list(est.translation) + list(est.rotation)
        return parity_record("reference (committed output, tests/golden/rt3d_c4_reference.json)",
                             [(True, result[1][0], pose, True, g["score"], g["pose"])])

    def describe(self, stats, found):
        return {"workload": f"C4: 3D RealTimeCorrelativeScanMatcher, {self.n_points}-point cloud vs "
                            f"150^3 HybridGrid ({len(self.vox)} voxels), window 0.5 m / 2 deg",
                "search_space_lookups_per_step": stats["candidates_scored"] * self.n_points,
                "bounds_evaluated_per_step": stats["coarse_candidates"],
                "finalists_rescored_exactly": stats["nodes_expanded"],
                "note": "candidates = the reference's exhaustive search space (every one of them is "
                        "either scored or excluded by an upper bound of its 2x2x2 block of "
                        "translations); bounds_evaluated = group bounds + candidates scored one by one"}

    def roofline(self, acc, steps, pmc):
        """Group pass (the dominant kernel): one byte lookup per (rotation, block of translations,
        point).  Tiled path (default): the lookups are ds_read_u8 gathers from an LDS tile of the
        dilated brick, 2 LDS cycles per conflict-free wave-instruction (MI355X_MICROARCH.md, LDS
        table) = 256 CUs x 2.4 GHz / 2 x 64 lanes = 1.97e13 lookups/s; the r03_c4 SQ counters
        (profiles/) show where the rest goes: LDS array busy 47 % of the kernel (40 % of that
        bank conflicts), 10.6 vector instructions per wave-lookup of which 6.5 are the lookup
        itself, waves in s_waitcnt half of their time (five barriers per point chunk) -- counters
        taken before the lookup loop was hand-scheduled and the workgroups halved (DESIGN 5.4:
        6.6 per lookup in the loop, two workgroups per CU).  Gather
        path (debug switch rt3d_no_tiles): buffer_load_ubyte from L2, measured ceiling 19 cycles per
        64-lane gather per CU = 2.05e12 lookups/s (profiles/r02_rt3d_gather_ceiling.txt)."""
        k_ms = acc["dominant_kernel_ms"] / steps
        cand = acc["candidates_scored"] / steps
        scans = acc["num_scans"] / steps
        secs = max(k_ms, 1e-9) * 1e-3
        side = round((cand / max(scans, 1)) ** (1.0 / 3.0))
        groups = ((side + 1) // 2) ** 3
        bulk = acc["coarse_candidates"] / steps != cand       # the bounds path ran
        tiles = bulk and groups <= 1024
        lookups = scans * (groups if bulk else cand / max(scans, 1)) * self.n_points
        dense_group_lookups = lookups
        # Round 4: the rotation-block level.  The library reports the bounds above the candidates
        # it evaluated (rotation blocks + the (rotation, group) pairs they leave), their lookups
        # and the time of all their passes: that is what `achieved` is measured on.
        if bulk and acc.get("expansion_lookups", 0) > 0 and acc.get("expansion_ms", 0) > 0:
            lookups = acc["expansion_lookups"] / steps
            k_ms = acc["expansion_ms"] / steps
            secs = max(k_ms, 1e-9) * 1e-3
        alg = cand * self.n_points * 2.0 + scans * self.n_points * 12.0
        peak = 256 * 2.4e9 / 2 * 64 / 1e9 if tiles else 2050.0              # G lookups/s
        kernel = ("Rt3DTileKernel<groups> (upper bounds of 2x2x2 rotations x 2x2x2 translations, then of the "
                  "2x2x2 blocks of translations of the surviving pairs: fixed-point "
                  "cell arithmetic, byte gathers from LDS tiles of the dilated uint8 brick)" if tiles
                  else "Rt3DBulkKernel<groups> (upper bounds of 2x2x2 blocks of translations on the "
                       "dilated uint8 brick)" if bulk else "Rt3DScoreKernel")
        # (counter traffic of what `achieved` spans: the rotation-block launch + the two launches
        # over the surviving (rotation, group) pairs; older passes name one dense group pass)
        if tiles:
            blocks_t, pairs_t = pmc("Rt3DTileKernel<true, false>", "c4"), pmc("Rt3DTileKernel<true, true>", "c4")
            traffic = (blocks_t + 2.0 * pairs_t if blocks_t is not None and pairs_t is not None
                       else pmc("Rt3DTileKernel<true>", "c4"))
        else:
            traffic = pmc("Rt3DBulkKernel<true>", "c4")
        return {"kernel": kernel,
                "bound": "lds" if tiles else "gather-issue", "achieved": lookups / secs / 1e9,
                "peak": peak, "unit": "Glookup/s", "frac": lookups / secs / 1e9 / peak,
                "traffic": traffic,
                "kernel_ms": k_ms,
                "group_level_bounds": acc.get("expansion_nodes", 0) / steps,
                "dense_group_bounds": scans * groups,
                "lookups_avoided_by_rotation_blocks": 1.0 - lookups / max(dense_group_lookups, 1.0),
                "algorithmic_bytes": alg,
                "hbm_frac_algorithmic_whole_step":
                    alg / (acc["device_ms"] / steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel_share_of_step_device_time": k_ms / (acc["device_ms"] / steps),
                "lookups_per_s_kernel": lookups / secs,
                "note": "achieved = byte lookups the kernel performs / kernel time; peak = "
                        + ("the LDS rate of conflict-free ds_read_u8 gathers (2 cycles per "
                           "wave-instruction and CU: 1.97e13 lookups/s)" if tiles else
                           "the measured gather-issue ceiling of the chip (19 cycles per 64-lane "
                           "buffer_load_ubyte per CU, profiles/r02_rt3d_gather_ceiling.txt)")
                        + ".  hbm_frac_algorithmic_whole_step prices the reference's exhaustive "
                        "search (SURVEY 8d: 2 B per candidate-point + 12 B per rotation-point) "
                        "at the time of the whole step against 8 TB/s; the bricks (7-15 MB) "
                        "are L2/MALL-resident"}


class Fast3DWorkload:
    """C5, one GPU's share: hi 0.1 m / low 0.45 m, depth 8 / full-resolution depth 3,
    pose_graph.lua windows; `--submaps` pairs per step through cmx_fast3d_match_batch."""

    def __init__(self, args, device, pairs=None, rank=0, world_size=1, sharded=False):
        from cartographer_amd import scan_matching_3d as sm3, sharding, synth
        self.sm3, self.sharding = sm3, sharding
        self.rank, self.world_size, self.sharded = rank, world_size, sharded
        size = (15.0, 15.0, 7.5)
        grid, world = synth.make_submap_3d(42, 0.1, size, 8, 32, 512)
        low, _ = synth.make_submap_3d(42, 0.45, size, 8, 32, 512)
        vox, low_vox = grid.voxels(), low.voxels()
        rng = np.random.default_rng(1)
        hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
        hist[10:14] += 6.0
        pos = world.free_position(77, 0.6)
        yaw = 0.4
        full = world.scan(pos, yaw, 32, 512, seed=1)
        self.hi = full[::6].copy()
        self.lo = full[::80].copy()
        scan_hist = np.roll(hist, -19).copy()
        opt = dict(branch_and_bound_depth=8, full_resolution_depth=3, min_rotational_score=0.77,
                   min_low_resolution_score=0.35, linear_xy_search_window=5.0,
                   linear_z_search_window=1.0, angular_search_window=math.radians(15.0))
        self.vox, self.low_vox, self.hist, self.opt = vox, low_vox, hist, opt   # (cpu baseline)
        self.scan_hist = scan_hist
        self.node = sm3.Rigid3d((pos[0] + 0.8, pos[1] - 0.6, pos[2] + 0.2),
                                (math.cos((yaw + 0.1) / 2), 0.0, 0.0, math.sin((yaw + 0.1) / 2)))
        self.data = sm3.TrajectoryNodeData(self.hi, self.lo, scan_hist)
        self.pairs = pairs or args.submaps or (C5_SUBMAPS_PER_GPU if sharded else 1)
        self.total = self.pairs * world_size
        self.begin = rank * self.pairs                 # equal blocks: shard_range of `total`
        self.matches_per_step = self.pairs
        self.gathered = self.best = None
        self.n_points = len(self.hi)
        # A batch is one node against DISTINCT submaps (C5: 256 over 8 GPUs = 32 per GPU), each
        # with its own grids and stack in HBM (~110 MB); the node was recorded in submap #0's
        # world, the others are the negatives a loop-closure search mostly meets.  They share
        # the histogram so that every pair passes the yaw pre-filter and is searched in full.
        # (sharded over N GPUs: rank r owns submaps r * pairs ... ; the node comes from #0's world)
        self.matchers = []
        self.host_grids = []           # (the parity gate's share of the block: voxel lists)
        keep = getattr(args, "parity_submaps", 4)
        if self.begin == 0:
            self.matchers.append(sm3.FastCorrelativeScanMatcher3D(0.1, vox, grid.grid_size, 0.45,
                                                                  low_vox, hist, **opt))
            self.host_grids.append((vox, low_vox))
        for gid in range(self.begin + len(self.matchers), self.begin + self.pairs):
            g, _ = synth.make_submap_3d(42 + gid, 0.1, size, 8, 32, 512)
            lw, _ = synth.make_submap_3d(42 + gid, 0.45, size, 8, 32, 512)
            gv, lv = g.voxels(), lw.voxels()
            self.matchers.append(sm3.FastCorrelativeScanMatcher3D(
                0.1, gv, g.grid_size, 0.45, lv, hist, **opt))
            if len(self.host_grids) < keep:
                self.host_grids.append((gv, lv))

    def exchange(self, found, scores, results, torch_device):
        """Every submap's optional constraint to every rank (9 words per submap) + the node-wide
        best match (all-reduce(max) of the packed key), as in 2D."""
        sh = self.sharding
        rows = np.zeros((self.pairs, 9), np.float64)
        for i, r in enumerate(results):
            if r is not None:
                rows[i, 0] = 1.0
                rows[i, 1] = float(np.float32(r["score"]))
                rows[i, 2:5] = r["pose_estimate"].translation
                rows[i, 5:9] = r["pose_estimate"].rotation
        self.gathered = sh.all_gather_rows(rows, self.total, self.rank, self.world_size,
                                           device=torch_device)
        self.best = sh.unpack_best_key(sh.all_reduce_best(
            sh.pack_best_key(found, scores, self.begin), device=torch_device))

    def search(self, k=0):
        sm3 = self.sm3
        if self.pairs == 1 and not self.sharded:
            got = self.matchers[0].match(self.node, sm3.Rigid3d(), self.data, 0.2)
            stats = self.matchers[0].last_stats
            found = np.array([got is not None], np.int32)
            scores = np.array([got["score"] if got else 0.0], np.float32)
            return found, scores, [got], stats
        results, stats = sm3.fast3d_match_batch(self.matchers, [self.node] * self.pairs,
                                                [sm3.Rigid3d()] * self.pairs, [0] * self.pairs,
                                                [0.2] * self.pairs, self.data)
        found = np.array([r is not None for r in results], np.int32)
        scores = np.array([r["score"] if r else 0.0 for r in results], np.float32)
        return found, scores, results, stats

    def reference_matcher(self, k=0):
        """The reference's fast_correlative_scan_matcher_3d.cc over the grids of this rank's
        submap #k (or the oracle port when oracle/_ref did not travel); built once per submap: the
        parity gate and the CPU baseline leg (submap #0) share them."""
        cache = self.__dict__.setdefault("_references", {})
        if k not in cache:
            from oracle import pyoracle as orc
            kind = _reference_kind()
            cls = (orc.ReferenceFastCorrelativeScanMatcher3D if kind == "reference"
                   else orc.FastCorrelativeScanMatcher3D)
            o = self.opt
            vox, low_vox = self.host_grids[k]
            cache[k] = (kind, cls(
                0.1, vox, 0.45, low_vox, self.hist, o["branch_and_bound_depth"],
                o["full_resolution_depth"], o["min_rotational_score"],
                o["min_low_resolution_score"], o["linear_xy_search_window"],
                o["linear_z_search_window"], o["angular_search_window"]))
        return cache[k]

    def reference_match(self, k=0):
        node = list(self.node.translation) + list(self.node.rotation)
        return self.reference_matcher(k)[1].match(node, [0, 0, 0, 1, 0, 0, 0], [1, 0, 0, 0],
                                                  self.hi, self.lo, self.scan_hist, 0.2)

    def parity(self, result):
        """The first `--parity-submaps` submaps of this rank's block (default 4: 0.7 s of one host
        core each incl. the reference's own stack; `--parity-submaps 256` = every pair of config
        [4]) matched by the reference: found / score / pose against the device's result of the same
        pair."""
        from concurrent.futures import ThreadPoolExecutor
        count = len(self.host_grids)
        if count == 0:
            return {"vs": None, "checked": 0}
        with ThreadPoolExecutor(min(count, _cores())) as pool:
            refs = list(pool.map(self.reference_match, range(count)))
        pairs = []
        for k, ref in enumerate(refs):
            got = result[2][k]
            pose = None if got is None else (list(got["pose_estimate"].translation) +
                                             list(got["pose_estimate"].rotation))
            pairs.append((got is not None, 0.0 if got is None else got["score"], pose,
                          ref["found"], ref["score"], ref["pose"]))
        for k in range(1, count):       # (the CPU baseline leg keeps submap #0's matcher only)
            self._references.pop(k, None)
        return parity_record(_reference_kind(), pairs)

    def describe(self, stats, found):
        out = {"workload": f"C5: 3D FastCorrelativeScanMatcher Match, one node against {self.total} "
                           f"distinct submap(s) per step (seeds 42..{41 + self.total}, the node from "
                           f"#0's world)" + (f", submap-sharded over {self.world_size} GPU(s), "
                                             f"{self.pairs} per GPU" if self.sharded else "") +
                           f", 150^3 hi-res 0.1 m + low-res 0.45 m grids, depth 8 / "
                           f"full-resolution depth 3, {self.n_points} hi-res points",
               "pairs_per_step": self.pairs, "submaps_per_gpu": self.pairs,
               "found": int(np.sum(found)), "nodes_expanded_per_step": stats["nodes_expanded"]}
        if self.gathered is not None:
            out["constraints_found_node_wide"] = int(np.sum(self.gathered[:, 0]))
            out["best_match"] = {"score": self.best[0], "submap": self.best[1]}
        return out

    def roofline(self, acc, steps, pmc):
        k_ms = acc["dominant_kernel_ms"] / steps
        coarse = acc["coarse_candidates"] / steps
        scans = acc["num_scans"] / steps
        secs = max(k_ms, 1e-9) * 1e-3
        alg = coarse * self.n_points * 1.0 + scans * self.n_points * 12.0
        # (the counter passes are of the shipped configuration: none for the no-families A/B leg)
        pmc_config = "c5" if self.pairs == 32 and not getattr(self, "no_pmc", False) else None
        coarse_line = {"kernel": "ScoreCoarse3D", "bound": "hbm", "achieved": alg / secs / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg / secs / 1e9 / HBM_PEAK_GBS,
                       "traffic": None if pmc_config is None else pmc("ScoreCoarse3D", pmc_config),
                       "kernel_ms": k_ms, "algorithmic_bytes": alg,
                       "note": "lowest-resolution scoring: 1 B per candidate-point + 12 B per "
                               "yaw-point (SURVEY 8d) / kernel time / 8 TB/s"}
        expand = expansion_roofline(
            acc, steps, "Expand3DKernel (one workgroup per node: one 8-byte oct gather per point)",
            None if pmc_config is None else pmc("Expand3DKernel", pmc_config),
            "lookups = nodes taken off the frontiers x points (an upper bound: nodes found below "
            "the bound when they are taken are skipped); span = HIP events around the expansion "
            "launches")
        if expand is None:
            return coarse_line
        # A batch over distinct submaps is bound by the memory side: every lookup reads an
        # 8-byte oct word from a 128-byte line nobody else wants, plus the point's 16-byte cell
        # record (coalesced, cached).  frac prices the algorithmic bytes against HBM; the
        # counter traffic (L2 misses x 128 B) next to it says how much of the peak the line
        # granularity actually consumes.
        secs = expand["kernel_ms"] * 1e-3
        # 8 B per (node, point) lookup -- the oct word.  (Until round 3 this line priced
        # 24 B, counting the point's 16-byte cell record once per lookup; a family reads that
        # record once for up to eight sibling nodes, so the numerator overstated.  Kept as
        # frac_24B for comparison with the older records.)
        lookups_launch = expand["lookups_per_step"] / expand["launches_per_step"]
        alg_launch = lookups_launch * 8.0
        expand["gather_issue"] = {k: expand[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        expand.update({"bound": "hbm", "achieved": alg_launch / secs / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": alg_launch / secs / 1e9 / HBM_PEAK_GBS,
                       "frac_24B": lookups_launch * 24.0 / secs / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes": alg_launch,
                       "traffic_per_algorithmic_byte": None if expand["traffic"] is None
                       else expand["traffic"] / alg_launch,
                       "hbm_frac_traffic": None if expand["traffic"] is None
                       else expand["traffic"] / secs / 1e9 / HBM_PEAK_GBS})
        expand["note"] = ("achieved = algorithmic bytes per launch (8 B per lookup: the oct word, "
                          "SURVEY 8d) / average launch time / 8 TB/s; hbm_frac_traffic = rocprofv3 "
                          "FETCH_SIZE x2 + WRITE_SIZE of the same command per launch / the same "
                          "time / 8 TB/s (32 distinct submaps: every L2 miss a 128-byte line for 8 "
                          "bytes); gather_issue: the same lookups against the chip's gather-issue "
                          "ceiling.  " + expand["note"])
        expand["lowest_resolution_scoring"] = coarse_line
        return expand


def make_workload(name, args, device, rank, world_size):
    if name == "c2":
        return Fast2DWorkload(args, device, rank, world_size, sharded=False)
    if name == "c3":
        return Fast2DWorkload(args, device, rank, world_size, sharded=True)
    if name == "c1":
        return Rt2DWorkload(args, device)
    if name == "c4":
        return Rt3DWorkload(args, device)
    return Fast3DWorkload(args, device, rank=rank, world_size=world_size,
                          sharded=world_size > 1 or args.force_dist)


STAT_KEYS = ("candidates_scored", "coarse_candidates", "dominant_kernel_ms", "device_ms",
             "num_scans", "expansion_ms", "expansion_launches", "expansion_nodes",
             "expansion_lookups", "refined_candidates", "finalists")
MS_KEYS = ("dominant_kernel_ms", "device_ms", "expansion_ms")    # HIP-event times: only recorded
                                                                 # under cmx_debug_set("timing", 1)
GATHER_PEAK_GLOOKUPS = 2050.0   # measured: 19 cycles per 64-lane gather instruction per CU
                                # (profiles/r02_rt3d_gather_ceiling.txt) x 256 CUs x 2.4 GHz


def expansion_roofline(acc, steps, kernel, traffic, note, algorithmic=None):
    """The branch-and-bound expansion launches (fast 2D: TreeQueueKernel / ExpandWaveKernel, fast
    3D: Expand3DKernel), priced against the rate at which the chip issues wave-wide gathers --
    a builder-measured ceiling (tools/gather_ceiling.hip), so SURVEY 8d's algorithmic bytes over
    the HBM peak are printed next to it.  None when the call timed none."""
    launches = acc["expansion_launches"] / steps
    if launches <= 0 or acc["expansion_ms"] <= 0:
        return None
    span_ms = acc["expansion_ms"] / steps
    lookups = acc["expansion_lookups"] / steps
    secs = span_ms * 1e-3
    out = {"kernel": kernel, "bound": "gather-issue", "achieved": lookups / secs / 1e9,
           "peak": GATHER_PEAK_GLOOKUPS, "unit": "Glookup/s",
           "frac": lookups / secs / 1e9 / GATHER_PEAK_GLOOKUPS, "traffic": traffic,
           "launches_per_step": launches, "kernel_ms": span_ms / launches,
           "span_ms_per_step": span_ms, "nodes_per_step": acc["expansion_nodes"] / steps,
           "lookups_per_step": lookups,
           "kernel_share_of_step_device_time": acc["expansion_ms"] / max(acc["device_ms"], 1e-9),
           "note": note}
    if algorithmic is not None:
        out["algorithmic_bytes"] = algorithmic
        out["algorithmic_GBps"] = algorithmic / secs / 1e9
        out["hbm_frac_algorithmic"] = algorithmic / secs / 1e9 / HBM_PEAK_GBS
        out["hbm_frac_traffic"] = (None if traffic is None else
                                   traffic / secs / 1e9 / HBM_PEAK_GBS)
    return out


def set_timing(on):
    """The library's HIP-event brackets (cmx_match_stats *_ms).  OFF in every timed region -- a
    production caller does not ask for them and they cost a latency-bound call ~15 % of its wall
    time -- and ON for the untimed instrumented passes the kernel times come from."""
    from cartographer_amd import _lib
    _lib.debug_set(timing=1 if on else 0)


def instrumented(workload, acc, steps, passes=None):
    """A few untimed passes with the event brackets on; their *_ms sums, scaled to `steps`
    passes, replace the (zero) ones of the timed region in `acc`."""
    n = passes or max(3, min(steps, 20))
    inserted = getattr(workload, "insert_s", None)      # (dirty-grid leg: its own bookkeeping)
    set_timing(True)
    try:
        ms = {k: 0.0 for k in MS_KEYS}
        workload.search()
        for _ in range(n):
            r = workload.search()
            for k in MS_KEYS:
                ms[k] += r[3].get(k, 0)
    finally:
        set_timing(False)
        if inserted is not None:
            workload.insert_s = inserted
    for k in MS_KEYS:
        acc[k] = ms[k] * steps / n


def measure(workload, steps, warmup, sync):
    """steps timed passes of workload.search() without the library's event brackets, then the
    instrumented passes for the kernel times; returns (seconds, acc, last result)."""
    acc = {k: 0.0 for k in STAT_KEYS}
    last = None
    set_timing(False)
    for _ in range(warmup):
        workload.search()
    sync()
    if hasattr(workload, "insert_s"):
        workload.insert_s = 0.0     # (dirty-grid leg: only the insertions between the TIMED steps)
    gc.collect()
    gc.disable()        # a generation-2 pass of the interpreter (tens of ms with torch loaded)
    try:                # must not land in a timed region of a few steps
        t0 = time.perf_counter()
        for _ in range(steps):
            last = workload.search()
            for k in STAT_KEYS:
                acc[k] += last[3].get(k, 0)
        sync()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    instrumented(workload, acc, steps)
    sync()
    return dt, acc, last


def other_configs(args, device, sync, pmc):
    """C1 (single + batched), one GPU's share of C3, C4 and C5 on this GPU, a few passes each:
    the driver's one JSON line then carries every BASELINE config.  Failures are reported, never
    raised: the headline must not depend on them."""
    out = {}

    def run(name, factory, steps, warmup, cpu_leg=None):
        try:
            t0 = time.perf_counter()
            w = factory()
            par = None if args.no_parity else parity_gate(w)   # device vs reference, BEFORE timing
            dt, acc, last = measure(w, steps, warmup, sync)
            entry = w.describe(last[3], last[0])
            entry["parity"] = par
            inserted = getattr(w, "insert_s", 0.0)
            if inserted:       # (dirty-grid leg: the insertions between the steps are not the step)
                entry["grid_insertions_ms_per_step"] = inserted / steps * 1e3
                dt -= inserted
            if cpu_leg is not None:
                pending_cpu.append((name, cpu_leg, w))     # after every GPU leg, see below
            entry.update({
                "ms_per_step": dt / steps * 1e3,
                "candidates_per_s": acc["candidates_scored"] / dt,
                "summed_candidates_per_s": acc.get("coarse_candidates", 0) / dt,
                "matches_per_s": w.matches_per_step * steps / dt,
                "device_ms_per_step": acc["device_ms"] / steps,
                "steps": steps, "roofline": w.roofline(acc, steps, pmc),
                "setup_s": time.perf_counter() - t0 - dt,
            })
            out[name] = entry
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    # The CPU baselines (32 host threads for seconds each) run after ALL GPU legs: a host-bound
    # leg measured right behind one of them came out 1.5x slower (C1 batch: 0.19 vs 0.13 ms).
    pending_cpu = []
    cpu = {} if args.no_cpu_baseline else {
        "c1_batch128": lambda w: cpu_baseline_c1(w, min(3.0, args.cpu_seconds)),
        "c4": lambda w: cpu_baseline_c4(w, min(6.0, args.cpu_seconds)),
        "c5_single": lambda w: cpu_baseline_c5(w, min(5.0, args.cpu_seconds))}
    run("c1_single", lambda: Rt2DWorkload(args, device, matches=1), 200, 50)
    run("c1_batch128", lambda: Rt2DWorkload(args, device, matches=128), 200, 30, cpu.get("c1_batch128"))
    run("c1_batch128_dirty", lambda: Rt2DWorkload(args, device, matches=128, dirty=True), 50, 5)
    run("c1_batch128_grid400", lambda: Rt2DWorkload(args, device, matches=128, grid=400), 50, 5)
    # (host-bound legs right behind their parity gate -- 256 reference matches on every host
    # thread: thirty warm-up calls before the clock starts, the first ones run 1.3x slower)
    run("c1_batch1024", lambda: Rt2DWorkload(args, device, matches=1024), 100, 30)
    run("c1_batch1024_dirty", lambda: Rt2DWorkload(args, device, matches=1024, dirty=True), 30, 10)
    run("c1_batch128_8_threads", lambda: Rt2DPipelinedWorkload(args, device, 128, 8, 4), 25, 5)
    run("c1_tsdf", lambda: Rt2DTsdfWorkload(args, device), 30, 5)
    run("c2_easy", lambda: Fast2DConcurrentWorkload(args, device, 16, scans=1), 40, 5)
    sub = argparse.Namespace(**vars(args))
    sub.submaps = 16
    run("c3_share_16_submaps", lambda: Fast2DWorkload(sub, device, 0, 1, sharded=True), 5, 2)
    # (one GPU's share of config [2] when it is sharded over eight: 64 submaps -- a batch the
    # library runs as independent single searches over its host pool, round 6)
    sub64 = argparse.Namespace(**vars(args))
    sub64.submaps = C3_SUBMAPS_PER_GPU
    run("c3_share_64_submaps", lambda: Fast2DWorkload(sub64, device, 0, 1, sharded=True), 4, 2)
    run("c4", lambda: Rt3DWorkload(args, device), 3, 1, cpu.get("c4"))
    run("c5_single", lambda: Fast3DWorkload(args, device, pairs=1), 10, 2, cpu.get("c5_single"))
    run("c5_share_32_submaps", lambda: Fast3DWorkload(args, device, pairs=32), 4, 2)
    for name, cpu_leg, w in pending_cpu:
        try:
            out[name]["cpu_baseline"] = cpu_leg(w)
        except Exception as e:      # noqa: BLE001
            out[name]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}

    return out


# --------------------------------------------------------------------------------------
# The ONE line on stdout.  The driver's record keeps a line of a few KB (round 4's 30 KB line
# came back unparsed): the headline is scalars + five small objects, everything else -- the
# per-config blocks with their notes and nested rooflines -- goes to a file next to it.
# --------------------------------------------------------------------------------------
LINE_LIMIT = 4096

CONFIG_KEYS = ("name", "host_threads", "passes_per_step", "ms_per_pass", "candidates_per_step",
               "candidates_per_pass", "lowest_resolution_candidates_per_pass",
               "candidates_summed_per_pass", "matches_per_s",
               "device_ms_per_pass", "timed_region_s", "submaps_per_gpu", "rotations", "found",
               "nodes_expanded_per_step", "single_stream_ms_per_search",
               "constraints_found_node_wide", "best_match", "matches_per_step", "pairs_per_step")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                 "algorithmic_bytes", "hbm_frac_algorithmic", "hbm_frac_traffic",
                 "kernel_ms_in_the_timed_region", "frac_in_the_timed_region")
CPU_KEYS = ("value", "unit", "cores", "kind", "matches_per_s", "single_thread_candidates_per_s")
SUMMARY_KEYS = ("ms", "cand_per_s", "matches_per_s", "frac", "bound", "cpu", "parity", "error")


def _num(v, digits=6):
    """Floats to `digits` significant figures (the line is for reading; the details file keeps
    everything); containers recursively."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, (float, np.floating)):
        v = float(v)
        return v if v == 0 or not math.isfinite(v) else float(f"{v:.{digits}g}")
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, dict):
        return {k: _num(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_num(x, digits) for x in v]
    return v


def _clip(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[:limit - 3] + "..."


def headline(out, details_path):
    """The compact line from the full record: contract keys as they are; `config`, `roofline`,
    `cpu_baseline` cut down to their scalar keys (text clipped); `summary` one small object per
    measured config.  Asserts the size: a longer line is a bug here, not a driver problem."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                                "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data", "constraints_per_s")}
    cfg = out["config"]
    line["config"] = {"workload": _clip(cfg["workload"], 150)}
    line["config"].update({k: cfg[k] for k in CONFIG_KEYS if k in cfg})
    roof = out["roofline"]
    line["roofline"] = {"kernel": _clip(roof.get("kernel", ""), 80)}
    line["roofline"].update({k: roof[k] for k in ROOFLINE_KEYS if k in roof})
    line["roofline"]["traffic_source"] = None if roof.get("traffic") is None else "profiles/"
    if "cpu_baseline" in out:
        base = out["cpu_baseline"]
        line["cpu_baseline"] = {k: base[k] for k in CPU_KEYS if k in base}
        line["cpu_baseline"]["sample"] = _clip(base.get("sample", ""), 100)
    if "parity" in out:
        line["parity"] = out["parity"]
    line["details"] = details_path
    line["summary"] = {name: {k: e[k] for k in SUMMARY_KEYS if e.get(k) is not None}
                       for name, e in out.get("summary", {}).items()}
    line = _num(line)
    line["summary"] = _num(line["summary"], 4)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes (limit {LINE_LIMIT})"
    return text


def write_details(out, path):
    """The full record (per-config blocks, notes, nested rooflines): a file, and stderr."""
    if not path:
        return None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, default=float)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as e:
        sys.stderr.write(f"bench details not written ({e})\n")
        return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = args.gpus > 1 or world_size > 1 or args.force_dist
    if use_dist:
        assert world_size == args.gpus, "launch with torch.distributed.run --nproc-per-node N"
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if torch.cuda.is_available() else 0
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(device)
    torch_device = os.environ.get("CMX_BENCH_TORCH_DEVICE", f"cuda:{device}")   # tests: "cpu"

    name = args.config
    if name == "auto":
        name = "c2" if world_size == 1 and not args.force_dist else "c3"
    if name in ("c1", "c4"):
        assert world_size == 1, f"--config {name} is a single-GPU workload"
    workload = make_workload(name, args, device, rank, world_size)
    # c3 / sharded c5 always run their exchange (a no-op gather with one rank): same code path
    # at every N
    sharded = use_dist or name == "c3"
    exchange = getattr(workload, "exchange", None) if sharded else None
    threads = args.concurrency or (16 if name == "c2" and world_size == 1 and not use_dist else 1)
    pool = None
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(threads)

    def run_passes(num_passes, acc):
        """`num_passes` passes of the hot path (one step, or a warmup / calibration run); returns
        the last result."""
        last = None

        def add(result):
            for k in STAT_KEYS:
                acc[k] += result[3].get(k, 0)

        if pool is None:
            for k in range(num_passes):
                last = workload.search(k)
                if exchange is not None:
                    exchange(last[0], last[1], last[2], torch_device)
                add(last)
        elif exchange is None:
            # Independent searches issued from T host threads, each looping over its share.
            shares = [num_passes // threads + (1 if i < num_passes % threads else 0)
                      for i in range(threads)]

            def worker(args_):
                # (every thread sums the counters of its own searches -- between its calls, while
                # the other threads are inside theirs -- instead of the issuing thread walking all
                # results after the last search has returned: that walk was 5 % of a step)
                first, n = args_
                mine = {k: 0 for k in STAT_KEYS}
                r = None
                for j in range(n):
                    r = workload.search(first + j)
                    for k in STAT_KEYS:
                        mine[k] += r[3].get(k, 0)
                return mine, r
            starts = [sum(shares[:i]) for i in range(threads)]
            for mine, r in pool.map(worker, zip(starts, shares)):
                for k in STAT_KEYS:
                    acc[k] += mine[k]
                if r is not None:
                    last = r
        else:
            # Rounds of T concurrent searches, then the collectives of each pass on this thread
            # (collectives must be issued in the same order on every rank).
            done = 0
            while done < num_passes:
                n = min(threads, num_passes - done)
                for r in [f.result() for f in [pool.submit(workload.search, done + j)
                                               for j in range(n)]]:
                    exchange(r[0], r[1], r[2], torch_device)
                    add(r)
                    last = r
                done += n
        return last

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- parity gate: this rank's device result against the reference, before any timing ----
    parity = None if args.no_parity else parity_gate(workload)

    # ---- passes per step: a step is one pass over a BATCH of searches, sized (untimed) so that
    # it lasts >= 30 ms; every rank uses the same number ------------------------------------
    scratch = {k: 0.0 for k in STAT_KEYS}
    passes = args.passes_per_step

    def agree(count):
        """The same pass count on every rank (MAX).  A sharded pass contains collectives, so
        ranks must never run different numbers of passes -- not in the calibration either --
        and must take the same decisions about repeating it."""
        if not use_dist:
            return count
        t = torch.tensor([count], dtype=torch.int64, device=torch_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def round_up(count):
        count = int(min(8192, max(1, count)))
        if pool is not None:
            count = (count + threads - 1) // threads * threads
        return count

    if passes <= 0:
        probe = max(2 * threads, 4)
        run_passes(probe, scratch)           # first touches, clocks
        fence()
        t0 = time.perf_counter()
        run_passes(probe, scratch)
        fence()
        per_pass = (time.perf_counter() - t0) / probe
        passes = agree(round_up(math.ceil(0.030 / max(per_pass, 1e-7))))
        # one trial step at that size, then the final size (the probe above includes cold starts)
        for _ in range(2):
            fence()
            t0 = time.perf_counter()
            run_passes(passes, scratch)
            fence()
            step_s = time.perf_counter() - t0
            wanted = passes if step_s >= 0.030 else round_up(
                math.ceil(passes * 0.033 / max(step_s, 1e-6)))
            wanted = agree(wanted)
            if wanted == passes:
                break
            passes = wanted
    set_timing(False)          # (the probe and calibration above ran without them too: default)
    for _ in range(args.warmup):
        run_passes(passes, scratch)

    fence()
    gc.collect()
    gc.disable()        # see measure()
    acc = {k: 0.0 for k in STAT_KEYS}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        found, scores, poses, stats = run_passes(passes, acc)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    total_passes = passes * args.steps
    # One more step, untimed, with the library's HIP-event brackets on: the kernel / device times
    # of the roofline block (every rank runs it: a sharded pass contains collectives).
    set_timing(True)
    try:
        acc_ms = {k: 0.0 for k in STAT_KEYS}
        run_passes(passes, acc_ms)
        fence()
    finally:
        set_timing(False)
    for k in MS_KEYS:
        acc[k] = acc_ms[k] * args.steps

    # MAX over ranks of the elapsed time; SUM of the work.
    cand_local = acc["candidates_scored"]
    matches_local = workload.matches_per_step * total_passes
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        w = torch.tensor([int(cand_local), int(matches_local)], dtype=torch.int64,
                         device=torch_device)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        cand_total, matches_total = int(w[0].item()), int(w[1].item())
    else:
        cand_total, matches_total = cand_local, matches_local

    out = None
    if rank == 0:
        def pmc(kernel, config=""):
            return pmc_traffic_bytes(args.pmc_dir, args.pmc_tag, kernel, config)
        config = workload.describe(stats, found)
        config.update({
            "name": name,
            "host_threads": threads,
            "passes_per_step": passes,
            "ms_per_pass": elapsed / total_passes * 1e3,
            "candidates_per_step": cand_local / args.steps,
            "candidates_per_pass": cand_local / total_passes,
            "lowest_resolution_candidates_per_pass": acc["coarse_candidates"] / total_passes,
            "matches_per_s": matches_total / elapsed,
            "device_ms_per_pass": acc["device_ms"] / total_passes,
            "timed_region_s": elapsed,
            "candidate_count": "every scored candidate at every depth, counted once where it is "
                               "scored (dive and tie re-scoring included); fast 2D from depth 5 on: "
                               "the lowest-resolution candidates get ONE sum per three rotations "
                               "(group bounds, DESIGN 5.1) -- candidates_summed_per_pass counts sums",
        })
        if name in ("c2", "c3") and args.depth >= 5:
            coarse = acc["coarse_candidates"] / total_passes
            config["candidates_summed_per_pass"] = cand_local / total_passes - coarse + coarse / 3.0
        roof = workload.roofline(acc, total_passes, pmc)
        roof["traffic_source"] = (None if roof.get("traffic") is None else
                                  f"rocprofv3 --pmc passes of this command, {args.pmc_dir}/"
                                  f"{args.pmc_tag}*_pmc_*.csv (not measured in this run)")
        out = {
            "metric": "candidate poses scored/sec",
            "value": cand_total / elapsed,
            "unit": "candidates/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32" if name in ("c2", "c3", "c5") else
                     ("u16->int32 bulk, f32 finalists" if name == "c1" else
                      "u8->int32 bounds, f32 finalists"),
            "data": "synthetic",
            "constraints_per_s": matches_total / elapsed,
            "config": config,
            "roofline": roof,
        }
        if parity is not None:
            out["parity"] = parity
        summary = {name: {"ms": config["ms_per_pass"], "cand_per_s": out["value"],
                          "frac": roof.get("frac"), "bound": roof.get("bound"),
                          "parity": _parity_word(parity)}}
        if world_size == 1 and not use_dist and not args.no_other and name == "c2":
            # Single-stream latency of the headline workload, then every other BASELINE config.
            single = Fast2DWorkload(args, device, 0, 1, sharded=False)
            dt1, acc1, _ = measure(single, 200, 20, torch.cuda.synchronize)
            config["single_stream_ms_per_search"] = dt1 / 200 * 1e3
            config["single_stream_candidates_per_s"] = acc1["candidates_scored"] / dt1
            if config.get("host_threads", 1) > 1:
                # A launch that shares the chip with seven others of its kind has no meaningful
                # "duration": the roofline block is taken from this single-stream leg (same
                # workload, same run, HIP events on the kernel's own stream) and keeps the
                # concurrent figure next to it.
                concurrent = roof
                roof = single.roofline(acc1, 200, pmc)
                roof["kernel_ms_in_the_timed_region"] = concurrent.get("kernel_ms")
                roof["frac_in_the_timed_region"] = concurrent.get("frac")
                roof["note"] += ("  The timed region issues searches from "
                                 f"{config['host_threads']} host threads; kernel_ms / frac here are "
                                 "from the single-stream leg that follows it, *_in_the_timed_region "
                                 "the average per launch while eight searches overlap.")
                out["roofline"] = roof
                summary[name]["frac"] = roof.get("frac")
            other = other_configs(args, device, torch.cuda.synchronize, pmc)
            out["details"] = other
            # The driver's record keeps scalars: every config's line flat in `config` ...
            for key, e in other.items():
                short = {"c1_single": "c1_single", "c1_batch128": "c1b128",
                         "c1_batch128_dirty": "c1b128_dirty", "c1_batch128_grid400": "c1b128_g400",
                         "c1_batch1024": "c1b1024",
                         "c1_batch128_8_threads": "c1b128t8", "c1_tsdf": "c1_tsdf",
                         "c2_easy": "c2_easy", "c1_batch1024_dirty": "c1b1024_dirty",
                         "c3_share_16_submaps": "c3s16", "c3_share_64_submaps": "c3s64",
                         "c4": "c4", "c5_single": "c5_single",
                         "c5_share_32_submaps": "c5s32"}.get(key, key)
                if "error" in e:
                    config[f"{short}_error"] = e["error"][:80]
                    summary[short] = {"error": e["error"][:60]}
                    if e["error"].startswith("ParityError") and "parity" in out:
                        out["parity"].setdefault("failed_elsewhere", []).append(short)
                    continue
                r = e.get("roofline") or {}
                config[f"{short}_ms"] = e["ms_per_step"]
                config[f"{short}_cand_per_s"] = e["candidates_per_s"]
                config[f"{short}_matches_per_s"] = e["matches_per_s"]
                if r:
                    config[f"{short}_frac"] = r.get("frac")
                    config[f"{short}_bound"] = r.get("bound")
                    config[f"{short}_kernel_ms"] = r.get("kernel_ms")
                c = e.get("cpu_baseline") or {}
                if "value" in c:
                    config[f"{short}_cpu"] = c["value"]
                    config[f"{short}_cpu_unit"] = c["unit"]
                    config[f"{short}_cpu_cores"] = c["cores"]
                summary[short] = {"ms": e["ms_per_step"], "cand_per_s": e["candidates_per_s"],
                                  "matches_per_s": e["matches_per_s"], "frac": r.get("frac"),
                                  "bound": r.get("bound"), "cpu": c.get("value"),
                                  "cpu_unit": c.get("unit"), "cpu_cores": c.get("cores"),
                                  "parity": _parity_word(e.get("parity"))}
        # The headline's CPU baseline runs LAST (after every GPU leg: seconds of 32 busy host
        # threads right before a host-bound GPU leg distort it).  JSON key order is irrelevant.
        if not args.no_cpu_baseline and world_size == 1 and name in ("c2", "c3"):   # rank 0, N = 1
            out["cpu_baseline"] = cpu_baseline(workload.cells0, workload.lim0, args.depth,
                                               workload.host_scans, args.min_score,
                                               args.cpu_seconds)
        elif not args.no_cpu_baseline and world_size == 1 and not use_dist:
            leg = {"c1": cpu_baseline_c1, "c4": cpu_baseline_c4, "c5": cpu_baseline_c5}[name]
            try:
                out["cpu_baseline"] = leg(workload, args.cpu_seconds)
            except Exception as e:      # noqa: BLE001
                sys.stderr.write(f"cpu baseline unavailable: {e}\n")
        if "cpu_baseline" in out:
            summary[name]["cpu"] = out["cpu_baseline"]["value"]
            summary[name]["cpu_cores"] = out["cpu_baseline"]["cores"]
        # ... and once more, compact, as the LAST key of the line (the tail of stdout).
        out["summary"] = summary
    if pool is not None:
        pool.shutdown()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which a pipe only sees at exit:
        # drain it first so that the JSON line is the LAST thing on stdout.
        import ctypes
        ctypes.CDLL(None).fflush(None)
        details = write_details(out, args.details)
        sys.stdout.write(headline(out, details) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
