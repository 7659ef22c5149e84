#!/usr/bin/env python3
"""bench.py — cell-steps/s of the grid microclimate solver on MI355X.

One "step" = one pass of the hot path (runmicro1Cpp geometry: vector forcing, reqhgt = 0.05 m, all 10
outputs) over the whole workload of a rank: a simulated year (8760 hourly steps) of its raster block.
Inputs are resident in HBM before the timed region; outputs go to a device ring ("HBM ring, no D2H")
because a year of outputs (11.8 TB at 4096^2) fits nowhere.

  --config 2 (default)  BASELINE.json configs[2]: 4096 x 4096 synthetic DTM per GPU, terrain inputs (slope, aspect,
                        24 horizons, sky view, 8 wind-shelter maps) pre-computed ON THE DEVICE from the DTM
  --config 1            configs[1]: 1024 x 1024, SURVEY 8d's random terrain inputs
  --config 3            configs[3]: 8192 x 8192 dealt to 8 ranks in row blocks (strong scaling), device terrain with
                        the +-128-row halo exchanged over RCCL; with fewer than 8 GPUs every rank runs its block of
                        the 8-block partition (one rank's share), halo rows generated instead of exchanged
  --config 4            configs[4]: 4096 x 4096 + the snow branch (see run_snow_config)

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child before torch / HIP are imported and relays
rank 0's JSON line (never re-executes a process that has touched the GPU).  Under an external torch.distributed.run
(the driver's way) it is one rank.  The only data-path collective of the solver is the all-reduce of the (sum, count)
of log(twi)/tfact (src/microclimfCpp.cpp:993-1004); the terrain pre-compute exchanges halo rows point-to-point.

After the timed region (outside it) a sample of cells of the LAST ring slot is fetched (mcf_plan_fetch_cells) and
compared with the CPU oracle run on exactly those cells: `verified`.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

CONFIGS = {
    1: dict(rows=1024, cols=1024, terrain="random", scaling="weak", share=0),
    2: dict(rows=4096, cols=4096, terrain="device", scaling="weak", share=0),
    3: dict(rows=8192, cols=8192, terrain="device", scaling="strong", share=8),
    4: dict(rows=4096, cols=4096, terrain="device", scaling="strong", share=8),
}
KERNEL_SOURCES = ("microclimf_amd/csrc/mcf_kernels.hip", "microclimf_amd/csrc/mcf_device.hpp",
                  "microclimf_amd/csrc/mcf_kernels.h", "microclimf_amd/csrc/Makefile")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[N]; sets --rows/--cols/--terrain/--scaling unless given")
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (weak) / of the whole raster (strong)")
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--tsteps", type=int, default=8760)
    ap.add_argument("--reqhgt", type=float, default=0.05)
    ap.add_argument("--ring-days", type=int, default=10)
    ap.add_argument("--ring-slots", type=int, default=2)
    ap.add_argument("--ring-gb", type=float, default=262.0, help="HBM the plan may use for tables + output ring")
    ap.add_argument("--cells-per-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (configs[1], array forcing, "
                                                                 "coarse array forcing)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--verify-cells", type=int, default=256)
    ap.add_argument("--cpu-sample", type=str, default="144x144x720")
    ap.add_argument("--terrain", choices=["random", "device"], default=None,
                    help="'device': slope/aspect/hor/svfa/wsa come from mcf_precompute_terrain run on the synthetic DTM; "
                         "'random': SURVEY 8d's random terrain inputs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="'weak': --rows rows PER GPU; 'strong': --rows is the whole raster, dealt to the ranks in row blocks")
    ap.add_argument("--share", type=int, default=None,
                    help="strong scaling only: partition the raster into this many row blocks even when fewer ranks run "
                         "(rank r solves block r) — one rank's share of an 8-GPU job on a 1-GPU box")
    ap.add_argument("--coarse", type=str, default="",
                    help="primary measurement in the `.runmodel2Cpp` geometry instead: CRxCC coarse climate grid interpolated "
                         "inside the solver (mcf.h array_forcing == 2)")
    ap.add_argument("--array-forcing", action="store_true",
                    help="primary measurement in the runmicro2Cpp geometry instead; ring_slots x ring_days days of forcing "
                         "are resident in HBM and solved repeatedly")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY (tests/test_bench_launcher_cpu.py): gloo backend, no GPU, a stand-in for the solver — "
                         "exercises the rank fan-out, partition and collectives; the line carries \"stub\": true")
    a = ap.parse_args(argv)
    preset = CONFIGS[a.config]
    for k, v in preset.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


# ----------------------------------------------------------------------------------------------------------------
# N > 1 without an external launcher: fan out BEFORE anything touches the GPU
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Parent side of `bench.py --gpus N`: one child process group, one rank per GPU, rank 0's JSON line relayed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if p.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank child exited with {p.returncode}" + ("" if line else " and printed no result"),
              file=sys.stderr)
        return p.returncode or 1
    return 0


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline legs (the oracle as the thing timed: allowed for this leg only)
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(args):
    """The oracle (a scalar C restatement of the reference loop, kind 'port') timed on one host core over a bounded
    sample of the same synthetic workload."""
    from microclimf_amd import synthetic
    from oracle import oracle as O
    r, c, t = (int(v) for v in args.cpu_sample.split("x"))
    a = synthetic.workload(r, c, t, reqhgt=args.reqhgt, start_doy=152)
    O.load()
    t0 = time.perf_counter()
    O.run_grid(**a)
    dt = time.perf_counter() - t0
    valid = int((~np.isnan(a["vegp"]["hgt"])).sum())
    return {"value": valid * (t // 24) * 24 / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
            "sample": f"{r}x{c} cells x {t} hourly steps of the same seeded synthetic workload, "
                      f"oracle/mcf_oracle.c (gcc -O2, 1 thread), {dt:.1f} s"}


def _cpu_worker(job):
    from microclimf_amd import synthetic
    from oracle import oracle as O
    r, c, t, reqhgt, seed = job
    a = synthetic.workload(r, c, t, reqhgt=reqhgt, start_doy=152, seed=seed)
    O.load()
    t0 = time.perf_counter()
    O.run_grid(**a)
    return int((~np.isnan(a["vegp"]["hgt"])).sum()) * (t // 24) * 24, time.perf_counter() - t0


def cpu_baseline_all_cores(args):
    """The same oracle on every host core the job may use ('1 thread and all host cores'): one process per
    core, each solving its own raster of the sample's size — cells are independent, so this is what an OpenMP loop over
    cells would give.  Forked BEFORE the GPU is initialised (no exec from a process that holds the device)."""
    import multiprocessing as mp
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    r, c, t = (int(v) for v in args.cpu_sample.split("x"))
    jobs = [(r, c, t, args.reqhgt, 20240321 + k) for k in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    units = sum(u for u, _ in res)
    busy = max(d for _, d in res)
    return {"value": units / busy, "unit": "cell-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x ({r}x{c} cells x {t} hourly steps), oracle/mcf_oracle.c (gcc -O2), slowest "
                      f"worker {busy:.1f} s, wall {wall:.1f} s incl. input generation"}


# ----------------------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------------------
def kernel_hash() -> str:
    if os.environ.get("MCF_KERNEL_HASH"):          # (tools/summarize_*.py: the stamp of the tree a profile was TAKEN from, recorded when
        return os.environ["MCF_KERNEL_HASH"]       # the call was launched, if the tree has moved on by the time it is summarised)
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()[:16]


SNOW_KERNEL_SOURCES = KERNEL_SOURCES + ("microclimf_amd/csrc/mcf_snow.hip", "microclimf_amd/csrc/mcf_snow_device.hpp",
                                        "microclimf_amd/csrc/mcf_terrain.hip")


def snow_kernel_hash():
    """... of everything the configs[4] pipeline launches (solver, snow model, snow-day microclimate, terrain refresh)"""
    if os.environ.get("MCF_SNOW_KERNEL_HASH"):
        return os.environ["MCF_SNOW_KERNEL_HASH"]
    h = hashlib.sha256()
    for rel in SNOW_KERNEL_SOURCES:
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()[:16]


def committed_pipeline_traffic():
    """HBM bytes of a simulated year of the configs[4] pipeline, summed over its kernels from the newest
    profiles/*_c4_aux_pmc_summary.json (tools/profile_aux.sh <tag>_c4 bench.py --config 4 --share 8 ...; per kernel launches x
    mean FETCH_SIZE x 2 + WRITE_SIZE, separate counter passes) — withheld, with the reason, when it was taken from other sources."""
    best, note = None, "no profiles/*_c4_aux_pmc_summary.json"
    for pf in sorted((ROOT / "profiles").glob("*_c4_aux_pmc_summary.json")):
        try:
            pj = json.loads(pf.read_text())
        except Exception:
            continue
        meta = pj.get("_meta") or {}
        if meta.get("kernel_hash") != snow_kernel_hash():
            note = f"stale: {pf.name} was taken from sources {meta.get('kernel_hash')}, this run is {snow_kernel_hash()}"
            continue
        tot, per = 0.0, {}
        for k, e in pj.items():
            hb = e.get("hbm_bytes_per_launch") if isinstance(e, dict) else None
            if not hb:
                continue
            b = (hb["read"] + hb["write"]) * e["launches"]
            per[k[:40]] = {"launches": e["launches"], "read": hb["read"] * e["launches"], "write": hb["write"] * e["launches"]}
            tot += b
        best, note = {"bytes_per_profiled_run": tot, "kernels": per, "source": f"profiles/{pf.name}",
                      "profiled_command": meta.get("command"), "kernel_hash": meta.get("kernel_hash")}, None
    return best, note


def committed_counters(rows, cols, ring_days):
    """HBM traffic and VALU counters of k_solve from profiles/traffic.json — measured by rocprofv3 --pmc in separate
    passes (tools/profile_round.sh), stamped with the hash of the kernel sources they were taken from.  A stamp that
    does not match the sources being run means the counters are STALE: they are withheld and the line says so."""
    tf = ROOT / "profiles" / "traffic.json"
    if not tf.exists():
        return None, None, "no profiles/traffic.json"
    try:
        tj = json.loads(tf.read_text())
    except Exception as e:
        return None, None, f"unreadable profiles/traffic.json: {e}"
    entries = tj.get("entries", [tj])
    for e in entries:
        if e.get("rows") == rows and e.get("cols") == cols and e.get("ring_days") == ring_days:
            if e.get("kernel_hash") != kernel_hash():
                return None, None, (f"stale: counters of {e.get('tag')} were taken from kernel sources "
                                    f"{e.get('kernel_hash')}, this run is {kernel_hash()}")
            pf = ROOT / "profiles" / f"{e.get('tag')}_pmc_summary.json"
            pj = json.loads(pf.read_text()) if pf.exists() else None
            return e, pj, None
    # no counter run of this raster: k_solve's bytes per cell-step do not depend on the raster's shape — the current-source entry
    # whose launches are closest in days, its reads (per-launch constant images) to be scaled by cells, its writes by cell-steps
    cand = [e for e in entries if e.get("kernel_hash") == kernel_hash() and e.get("read_bytes") is not None and e.get("ring_days")]
    if cand:
        e = dict(min(cand, key=lambda x: abs(x["ring_days"] - ring_days)))
        e["other_raster"] = f"{e.get('rows')}x{e.get('cols')}, {e.get('ring_days')}-day launches"
        pf = ROOT / "profiles" / f"{e.get('tag')}_pmc_summary.json"
        pj = json.loads(pf.read_text()) if pf.exists() else None
        return e, pj, None
    return None, None, f"no counters for {rows}x{cols} with {ring_days}-day launches"


def fit_ring(args, cells, af):
    """The output ring (and, with array forcing, the forcing slabs) must fit the GPU beside the plan's tables (1.2 KB per
    cell): shrink until slots x days x 24 h x cells x 8 B x (10 outputs [+ 15 forcing arrays]) stays under the budget
    (--ring-gb, default 230 of the 288 GB).  Longer launches first: every launch pays a fixed ~6.6 us per tile (staging the
    tile's constants, the soil-state prologue, the tail), 15 % of a 4-day launch and 9 % of a 7-day one."""
    cpb = args.cells_per_block or (32 if af else 21)
    pad = (((cpb * 24 + 63) // 64) * 64) / (cpb * 24.0)       # the tiled ring's blocks are whole wave-rows (21 cells: 512 / 504)
    per_day = cells * 24 * 8 * (10 * pad + (15 if af else 0))
    slots, days = args.ring_slots, args.ring_days
    budget = args.ring_gb * 1e9 - cells * 1500.0      # inputs 23 x 8 B, hor / wsa 256 B, tile-major constant table 976 B per cell
    while slots * days * per_day > budget and (slots > 1 or days > 1):
        if slots > 1 and not af:
            slots -= 1
        elif days > 1:
            days -= 1
        else:
            break
    return slots, days


class Clock:
    """Barrier + device synchronisation on both sides of the timed region, MAX over ranks."""

    def __init__(self, use_dist, dist, torch):
        self.use_dist, self.dist, self.torch = use_dist, dist, torch

    def fence(self, plan=None):
        if plan is not None:
            plan.sync()
        if self.torch is not None and self.torch.cuda.is_available():
            self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        if self.torch is not None and self.torch.cuda.is_available():
            self.torch.cuda.synchronize()


def timed_year(plan, ndays, ring_days, ring_slots, steps, warmup, clock):
    """`warmup` untimed and `steps` timed passes over the series in launches of `ring_days` days into the ring.
    Returns (seconds, kernel ms, launches, resident) — resident[(slot, day offset in slot)] = absolute day."""
    resident = {}

    def one_step():
        slot = 0
        for d0 in range(0, ndays, ring_days):
            nd = min(ring_days, ndays - d0)
            plan.run_days(d0, nd, slot)
            for k in range(nd):
                resident[(slot, k)] = d0 + k
            slot = (slot + 1) % ring_slots

    for _ in range(warmup):
        one_step()
    clock.fence(plan)
    plan.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    clock.fence(plan)
    dt = time.perf_counter() - t0
    kms, klaunches = plan.kernel_stats()
    plan.kernel_timing(False)
    return dt, kms, klaunches, resident


def verify_sample(plan, a, resident, twi_mean, ncells, seed=7, af=False, cpos=None):
    """Outside the timed region: fetches `ncells` cells x every day still resident in the ring (what the LAST timed pass
    left there) and compares all ten outputs with the CPU oracle run on exactly those cells (same forcing, the raster's
    twi mean installed).  The oracle is the checker here, never the thing timed.
    af: array forcing — the cells' own forcing columns go to the runmicro2Cpp oracle.  cpos (coarse array forcing): the
    sample is a sub-grid of raster rows x columns, for which `.runmodel2Cpp`'s resampling is restated with numpy
    (oracle/coarse_oracle.py) before the same oracle."""
    import ctypes as C
    from oracle import oracle as O
    hgt = a["vegp"]["hgt"]
    rows, cols = hgt.shape
    N = rows * cols
    rng = np.random.default_rng(seed)
    flat_h = hgt.reshape(-1, order="F")
    sub = dict(a)
    if cpos is not None:
        from oracle import coarse_oracle as CO
        k = max(2, int(round(ncells ** 0.5)))
        ri = np.sort(rng.choice(rows, size=min(k, rows), replace=False))
        ci = np.sort(rng.choice(cols, size=min(k, cols), replace=False))
        cells = (ri[:, None] + rows * ci[None, :]).reshape(-1, order="F").astype(np.int64)
        grid = lambda m: np.asfortranarray(np.asarray(m)[np.ix_(ri, ci)])                   # noqa: E731
        sub["vegp"] = {kk: grid(v) for kk, v in a["vegp"].items()}
        sub["soilc"] = {kk: grid(v) for kk, v in a["soilc"].items()}
        sub["climdata"], sub["pointm"] = CO.expand(a["climdata"], a["pointm"], np.asarray(cpos["rowpos"])[ri], np.asarray(cpos["colpos"])[ci])
        sub["lat"], sub["lon"] = grid(a["lat"]), grid(a["lon"])
    else:
        pick = set(int(v) for v in rng.choice(N, size=min(ncells, N), replace=False))
        # make sure the three cell classes are in the sample: NA cells, bare ground, vegetated
        for mask in (np.isnan(flat_h), flat_h == 0.0):
            idx = np.flatnonzero(mask)
            for v in idx[:4]:
                pick.add(int(v))
        cells = np.array(sorted(pick), dtype=np.int64)
    n = cells.size

    def take(m):
        m = np.asarray(m)
        if m.ndim < 2:
            return m
        if m.ndim == 2:
            return np.asfortranarray(m.reshape(-1, order="F")[cells].reshape(n, 1))
        d = m.shape[2]
        return np.asfortranarray(m.reshape(N, d, order="F")[cells].reshape(n, 1, d))

    if cpos is None:
        sub["vegp"] = {k: take(v) for k, v in a["vegp"].items()}
        sub["soilc"] = {k: take(v) for k, v in a["soilc"].items()}
        if af:
            sub["climdata"] = {k: take(v) for k, v in a["climdata"].items()}
            sub["pointm"] = {k: take(v) for k, v in a["pointm"].items()}
            sub["lat"], sub["lon"] = take(a["lat"]), take(a["lon"])
    lib = O.load()
    lib.orc_set_twi_mean_override.argtypes = [C.c_double, C.c_int]
    lib.orc_set_twi_mean_override(float(twi_mean), 1)
    t0 = time.perf_counter()
    try:
        want = O.run_grid(**sub, array_forcing=bool(af or cpos is not None))
    finally:
        lib.orc_set_twi_mean_override(0.0, 0)
    t_or = time.perf_counter() - t0
    worst, nan_ok, nvals, worst_var = 0.0, True, 0, None
    days = sorted(set(resident.values()))
    for (slot, off), day in sorted(resident.items()):
        for var, w in want.items():
            got = plan.fetch_cells(slot, var, off * 24, 24, cells)
            ref = w.reshape(n, w.shape[2], order="F")[:, day * 24:(day + 1) * 24]
            if not np.array_equal(np.isnan(got), np.isnan(ref)):
                nan_ok = False
            fin = np.isfinite(ref) & np.isfinite(got)
            if fin.any():
                e = float((np.abs(got[fin] - ref[fin]) / (1.0 + np.abs(ref[fin]))).max())
                if e > worst:
                    worst, worst_var = e, var
            nvals += int(ref.size)
    tol = 1e-6
    return {"cells": int(n), "steps": 24 * len(resident), "days": [int(days[0]), int(days[-1])], "values": nvals,
            "max_scaled_err": worst, "worst_var": worst_var, "na_pattern_equal": bool(nan_ok), "tolerance": tol,
            "ok": bool(nan_ok and worst <= tol),
            "how": "mcf_plan_fetch_cells on the ring as the last timed pass left it vs oracle/mcf_oracle.c on the same "
                   f"cells over the whole series ({t_or:.1f} s), |HIP - oracle| / (1 + |oracle|)"
                   + ("; the coarse fields resampled for the sample by oracle/coarse_oracle.py" if cpos is not None else "")}


def committed_summary(mode, rows, cols, ring_days):
    """PMC summary of the array-forcing ('af') / coarse ('coarse') solver from profiles/*_<mode>_pmc_summary.json: the newest
    one whose kernel hash is that of the sources being run (tools/profile_round.sh <tag>_<mode> --config 1 ...)."""
    best, note = None, f"no profiles/*_{mode}_pmc_summary.json for {rows}x{cols} with {ring_days}-day launches"
    for pf in sorted((ROOT / "profiles").glob(f"*_{mode}_pmc_summary.json")):
        try:
            pj = json.loads(pf.read_text())
        except Exception:
            continue
        if (pj.get("rows"), pj.get("cols"), pj.get("ring_days")) != (rows, cols, ring_days):
            continue
        if pj.get("kernel_hash") != kernel_hash():
            note = f"stale: {pf.name} was taken from kernel sources {pj.get('kernel_hash')}, this run is {kernel_hash()}"
            continue
        best, note = (pf, pj), None
    return best, note


def valu_block(pj, rate_per_gpu, source, khash):
    """Issue accounting from a PMC summary.  SQ_ACTIVE_INST_VALU counts one unit per VALU wave-instruction and four per
    v_rcp_f64 / v_rsq_f64 (profiles/r04_microbench_waves_pmc.txt): `issue_slots`.  A slot costs 4.16 cycles of a SIMD with
    four resident waves issuing nothing but fp64 (4.38 from the waves' own clocks; 32-bit VOP1 / VOP2 cost 2.2-2.4 alone
    but only v_mov_b32 keeps that price beside fp64 work): the issue peak is 1024 SIMDs x clock / 4.16 slots per second."""
    m = pj.get("per_launch_mean", {})
    if "SQ_INSTS_VALU" not in m:
        return None
    cs = pj["cell_steps_per_launch"]
    per_cs = m["SQ_INSTS_VALU"] * 64.0 / cs
    slots = m.get("SQ_ACTIVE_INST_VALU", m["SQ_INSTS_VALU"]) * 64.0 / cs
    clock = float(pj.get("clock_ghz") or 2.4)
    return {"insts_per_cell_step": per_cs, "issue_slots_per_cell_step": slots,
            "frac_of_issue_peak": slots * rate_per_gpu / 64.0 / (1024 * clock * 1e9 / 4.16),
            "cycles_per_slot": 4.16, "clock_ghz_measured": clock, "busy_fraction_measured": pj.get("valu_busy_fraction"),
            "source": source, "kernel_hash": khash}


def roofline_block(valid, T, steps_per_launch, avg_ms, klaunches, af, rows, cols, ring_days, rate_per_gpu, coarse):
    """Roofline of the dominant kernel (k_solve).  ALGORITHMIC bytes per launch = valid cells x steps per launch x
    (8 B x n_out written + 440 B / T read); array forcing reads 15 arrays per cell-step (+8 B for the mxtc pre-pass):
    208 B [SURVEY 8d].  The HBM fraction is what BASELINE.json's metric asks for; the roof that BINDS is fp64 VALU
    issue (software exp / log / divide), reported next to it from the committed PMC counters of the same sources."""
    n_out = 10
    bpl = valid * steps_per_launch * ((8.0 * n_out + 128.0) if af else (8.0 * n_out + 440.0 / T))
    achieved = bpl / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, valu, note, basis = None, None, None, None
    cs_bench = valid * steps_per_launch           # cell-steps of THIS run's mean launch: what `bpl` and `avg_ms` are per

    def on_this_basis(read_b, write_b, cs_pmc, source):
        """The counter bytes were taken from a run whose mean launch covered cs_pmc cell-steps (80 days: whole launches only);
        this run's mean launch covers cs_bench (a year ends in a shorter launch).  Writes — and array forcing's reads — go with
        the cell-steps; vector forcing's reads are the per-launch constant images and do not.  `traffic` is on THIS run's
        basis, the one `algorithmic_bytes_per_launch` is on."""
        k = cs_bench / cs_pmc if cs_pmc else 1.0
        t = (read_b * (k if (af or coarse) else 1.0)) + write_b * k
        return t, {"source": source, "counter_run_cell_steps_per_launch": cs_pmc, "this_run_cell_steps_per_launch": cs_bench,
                   "scaled_by": k, "counter_run_bytes": {"read": read_b, "write": write_b},
                   "ratio_to_algorithmic": t / bpl if bpl else None}

    if not af and not coarse:
        e, pj, note = committed_counters(rows, cols, ring_days)
        if e is not None:
            cs_pmc = (pj or {}).get("cell_steps_per_launch") or e.get("cell_steps_per_launch")
            if cs_pmc and e.get("read_bytes") is not None and e.get("other_raster"):
                # (another raster's counters: the reads go with the cells of a launch, the writes with its cell-steps)
                k_w = cs_bench / cs_pmc
                k_r = (cs_bench / max(steps_per_launch, 1e-9)) / (cs_pmc / (24.0 * e["ring_days"]))
                traffic = e["read_bytes"] * k_r + e["write_bytes"] * k_w
                basis = {"source": f"profiles/traffic.json [{e.get('tag')}] — counters of ANOTHER raster ({e['other_raster']}): k_solve's "
                                   "bytes per cell-step do not depend on the raster's shape",
                         "counter_run_cell_steps_per_launch": cs_pmc, "this_run_cell_steps_per_launch": cs_bench,
                         "reads_scaled_by_cells": k_r, "writes_scaled_by_cell_steps": k_w,
                         "counter_run_bytes": {"read": e["read_bytes"], "write": e["write_bytes"]},
                         "ratio_to_algorithmic": traffic / bpl if bpl else None}
            elif cs_pmc and e.get("read_bytes") is not None:
                traffic, basis = on_this_basis(e["read_bytes"], e["write_bytes"], cs_pmc, f"profiles/traffic.json [{e.get('tag')}]")
            else:
                traffic = e.get("hbm_bytes_per_launch")
            if pj is not None:
                valu = valu_block(pj, rate_per_gpu, f"profiles/{e.get('tag')}_pmc_summary.json", e.get("kernel_hash"))
    else:
        best, note = committed_summary("af" if af else "coarse", rows, cols, ring_days)
        if best is not None:
            pf, pj = best
            hb = pj.get("hbm_bytes_per_launch") or {}
            if hb.get("read") is not None and pj.get("cell_steps_per_launch"):
                traffic, basis = on_this_basis(hb["read"], hb["write"], pj["cell_steps_per_launch"], f"profiles/{pf.name}")
            else:
                traffic = hb.get("total")
            valu = valu_block(pj, rate_per_gpu, f"profiles/{pf.name}", pj.get("kernel_hash"))
    rb = {"bound": "fp64_valu", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
          "frac_is": "achieved algorithmic HBM bytes / 8 TB/s (the metric BASELINE.json names); the binding roof is "
                     "fp64 VALU issue at a power-limited clock, see `valu`: with the output stream beside it the shader clock "
                     "drops 2.38 -> 2.15 GHz (10.8 %) and the launch costs 7.9 % more cycles (timing experiments: "
                     "profiles/r03_timing_experiments.txt, DESIGN 5)",
          "traffic": traffic, "kernel": "k_solve", "avg_launch_ms": avg_ms, "launches": int(klaunches),
          "algorithmic_bytes_per_launch": bpl, "traffic_basis": basis, "valu": valu}
    if note:
        rb["counters"] = note
    return rb


# ----------------------------------------------------------------------------------------------------------------
def stub_main(args, world, rank):
    """TEST ONLY: the rank fan-out, partition and collectives with a stand-in solver over gloo (no GPU, no HIP)."""
    import torch.distributed as dist
    from microclimf_amd import synthetic
    from microclimf_amd.distributed import allreduce_max, allreduce_sum, allreduce_twi_mean, row_block
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols = args.rows, args.cols
    row0, rows_total = rank * rows, rows * world
    if args.scaling == "strong":
        rows_total = args.rows
        row0, rows = row_block(rank, max(world, args.share or 0), rows_total)
    vegp, soilc, _ = synthetic.rasters(rows, cols, row0, rows_total, reqhgt=args.reqhgt)
    twi = soilc["twi"]
    s, n = float(np.sum(np.log(twi) / 1.5)), float(twi.size)
    mean = allreduce_twi_mean(s, n)
    valid = float((~np.isnan(vegp["hgt"])).sum())
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01)
    if world > 1:
        dist.barrier()
    dt = allreduce_max(time.perf_counter() - t0)
    valid_all = allreduce_sum(valid)
    if rank == 0:
        ndays = args.tsteps // 24
        print(json.dumps({"metric": "cell-steps/s", "value": valid_all * ndays * 24 * args.steps / dt, "unit": "cell-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic", "stub": True,
                          "config": {"workload": "STUB (no solver ran): launcher / partition / collectives only",
                                     "rows_per_gpu": rows, "cols": cols, "valid_cells": int(valid_all), "twi_mean": mean}}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def build_workload(args, synthetic, rank, world, local_rank, rows, cols, row0, rows_total, T, af, coarse, exchange_ok):
    """Inputs of one rank's block (+ the device terrain pre-compute when asked for).  Returns (args dict, coarse
    positions or None, terrain seconds or None)."""
    cpos = None
    if coarse:
        a, rp, cp = synthetic.coarse_workload(rows, cols, T, coarse[0], coarse[1], reqhgt=args.reqhgt, row0=row0,
                                              rows_total=rows_total)
        cpos = {"rowpos": rp, "colpos": cp}
    else:
        a = synthetic.workload(rows, cols, T, reqhgt=args.reqhgt, row0=row0, rows_total=rows_total,
                               array_forcing=af, start_doy=152 if af else 1)
    terrain_s = None
    if args.terrain == "device":
        from microclimf_amd.terrain import HALO, precompute_terrain, precompute_terrain_tiled
        if exchange_ok:
            _, _, dtm = synthetic.rasters(rows, cols, row0, rows_total, reqhgt=args.reqhgt)
            tt0 = time.perf_counter()
            ter = precompute_terrain_tiled(dtm, 1.0, a["zref"], rank, world, row0, rows_total, device=local_rank)
        else:
            # one rank's share of a larger partition without its neighbours: the halo rows are generated, not exchanged
            hn = min(HALO, row0)
            hs = min(HALO, rows_total - (row0 + rows))
            _, _, ext = synthetic.rasters(rows + hn + hs, cols, row0 - hn, rows_total, reqhgt=args.reqhgt)
            tt0 = time.perf_counter()
            ter = precompute_terrain(ext, 1.0, a["zref"], halo_north=hn, halo_south=hs, row0=row0, rows_total=rows_total,
                                     device=local_rank)
        terrain_s = time.perf_counter() - tt0
        a["soilc"].update(ter)
    return a, cpos, terrain_s


def measure(args, torch, dist, use_dist, rank, world, local_rank, *, rows, cols, row0, rows_total, T, af, coarse, steps, warmup,
            ring_slots, ring_days, exchange_ok, verify):
    """One measurement: build inputs, plan, twi all-reduce, timed passes, optional verification."""
    from microclimf_amd import synthetic
    from microclimf_amd.api import Plan
    from microclimf_amd.distributed import allreduce_max, allreduce_sum, allreduce_twi_mean
    ndays = T // 24
    if af:
        T = min(T, ring_days * ring_slots * 24)
        ndays = T // 24
    tg0 = time.perf_counter()
    a, cpos, terrain_s = build_workload(args, synthetic, rank, world, local_rank, rows, cols, row0, rows_total, T, af, coarse,
                                        exchange_ok)
    setup_s = time.perf_counter() - tg0
    plan = Plan(**a, ring_days=ring_days, ring_slots=ring_slots, device=local_rank,
                cells_per_block=args.cells_per_block, array_forcing=af, coarse=cpos)
    try:
        if af:
            for sl, d0 in enumerate(range(0, ndays, ring_days)):
                plan.upload_forcing_days(d0, min(ring_days, ndays - d0), sl)
        # the solver's one global reduction: mean of log(twi)/tfact over the WHOLE raster
        s, n = plan.twi_partial()
        twi_mean = allreduce_twi_mean(s, float(n))      # one 2-double all-reduce (RCCL over xGMI)
        plan.set_twi_mean(twi_mean)
        valid = plan.valid_cells
        clock = Clock(use_dist, dist, torch)
        dt, kms, klaunches, resident = timed_year(plan, ndays, ring_days, ring_slots, steps, warmup, clock)
        dt = allreduce_max(dt)                   # slowest rank
        valid_all = allreduce_sum(float(valid))  # whole-job units
        res = {"dt": dt, "kms": kms, "klaunches": klaunches, "valid": valid, "valid_all": valid_all, "ndays": ndays, "T": T,
               "terrain_s": terrain_s, "setup_s": setup_s, "value": valid_all * ndays * 24 * steps / dt, "verified": None,
               "plan_bytes": plan.device_bytes, "dispatch": plan.dispatch_stats()}
        if verify and rank == 0:
            res["verified"] = verify_sample(plan, a, resident, twi_mean, args.verify_cells, af=af, cpos=cpos)
    finally:
        plan.close()
    return res


def secondary_block(args, torch, dist, local_rank):
    """Short measurements of the other geometries on this GPU (single rank): configs[1], array forcing, coarse array
    forcing.  Same kernels, same accounting; each entry names its workload."""
    out = {}
    sub = argparse.Namespace(**vars(args))
    sub.terrain = "random"
    jobs = [("configs[1]", dict(rows=1024, cols=1024, af=False, coarse=None, steps=3, ring=(2, 10))),
            # (one 12-wave workgroup per CU: nothing hides a tile's prologue, so the launches are 10 days long: +3.5 % over 5)
            ("array_forcing", dict(rows=1024, cols=1024, af=True, coarse=None, steps=5, ring=(2, 10))),
            ("coarse_forcing_8x8", dict(rows=1024, cols=1024, af=False, coarse=(8, 8), steps=3, ring=(2, 10)))]
    for name, j in jobs:
        if name == "configs[1]" and args.config == 1 and (args.rows, args.cols) == (1024, 1024):
            continue      # that is the primary line
        try:
            slots, days = j["ring"]
            r = measure(sub, torch, dist, False, 0, 1, local_rank, rows=j["rows"], cols=j["cols"], row0=0, rows_total=j["rows"],
                        T=8760, af=j["af"], coarse=j["coarse"], steps=j["steps"], warmup=1, ring_slots=slots, ring_days=days,
                        exchange_ok=True, verify=not args.no_verify)
            steps_per_launch = (r["ndays"] * 24 * j["steps"]) / max(r["klaunches"], 1)
            avg_ms = r["kms"] / max(r["klaunches"], 1)
            rb = roofline_block(r["valid"], r["T"], steps_per_launch, avg_ms, r["klaunches"], j["af"], j["rows"], j["cols"], days,
                                r["value"], j["coarse"])
            what = ("vector forcing (runmicro1Cpp geometry), random terrain inputs [BASELINE.json configs[1]]" if name == "configs[1]"
                    else f"array forcing (runmicro2Cpp geometry), {r['T']} steps of forcing resident in HBM" if j["af"]
                    else "coarse array forcing (8x8 climate grid interpolated in the solver, `.runmodel2Cpp` geometry), whole year")
            out[name] = {"workload": f"{j['rows']}x{j['cols']} cells x {r['T']} hourly steps, {what}", "value": r["value"],
                         "unit": "cell-steps/s", "steps": j["steps"], "ms_per_step": r["dt"] / j["steps"] * 1e3,
                         "hbm_frac": rb["frac"], "achieved_GBps": rb["achieved"], "avg_launch_ms": avg_ms,
                         "bytes_per_cell_step": 208.0 if j["af"] else 80.0 + 440.0 / r["T"]}
            if rb.get("valu"):
                out[name]["valu"] = rb["valu"]
            if rb.get("traffic") is not None:
                out[name]["traffic"] = rb["traffic"]
                out[name]["algorithmic_bytes_per_launch"] = rb["algorithmic_bytes_per_launch"]
                out[name]["traffic_basis"] = rb["traffic_basis"]
            if rb.get("counters"):
                out[name]["counters"] = rb["counters"]
            if r["verified"]:
                out[name]["verified"] = r["verified"]
        except Exception as e:    # a secondary figure never takes the primary line down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stub:
        return stub_main(args, world, rank)
    if args.config == 4:
        try:
            from tools.bench_snow import run_snow_config      # kept out of the default path
        except ImportError:
            raise SystemExit("bench.py --config 4 (solver + snow branch) has no bench mode yet: tools/snow_rate.py measures "
                             "the snow kernels on one GPU")
        return run_snow_config(args, world, rank, local_rank)
    cpu_first = None
    if rank == 0 and not args.no_cpu_baseline:
        # Rank 0 (a fresh process under torch.distributed.run too) times the CPU legs before torch / HIP are touched — the
        # all-cores leg forks worker processes; the other ranks wait for it at the process group's first barrier.  With
        # N > 1 ranks on the box the legs share the host with the other ranks' start-up: a reported baseline, not a target.
        sys.stdout.flush()
        _saved = os.dup(1)
        os.dup2(2, 1)
        cpu_first = (cpu_baseline(args), cpu_baseline_all_cores(args))
        os.dup2(_saved, 1)
        os.close(_saved)
    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints its library path at
    # init) write to fd 1 directly, so fd 1 is pointed at stderr until the line is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MCF_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the ranks share a device,
    # the collectives run over gloo on host tensors); the driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("MCF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # MCF_BENCH_FORCE_DIST=1 exercises the RCCL path (init, all-reduce, barrier) on a single rank
    use_dist = world > 1 or os.environ.get("MCF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_library()
    if use_dist:
        dist.barrier()
    from microclimf_amd.distributed import row_block

    rows, cols, T = args.rows, args.cols, args.tsteps
    row0, rows_total = rank * rows, rows * world
    nblocks = world
    if args.scaling == "strong":
        rows_total = args.rows
        nblocks = max(world, args.share or 0)
        row0, rows = row_block(rank, nblocks, rows_total)
    af = args.array_forcing
    coarse = tuple(int(v) for v in args.coarse.split("x")) if args.coarse else None
    ring_slots, ring_days = fit_ring(args, rows * cols, af)
    r = measure(args, torch, dist, use_dist, rank, world, local_rank, rows=rows, cols=cols, row0=row0, rows_total=rows_total, T=T,
                af=af, coarse=coarse, steps=args.steps, warmup=args.warmup, ring_slots=ring_slots, ring_days=ring_days,
                exchange_ok=(nblocks == world), verify=not args.no_verify)
    rc = 0
    if rank == 0:
        T = r["T"]
        steps_per_launch = (r["ndays"] * 24 * args.steps) / max(r["klaunches"], 1)
        avg_ms = r["kms"] / max(r["klaunches"], 1)
        geometry = (f"coarse array forcing ({coarse[0]}x{coarse[1]} climate grid interpolated in the solver, `.runmodel2Cpp` "
                    "geometry), whole series resident, " if coarse else
                    "array forcing (runmicro2Cpp geometry), forcing resident in HBM, " if af else
                    "vector forcing (runmicro1Cpp geometry), ")
        raster = (f"{rows}x{cols} synthetic DTM per GPU" if args.scaling == "weak" else
                  f"{rows_total}x{cols} synthetic DTM in {nblocks} row blocks, {world} of them solved by {world} GPU(s)"
                  if nblocks != world else f"{rows_total}x{cols} synthetic DTM over {world} GPU(s)")
        tag = "" if (af or coarse or T != 8760) else f" [BASELINE.json configs[{args.config}]]"
        line = {
            "metric": "cell-steps/s", "value": r["value"], "unit": "cell-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["dt"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{raster}, {T} hourly steps, {geometry}reqhgt={args.reqhgt}, no snow, "
                            + ("terrain inputs pre-computed on the device from the DTM" if args.terrain == "device"
                               else "random terrain inputs (SURVEY 8d)") + tag,
                "baseline_config": args.config, "rows_per_gpu": rows, "cols": cols, "tsteps": T, "outputs": 10,
                "valid_cells": int(r["valid_all"]),
                "sink": f"HBM ring ({ring_slots} slots x {ring_days} days), no D2H",
                "partition": "row blocks, one per GPU; all-reduce of twi (sum,count); terrain halo rows point-to-point"
                             + (" [this line is ONE GPU; the N > 1 path — RCCL all-reduce, halo send / recv — is UNMEASURED ON N > 1 HARDWARE: "
                                "no multi-GPU node has been available to any round; it is rehearsed over gloo, tests/test_bench_gpu.py]"
                                if world == 1 else "")
                             + ("" if os.environ.get("MCF_BENCH_BACKEND", "nccl") == "nccl" else
                                " [REHEARSAL: backend " + os.environ["MCF_BENCH_BACKEND"] + ", ranks share a GPU — not a scaling measurement]"),
                "terrain": ("on-device pre-compute from the synthetic DTM (untimed, see terrain_precompute_s)"
                            if r["terrain_s"] is not None else "random (SURVEY 8d config 2)"),
                "device_bytes": int(r["plan_bytes"]),
                "dispatch": r["dispatch"],
            },
            "terrain_precompute_s": r["terrain_s"],
            "input_setup_s": r["setup_s"],
            "roofline": roofline_block(r["valid"], T, steps_per_launch, avg_ms, r["klaunches"], af, rows, cols, ring_days,
                                       r["value"] / world, coarse),
            "verified": r["verified"],
        }
        if nblocks != world:
            line["config"]["share"] = f"block(s) 0..{world - 1} of a {nblocks}-block partition; halo rows generated, not exchanged"
        if r["verified"] is not None and not r["verified"]["ok"]:
            rc = 3
        if cpu_first is not None:
            line["cpu_baseline"], line["cpu_baseline_all_cores"] = cpu_first
        if world == 1 and not args.no_secondary and not af and not coarse:
            line["secondary"] = secondary_block(args, torch, dist, local_rank)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
        if rc:
            print("bench.py: the timed run's output does NOT match the oracle: " + json.dumps(r["verified"]), file=sys.stderr)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
