#!/usr/bin/env python3
"""bench.py — cell-steps/s of the grid microclimate solver on MI355X.

One "step" = one pass of the hot path (runmicro1Cpp geometry: vector forcing,
reqhgt = 0.05 m, all 10 outputs) over the whole workload of a rank:
BASELINE.json configs[1], a 1024 x 1024 synthetic DTM x 8760 hourly steps.
Inputs are resident in HBM before the timed region; outputs go to a device ring
(sink: HBM ring, no D2H) because one year of outputs (735 GB at 1024^2) does not
fit HBM.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
the raster is row-tiled: every rank owns a 1024-row block of a (1024*N) x 1024
raster (weak scaling); the only data-path collective is the all-reduce of the
(sum, count) of log(twi)/tfact (src/microclimfCpp.cpp:993-1004) over RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=1024, help="rows per GPU")
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--tsteps", type=int, default=8760)
    ap.add_argument("--reqhgt", type=float, default=0.05)
    ap.add_argument("--ring-days", type=int, default=10)
    ap.add_argument("--ring-slots", type=int, default=2)
    ap.add_argument("--cells-per-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=str, default="144x144x720")
    ap.add_argument("--terrain", choices=["random", "device"], default="random",
                    help="'device': slope/aspect/hor/svfa/wsa come from mcf_precompute_terrain run on the "
                         "synthetic DTM (BASELINE.json configs[2]); 'random': SURVEY 8d's random terrain inputs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="'weak' (default): --rows rows PER GPU; 'strong': --rows is the whole raster, dealt to the ranks "
                         "in row blocks (e.g. --rows 8192 --cols 8192 --scaling strong --gpus 8 = BASELINE.json configs[3])")
    ap.add_argument("--coarse", type=str, default="",
                    help="secondary measurement: CRxCC, e.g. 8x8 — `.runmodel2Cpp` geometry with the coarse climate / point-model "
                         "arrays interpolated inside the solver (mcf.h array_forcing == 2): the whole year is resident")
    ap.add_argument("--array-forcing", action="store_true",
                    help="secondary measurement: runmicro2Cpp geometry; ring_slots x ring_days days of forcing are "
                         "resident in HBM and solved repeatedly (a year of array forcing, 1.1 TB at 1024^2, "
                         "cannot be resident)")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (a scalar C restatement of the reference loop, kind 'port') timed on one
    host core over a bounded sample of the same synthetic workload."""
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


def main():
    args = parse()
    cpu_first = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        # both CPU legs run before torch / HIP are touched: the all-cores leg forks worker processes
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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # MCF_BENCH_FORCE_DIST=1 exercises the RCCL path (init, all-reduce, barrier) on a single rank
    use_dist = world > 1 or os.environ.get("MCF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_library()
    if use_dist:
        dist.barrier()
    from microclimf_amd import synthetic
    from microclimf_amd.api import Plan
    from microclimf_amd.distributed import allreduce_max, allreduce_sum, allreduce_twi_mean

    rows, cols, T = args.rows, args.cols, args.tsteps
    row0, rows_total = rank * rows, rows * world
    if args.scaling == "strong":
        from microclimf_amd.distributed import row_block
        rows_total = args.rows
        row0, rows = row_block(rank, world, rows_total)
    ndays = T // 24
    af = args.array_forcing
    coarse = tuple(int(v) for v in args.coarse.split("x")) if args.coarse else None
    # the output ring (and, with array forcing, the forcing slabs) must fit the GPU: shrink the days per slot
    # until slots x days x 24 h x cells x 8 B x (10 outputs [+ 15 forcing arrays]) stays under 160 GB
    per_day = rows * cols * 24 * 8 * (10 + (15 if af else 0))
    # (longer launches first: at 4096^2 one 4-day slot runs 11 % faster than two 2-day slots — the cell tables are
    # re-read once per launch)
    while args.ring_slots * args.ring_days * per_day > 160e9 and (args.ring_slots > 1 or args.ring_days > 1):
        if args.ring_slots > 1 and not af:
            args.ring_slots -= 1
        elif args.ring_days > 1:
            args.ring_days -= 1
        else:
            break
    if af:
        T = min(T, args.ring_days * args.ring_slots * 24)
        ndays = T // 24
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
        from microclimf_amd.terrain import precompute_terrain_tiled
        _, _, dtm = synthetic.rasters(rows, cols, row0, rows_total, reqhgt=args.reqhgt)
        tt0 = time.perf_counter()
        ter = precompute_terrain_tiled(dtm, 1.0, a["zref"], rank, world, row0, rows_total,
                                       device=local_rank)
        terrain_s = time.perf_counter() - tt0
        a["soilc"].update(ter)
    n_out = 10
    plan = Plan(**a, ring_days=args.ring_days, ring_slots=args.ring_slots, device=local_rank,
                cells_per_block=args.cells_per_block, array_forcing=af, coarse=cpos)
    if af:
        for sl, d0 in enumerate(range(0, ndays, args.ring_days)):
            plan.upload_forcing_days(d0, min(args.ring_days, ndays - d0), sl)
    # the solver's one global reduction: mean of log(twi)/tfact over the WHOLE raster
    s, n = plan.twi_partial()
    plan.set_twi_mean(allreduce_twi_mean(s, float(n)))      # one 2-double all-reduce (RCCL over xGMI)
    valid = plan.valid_cells

    def one_step():
        slot = 0
        for d0 in range(0, ndays, args.ring_days):
            plan.run_days(d0, min(args.ring_days, ndays - d0), slot)
            slot = (slot + 1) % args.ring_slots

    def fence():
        plan.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    plan.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt = time.perf_counter() - t0
    kms, klaunches = plan.kernel_stats()
    dt = allreduce_max(dt)                   # slowest rank
    valid_all = allreduce_sum(float(valid))  # whole-job units
    cellsteps_per_step = valid_all * ndays * 24
    value = cellsteps_per_step * args.steps / dt

    if rank == 0:
        # roofline of the dominant kernel (k_solve): ALGORITHMIC bytes per launch =
        # valid cells x steps per launch x (8 B x n_out written + 440 B / T read)   [SURVEY §8d]
        steps_per_launch = (ndays * 24 * args.steps) / max(klaunches, 1)
        # array forcing reads 15 arrays per cell-step (+8 B for the mxtc pre-pass): 208 B   [SURVEY §8d]
        bytes_per_launch = valid * steps_per_launch * ((8.0 * n_out + 128.0) if af else (8.0 * n_out + 440.0 / T))
        avg_ms = kms / max(klaunches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists() and not coarse and not af:
            try:
                tj = json.loads(tf.read_text())
                if tj.get("rows") == rows and tj.get("cols") == cols and tj.get("ring_days") == args.ring_days:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # second roof ("both fractions must be reported"): fp64 VALU issue slots.  Instructions per cell-step
        # come from the committed PMC summary of this kernel (SQ_INSTS_VALU); the peak is one wave-instruction per
        # 4 cycles per SIMD at the 2.4 GHz nominal clock (256 CUs x 4 SIMDs)
        valu = None
        try:
            ptag = json.loads(tf.read_text()).get("tag") if tf.exists() else None
        except Exception:
            ptag = None
        pf = ROOT / "profiles" / f"{ptag}_pmc_summary.json"
        if ptag and pf.exists() and not coarse and not af:
            try:
                pj = json.loads(pf.read_text())
                per_cs = pj["per_launch_mean"]["SQ_INSTS_VALU"] * 64.0 / pj.get("cell_steps_per_launch", 1038103 * 120)
                valu = {"insts_per_cell_step": per_cs, "frac_of_issue_peak": per_cs * (value / world) / 64.0 / (1024 * 2.4e9 / 4),
                        "busy_fraction_measured": pj.get("valu_busy_fraction"), "source": f"profiles/{ptag}_pmc_summary.json"}
            except Exception:
                valu = None
        line = {
            "metric": "cell-steps/s", "value": value, "unit": "cell-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ((f"{rows}x{cols} synthetic DTM per GPU" if args.scaling == "weak" else
                              f"{rows_total}x{cols} synthetic DTM over {world} GPU(s)") + f", {T} hourly steps, "
                             + (f"coarse array forcing ({coarse[0]}x{coarse[1]} climate grid interpolated in the solver, "
                                "`.runmodel2Cpp` geometry), whole series resident, " if coarse else
                                "array forcing (runmicro2Cpp geometry), forcing resident in HBM, "
                                if af else "vector forcing (runmicro1Cpp geometry), ")
                             + f"reqhgt={args.reqhgt}, no snow"
                             + ("" if (af or coarse) else (" [BASELINE.json configs[1]]" if (rows, cols) == (1024, 1024) else
                                               " [BASELINE.json configs[2]]" if (rows, cols) == (4096, 4096)
                                               else ""))),
                "rows_per_gpu": rows, "cols": cols, "tsteps": T, "outputs": n_out,
                "valid_cells": int(valid_all),
                "sink": f"HBM ring ({args.ring_slots} slots x {args.ring_days} days), no D2H",
                "partition": "row blocks, one per GPU; all-reduce of twi (sum,count) only",
                "terrain": ("on-device pre-compute from the synthetic DTM, %.2f s incl. H2D/D2H (untimed)" % terrain_s
                            if terrain_s is not None else "random (SURVEY 8d config 2)"),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "k_solve", "avg_launch_ms": avg_ms, "launches": int(klaunches),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu": valu,
                         "note": "fp64 VALU (software transcendentals) is the binding roof, see DESIGN.md"},
        }
        if cpu_first is not None:
            line["cpu_baseline"], line["cpu_baseline_all_cores"] = cpu_first
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    plan.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
