"""Terrain pre-compute of the library in the tree against another build of it (MCF_LIB_B), bit for bit, and the time of each.
    MCF_LIB_B=build/variants/libmcfhip_prevterrain.so python tools/terrain_ab.py [rows cols]"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def run(rows, cols, out):
    from microclimf_amd.terrain import precompute_terrain
    from microclimf_amd import synthetic
    _, _, dtm = synthetic.rasters(rows, cols)
    dtm[5, 7] = np.nan
    precompute_terrain(dtm[:64, :64], 1.0, 2.0)
    t = time.perf_counter()
    r = precompute_terrain(dtm, 1.0, 2.0)
    dt = time.perf_counter() - t
    # a row block with halos too (the stencils' range tests against the supplied rows)
    rb = precompute_terrain(dtm[100 - 100:100 + 300 + 128], 1.0, 2.0, halo_north=100, halo_south=128, row0=100, rows_total=rows)
    np.savez(out, dt=dt, **r, **{"b_" + k: v for k, v in rb.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 2048)
    res = {}
    for tag, lib in (("tree", None), ("other", os.environ["MCF_LIB_B"])):
        env = dict(os.environ)
        if lib:
            env["MCF_LIB"] = os.path.abspath(lib)
        f = f"/tmp/terrain_ab_{tag}.npz"
        subprocess.run([sys.executable, __file__, "--child", str(rows), str(cols), f], check=True, env=env)
        res[tag] = np.load(f)
    a, b = res["tree"], res["other"]
    print(f"{rows} x {cols}: tree {float(a['dt']):.3f} s, other {float(b['dt']):.3f} s")
    bad = 0
    for k in a.files:
        if k == "dt":
            continue
        same = np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64))
        print(f"  {k:12s} {'bit-identical' if same else 'DIFFERS max |d| = %.3e' % np.nanmax(np.abs(a[k] - b[k]))}")
        bad += not same
    sys.exit(1 if bad else 0)
