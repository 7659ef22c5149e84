#!/bin/bash
# Quick dynamic instruction count of k_solve (one SQ pass): usage tools/pmc_valu.sh <outdir> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- \
  python3 bench.py "$@" --tsteps 1920 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify > $out/sq.json 2> $out/sq.err
python3 - <<P
import csv, glob, json, collections
d = json.loads(open("$out/sq.json").read().strip().splitlines()[-1])
valid = d["config"]["valid_cells"]
acc = collections.defaultdict(list)
for f in glob.glob("$out/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = len(acc["SQ_INSTS_VALU"])
cs = valid * 1920 / n
m = {k: sum(v) / len(v) for k, v in acc.items()}
cyc = m["GRBM_GUI_ACTIVE"] / 8
print("launches %d  VALU/cell-step %.1f  SALU %.1f  LDS %.1f  valu_busy %.3f  wait/wave-cycles %.3f" % (
    n, m["SQ_INSTS_VALU"] * 64 / cs, m["SQ_INSTS_SALU"] * 64 / cs, m["SQ_INSTS_LDS"] * 64 / cs,
    m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
P
