#!/bin/bash
# Diagnostic counter pass (one mixed pass per variant; the judged numbers come from profile_round.sh's separate passes):
# usage tools/pmc_quick.sh <outdir> name="bench args" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%=*}; args=${spec#*=}
  rocprofv3 --kernel-trace --output-format csv -d $out/$name --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- \
    python3 bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify > $out/$name.json 2> $out/$name.err
  for c in FETCH_SIZE WRITE_SIZE; do      # one counter per pass: the two together stalled the run on this pool
    rocprofv3 --kernel-trace --output-format csv -d $out/${name}_mem --pmc $c -- \
      python3 bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify > /dev/null 2> $out/${name}_mem_$c.err
    echo "$name $c done"
  done
  python3 - <<P
import csv, glob, collections
for sub in ("$name", "${name}_mem"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not k.startswith("void mcf::k_solve<") and not k.startswith("mcf::k_solve<"):
                if "k_solve<" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print("$name", k[:60], {c: "%.4g" % (v / n[(k, c)]) for c, v in d.items()})
P
done
