set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03c
mkdir -p $out
rocprofv3 --list-avail > $out/list_avail.txt 2>&1 || true
B="python3 bench.py --config 1 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify"
pass() {   # name lib counters...
  local name=$1 lib=$2; shift 2
  ( [ "$lib" != "-" ] && export MCF_LIB=$PWD/$lib; rocprofv3 --kernel-trace --output-format csv -d $out/$name --pmc "$@" -- $B > $out/$name.json 2> $out/$name.err ) || echo "$name failed"
  echo "$name done"
}
for v in new:- r02:build/variants/libmcfhip_r02.so; do
  n=${v%%:*}; l=${v#*:}
  pass ${n}_sq $l SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
  pass ${n}_wr $l WRITE_SIZE
  pass ${n}_rd $l FETCH_SIZE
  pass ${n}_ea $l TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
  pass ${n}_tlb $l TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
done
python3 - <<P
import csv, glob, collections
for sub in sorted(glob.glob("$out/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_solve<" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(sub.split("/")[-2], k[:48], {c: "%.5g" % (v / n[(k, c)]) for c, v in d.items()}, "launches", max(n.values()))
P
