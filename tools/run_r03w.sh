cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03w; mkdir -p $out
V=build/variants
for v in shipped:- nostore:$V/libmcfhip_nostore.so storehot:$V/libmcfhip_storehot.so storeonly:$V/libmcfhip_storeonly.so; do
  n=${v%%:*}; l=${v#*:}
  ( [ "$l" != "-" ] && export MCF_LIB=$PWD/$l; rocprofv3 --kernel-trace --output-format csv -d $out/$n --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU -- python3 bench.py --config 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $out/$n.json 2> $out/$n.err )
  python3 - <<P
import csv, glob, collections
acc = collections.defaultdict(list); dur=[]
for f in glob.glob("$out/$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve<" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/$n/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve<" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
m={k:sum(v)/len(v) for k,v in acc.items()}; d=sum(dur)/len(dur)
cyc=m["GRBM_GUI_ACTIVE"]/8
print("$n launch %.3f ms  clock %.3f GHz  valu_busy %.3f  wait/wavecyc %.3f  wave-cycles/launch-cycles %.3f" % (d, cyc/(d*1e-3)/1e9, m["SQ_ACTIVE_INST_VALU"]*4/(1024*cyc), m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_WAVE_CYCLES"]*4/(cyc*4096)))
P
done
