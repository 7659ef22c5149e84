# final profile set of round 3 (run through gpurun)
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03b || echo "r03b failed"
bash tools/profile_round.sh r03b_c1 --config 1 || echo "r03b_c1 failed"
bash tools/profile_round.sh r03b_af --config 1 --array-forcing --ring-days 5 || echo "af failed"
bash tools/profile_round.sh r03b_coarse --config 1 --coarse 8x8 --ring-days 5 || echo "coarse failed"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_wr; mkdir -p $out
B="python3 bench.py --tsteps 1920 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-verify"
for v in shipped:- r02:build/variants/libmcfhip_r02.so; do
  n=${v%%:*}; l=${v#*:}
  ( [ "$l" != "-" ] && export MCF_LIB=$PWD/$l; rocprofv3 --kernel-trace --output-format csv -d $out/${n}_ea --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -- $B > $out/${n}_ea.json 2> $out/${n}_ea.err ) || echo "$n ea failed"
  ( [ "$l" != "-" ] && export MCF_LIB=$PWD/$l; rocprofv3 --kernel-trace --output-format csv -d $out/${n}_tlb --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -- $B > $out/${n}_tlb.json 2> $out/${n}_tlb.err ) || echo "$n tlb failed"
  echo "$n done"
done
python3 - <<P > $out/summary.txt
import csv, glob, collections
print("write path of k_solve at BASELINE configs[2] (4096^2, device terrain, 7-day launches, --tsteps 1920): per-launch means, rocprofv3 --pmc, one family per pass")
for sub in sorted(glob.glob("$out/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_solve<" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(sub.split("/")[-2], {c: "%.5g" % (sum(v) / len(v)) for c, v in acc.items()}, "launches", max(len(v) for v in acc.values()) if acc else 0)
P
cat $out/summary.txt
bash tools/profile_aux.sh r03b || echo "aux failed"
