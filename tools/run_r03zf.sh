set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03zf; mkdir -p $out
for d in 5 10; do
python3 bench.py --config 1 --array-forcing --ring-days $d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $out/af_$d.json 2>> $out/err.txt
python3 bench.py --config 1 --coarse 8x8 --ring-days $d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $out/coarse_$d.json 2>> $out/err.txt
done
python3 -c "
import json
for f in ['af_5','af_10','coarse_5','coarse_10']:
    d=json.load(open('$out/'+f+'.json')); print(f, '%.4e'%d['value'], d['roofline']['avg_launch_ms'], d['config'].get('sink'), d['config'].get('tsteps'))"
