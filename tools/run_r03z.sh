set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03za; mkdir -p $out
cat /sys/kernel/mm/transparent_hugepage/enabled > $out/thp.txt; nproc >> $out/thp.txt
for mode in "numpy" "plain" "plain_nothp"; do
  case $mode in
    numpy) env="";;
    plain) env="MCF_TEST_PLAIN_OUTPUTS=1";;
    plain_nothp) env="MCF_TEST_PLAIN_OUTPUTS=1 MCF_NO_THP=1";;
  esac
  echo "== $mode" >> $out/oneshot.txt
  env $env timeout -k 10 300 python3 tools/oneshot_rate.py 1024 1024 240 >> $out/oneshot.txt 2>&1
done
cat $out/thp.txt $out/oneshot.txt


