set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03r; mkdir -p $out
echo "== shipped build" > $out/snow_residue_ab.txt
timeout -k 10 500 python3 tools/snow_residue_ab.py >> $out/snow_residue_ab.txt 2> $out/a.err
echo "== mcf_snow.o built with -ffp-contract=off" >> $out/snow_residue_ab.txt
MCF_LIB=$PWD/build/variants/libmcfhip_snow_nofma.so timeout -k 10 500 python3 tools/snow_residue_ab.py >> $out/snow_residue_ab.txt 2> $out/b.err
cat $out/snow_residue_ab.txt
