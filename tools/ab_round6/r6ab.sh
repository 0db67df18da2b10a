# round 6, AB: the resolver's regions against the one-wave scan (-DRSV_WAVES_MAX=1) on the device, tag for tag, over channel counts (16 / 8 / 4 /
# 2 waves per channel), call lengths that are not whole blocks, template lengths and samples per symbol
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ab; mkdir -p $O
B=tools/native/corrbench
P=gr-ais_amd/lib/libaisx.so
R=tools/scratch/libaisx_rsv1.so
{
for args in "--nchan 1 --n 65536" "--nchan 7 --n 50001" "--nchan 300 --n 40000 --N 513" "--nchan 512 --n 65536 --N 112" "--nchan 700 --n 12345 --N 112 --sps 5.2" \
            "--nchan 1024 --n 65536 --N 20 --sps 2" "--nchan 1500 --n 30000" "--nchan 2048 --n 65536 --N 112" "--nchan 3000 --n 65536 --sps 8" "--nchan 4096 --n 65536" \
            "--nchan 64 --n 1048576 --N 896" "--nchan 16 --n 4000" "--nchan 5000 --n 20000 --N 112"; do
  echo "== $args"
  timeout 120 $B $P --ref $R --iters 2 $args 2>&1 | grep -E "compare|error|rror" 
done
} > $O/log.txt 2>&1
cat $O/log.txt
