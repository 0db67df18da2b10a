# round 6, U: the resolver with up to sixteen waves per channel (by channel count) against one wave per channel: call wall time of
# aisx_corr_process (main kernel + resolver) at 64 .. 8192 channels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6u; mkdir -p $O
B=tools/native/corrbench
{
for n in 64 256 1024 2048 4096 8192; do
  for v in exp rsv1 exp rsv1; do
    if [ $v = exp ]; then L=gr-ais_amd/lib/libaisx_exp.so; else L=tools/scratch/libaisx_$v.so; fi
    echo -n "nchan $n $v: "; timeout 120 $B $L --iters 200 --nchan $n 2>&1 | sed 's/tags.*//; s/.*main kernel/main kernel/'
  done
done
} > $O/log.txt 2>&1
cat $O/log.txt
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "claim" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep "claim sweep" $O/pytest.log
