# round 6, D: per-launch series of the correlator alone + clocks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
B=tools/native/corrbench
export CORRBENCH_SERIES=1
{
rocm-smi --showclocks --showpower 2>&1 | head -40
timeout 120 $B tools/scratch/libaisx_e_noemit_nols.so --iters 60
timeout 120 $B gr-ais_amd/lib/libaisx.so --iters 60
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks 2>&1 | grep -i -E "sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.05; done ) &
timeout 120 $B tools/scratch/libaisx_e_noemit_nols.so --iters 2000 
wait
} > $O/log.txt 2>&1
cat $O/log.txt
