# round 6, K: the shader clock during long runs of the 4f build and its CE_DBG variants (is "compute only" faster because it
# computes on whatever the LDS held -- less switching power, higher clock?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
B=tools/native/corrbench
{
for v in f fdbg8 fdbg9 fdbg10 fdbg11 fdbg88; do
    ( sleep 1.0; for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i -E "sclk|Power \(W\)" | sed 's/.*sclk clock level: [0-9]*: //; s/.*Power (W): / W=/' | tr '\n' ' '; echo; sleep 0.2; done ) &
    timeout 120 $B tools/scratch/libaisx_$v.so --iters 3000 2>&1 | sed 's/tags.*//'
    wait
done
} > $O/log.txt 2>&1
cat $O/log.txt
