# round 6, Y: the exchange between the second and the third pass of k_corr4f in registers (permlane swaps + DPP, -DCE_XP2=1)
# instead of through LDS: results against the product build, time / power / clock alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y; mkdir -p $O
B=tools/native/corrbench
{
timeout 120 $B tools/scratch/libaisx_f_xp.so --ref gr-ais_amd/lib/libaisx.so --iters 5
timeout 120 $B tools/scratch/libaisx_f_xp.so --ref gr-ais_amd/lib/libaisx.so --iters 5 --N 1000
} > $O/cmp.txt 2>&1
cat $O/cmp.txt | sed 's/tags(read.*//'
python tools/corr_energy.py $O f,f_xp,f,f_xp > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-210
