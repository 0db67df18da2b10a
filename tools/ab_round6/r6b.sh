# round 6, B: PMC passes of k_corr4e_main<896> alone (tools/native/corrbench), and the per-launch series
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
bash tools/pmc_passes.sh $O/corr896e k_corr4e -- tools/native/corrbench tools/scratch/libaisx_e.so --iters 5
python tools/pmc_table.py $O/corr896e > $O/corr896e.txt 2>&1
cat $O/corr896e.txt
