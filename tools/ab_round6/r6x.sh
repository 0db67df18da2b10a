# round 6, X: cache policy bits of k_corr4f's window loads (l*) and pass-through stores (s*) under the power limit: time, power, clock
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x; mkdir -p $O
python tools/corr_energy.py $O f,pol_l0,pol_l1,pol_l16,pol_l18,pol_s0,pol_s16,pol_s18,pol_l0s0,f > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-200
