mkdir -p gpurun_out/r2p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/calib/p0 -- ./tools/ubench/traffic_calib > gpurun_out/r2p/calib0.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2p/calib/p1 -- ./tools/ubench/traffic_calib > gpurun_out/r2p/calib1.log 2>&1
python3 tools/pmc_table.py gpurun_out/r2p/calib > gpurun_out/r2p/calib.json
# the correlator kernels of the product build, alone
bash tools/pmc_passes.sh gpurun_out/r2p/corr "k_corr4" -- ./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 5
python3 tools/pmc_table.py gpurun_out/r2p/corr > gpurun_out/r2p/corr.json
find gpurun_out/r2p -name "*agent_info*" -delete
cat gpurun_out/r2p/calib.json
