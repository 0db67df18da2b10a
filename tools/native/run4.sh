mkdir -p gpurun_out/r2d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh gpurun_out/r2d/dma "k_corr4" -- ./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 5
AISX_CORR_DMA=0 bash tools/pmc_passes.sh gpurun_out/r2d/old "k_corr4" -- ./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 5
python3 tools/pmc_table.py gpurun_out/r2d/dma > gpurun_out/r2d/dma.json
python3 tools/pmc_table.py gpurun_out/r2d/old > gpurun_out/r2d/old.json
tail -3 gpurun_out/r2d/dma.p0.log
find gpurun_out/r2d -name "*.csv" | wc -l
rm -rf gpurun_out/r2d/*/p*/*/*agent_info* 
