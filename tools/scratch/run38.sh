mkdir -p gpurun_out/r02msk
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh gpurun_out/r02msk 'k_msk<' -- python bench.py --steps 2 --warmup 1 --single-chain --no-cpu-baseline --parity-channels 0
