mkdir -p gpurun_out/r3a
for i in 1 2; do
AISX_CORR_DMA=0 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r3a/nodma_$i.json
python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r3a/dma_$i.json
done
