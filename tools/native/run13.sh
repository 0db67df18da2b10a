mkdir -p gpurun_out/r2m
for pad in 0 20 40 60 76; do
  AISX_MSK_LDS_PAD=$pad python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2m/pad$pad.json
done
for pad in 0 40; do
  AISX_CORR_DMA=0 AISX_MSK_LDS_PAD=$pad python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2m/nodma_pad$pad.json
done
