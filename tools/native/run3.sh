mkdir -p gpurun_out/r2c
./tools/ubench/dma_probe > gpurun_out/r2c/dma_probe.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --ref same --iters 20 > gpurun_out/r2c/corr.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --ref same --iters 20 --nchan 256 >> gpurun_out/r2c/corr.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --ref same --iters 10 --N 1120 --sps 5 >> gpurun_out/r2c/corr.txt 2>&1
AISX_CORR_DMA=2 ./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 10 >> gpurun_out/r2c/corr.txt 2>&1
cat gpurun_out/r2c/*.txt
