mkdir -p gpurun_out/r2n
for i in 1 2; do
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --ref exp/libaisx_dma1.so --iters 20 >> gpurun_out/r2n/corr.txt 2>&1
done
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --ref exp/libaisx_dma1.so --iters 20 --nchan 256 >> gpurun_out/r2n/corr.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --ref exp/libaisx_dma1.so --iters 10 --N 1120 --sps 5 >> gpurun_out/r2n/corr.txt 2>&1
AISX_CORR_DMA=2 ./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 10 --N 513 >> gpurun_out/r2n/corr.txt 2>&1
cat gpurun_out/r2n/corr.txt
