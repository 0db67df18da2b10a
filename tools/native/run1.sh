mkdir -p gpurun_out/r2a
./tools/ubench/valu_tput > gpurun_out/r2a/valu_tput.txt 2>&1
./tools/ubench/dma_probe > gpurun_out/r2a/dma_probe.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 20 > gpurun_out/r2a/corr_base.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 20 --N 112 >> gpurun_out/r2a/corr_base.txt 2>&1
./tools/native/corrbench gr-ais_amd/lib/libaisx.so --iters 20 --nchan 256 >> gpurun_out/r2a/corr_base.txt 2>&1
cat gpurun_out/r2a/*.txt
