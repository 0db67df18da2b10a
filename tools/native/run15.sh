mkdir -p gpurun_out/r2o
for rep in 1 2; do
for v in dma1 b0u0 b0u1 b1u0 b1u1; do
./tools/native/corrbench exp/libaisx_$v.so --iters 20 >> gpurun_out/r2o/corr.txt 2>&1
done
done
cat gpurun_out/r2o/corr.txt
