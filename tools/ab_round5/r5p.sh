export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gr_wrappers.py tests/test_abi_cpp.py tests/test_gpu_corr_msk.py tests/test_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2 3; do ./tests/abi_cpp/gr_blocks_harness tests/golden/config1_sched.bin | grep -E "HOST_PATH|PASS|FAIL" | cut -c1-200; done
