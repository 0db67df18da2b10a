# round 6, N: the front-end LDS claim against the channel count (tools/claim_sweep.py); the tightened chain gates
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
timeout 1500 python tools/claim_sweep.py --claims 0,16,24,28,32,40,48,56,63,72 --steps 20 --reps 2 > $O/sweep.jsonl 2> $O/sweep.err
python - <<'PY'
import json
for ln in open('gpurun_out/r6n/sweep.jsonl'):
    d=json.loads(ln); print(d['nchan'], d['chosen_claim_bytes'], d['chosen_over_best'], d['ms_per_step'])
PY
tail -3 $O/sweep.err
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_stages.py -x -q -m gpu -k "config3 or config4_8192 or any_vector_length or config4" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
