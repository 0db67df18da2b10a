# round 6, AC: the three-regime claim rule (23 KB between half and all of the CUs) against the sweep
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ac; mkdir -p $O
timeout 900 python tools/claim_sweep.py --nchan 4608,5120,6144,7168,8192,9216,12288 --claims 0,16,23,32,46,56 --steps 20 --reps 3 > $O/sweep.jsonl 2> $O/sweep.err
python - <<'PY'
import json
for ln in open('gpurun_out/r6ac/sweep.jsonl'):
    d=json.loads(ln); print(d['nchan'], d['chosen_claim_bytes'], d['chosen_over_best'], d['ms_per_step'])
PY
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "claim" -s 2>&1 | grep -E "claim sweep|passed|failed"
