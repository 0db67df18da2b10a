# round 6, O: where the claim's cliffs are (4096 and 8192 channels, 2 KB steps)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
timeout 900 python tools/claim_sweep.py --nchan 4096,8192,5120 --claims 44,48,50,52,54,56,58,60,61,62,63 --steps 20 --reps 2 > $O/sweep.jsonl 2> $O/sweep.err
python - <<'PY'
import json
for ln in open('gpurun_out/r6o/sweep.jsonl'):
    d=json.loads(ln); print(d['nchan'], d['chosen_claim_bytes'], d['chosen_over_best'], d['ms_per_step'])
PY
tail -3 $O/sweep.err
