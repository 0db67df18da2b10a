"""N > 1 path: channel sharding with two `gloo` ranks on the CPU.  The per-rank
worker here is the CPU oracle (the HIP path needs a GPU); what is under test is
the sharding arithmetic, the rendezvous and the rank-0 aggregation that
bench.py --gpus N uses: every channel is processed exactly once and the union of
the shards equals the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shard_channels_partition():
    from ais_amd.shard import shard_channels

    for total in (1, 7, 4096, 65536, 100):
        for world in (1, 2, 3, 8):
            got = [shard_channels(total, world, r) for r in range(world)]
            assert sum(c for _, c in got) == total
            pos = 0
            for first, cnt in got:
                assert first == pos
                pos += cnt
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        shard_channels(10, 2, 2)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, os.path.join(ROOT, "gr-ais_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import oracle_py as orc
    import synth
    from ais_amd.shard import gather_counts, max_over_ranks, shard_channels

    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, cnt = shard_channels(total, world, rank)
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    tmpl = synth.gmsk_waveform(np.array(lv, float), 4)[: len(lv) * 4].astype(np.complex64)
    ndet = []
    for c in range(first, first + cnt):
        x, _ = synth.make_channel(5000 + c, 8192, "P", 4, amp=1.0, cfo_max=10.0)
        _, _, tags = orc.Demod(4, tmpl, stages=0).step(x)
        ndet.append(int((tags["key"] == 2).sum()))
    dist.barrier()
    allc = gather_counts(np.array(ndet))
    tmax = max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((allc.tolist(), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shards_cover_all_channels():
    import torch.multiprocessing as mp

    import oracle_py as orc
    import synth

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    allc, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    tmpl = synth.gmsk_waveform(np.array(lv, float), 4)[: len(lv) * 4].astype(np.complex64)
    want = []
    for c in range(total):
        x, _ = synth.make_channel(5000 + c, 8192, "P", 4, amp=1.0, cfo_max=10.0)
        _, _, tags = orc.Demod(4, tmpl, stages=0).step(x)
        want.append(int((tags["key"] == 2).sum()))
    assert allc == want and tmax == 2.0


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks itself
    (torch.distributed.run on 127.0.0.1) and report n_gpus = 2; --dry keeps the ranks on the CPU
    (gloo rendezvous, channel sharding, max-over-ranks), nothing is computed."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--channels-per-gpu", "8192"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry"] is True and line["scaling"] == "weak"
    assert line["config"]["channels_per_gpu"] == 8192 and line["max_over_ranks_check"] == 2e-3
    # BASELINE config 4 as one flag, four ranks: 8192 channels each, named in the workload
    out4 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry", "--config4"],
                          env=env, capture_output=True, text=True, timeout=300)
    assert out4.returncode == 0, out4.stderr[-2000:]
    l4 = json.loads([ln for ln in out4.stdout.splitlines() if ln.startswith("{")][-1])
    assert l4["n_gpus"] == 4 and l4["config"]["channels_per_gpu"] == 8192 and "config 4" in l4["config"]["workload"]
    assert l4["max_over_ranks_check"] == 4e-3 and l4["rank_ms_per_step"] == {"min": 1.0, "max": 4.0}
    # one rank, no launcher
    out1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry"], env=env, capture_output=True, text=True,
                          timeout=300)
    assert out1.returncode == 0 and json.loads(out1.stdout.strip().splitlines()[-1])["n_gpus"] == 1
