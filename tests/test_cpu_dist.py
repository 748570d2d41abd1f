"""world_size-2 gloo test of the multi-GPU path's host logic: clip sharding + the one all_gather of token buffers."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_result(c):
    rng = np.random.default_rng(c)
    n = int(rng.integers(5, 60))
    return dict(tokens=[(int(t), 0, 0.0, 0.0) for t in rng.integers(0, 51865, n)], segments=[None] * int(rng.integers(1, 5)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from streamkit_amd import dist as skd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ids = skd.shard_clip_ids(16, rank, world)
    rows = skd.pack_tokens([_fake_result(c) for c in ids])
    allr = skd.gather_tokens(rows, world)
    q.put((rank, ids, {k: v for k, v in allr.items()}))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    from streamkit_amd import dist as skd
    assert skd.shard_clip_ids(8, 1, 2) == [1, 3, 5, 7] and skd.shard_clip_ids(512, 3, 8)[:3] == [3, 11, 19]
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    got = [q.get(timeout=120) for _ in range(2)]
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, ids, allr in got:
        assert sorted(allr.keys()) == list(range(16))           # every rank sees every clip's transcript
        for c in range(16):
            ref = _fake_result(c)
            assert allr[c]["ids"] == [t[0] for t in ref["tokens"]] and allr[c]["n_segments"] == len(ref["segments"])


def test_bench_launcher_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` starts its own ranks; with fewer than N devices it must fail loudly (exit code != 0, nothing printed as
    a result line) instead of timing one GPU and calling it N.  On the GPU-less build box every N > 1 is such a case."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but only" in r.stderr and "{" not in r.stdout


def test_one_rank_limit_for_everything_that_starts_ranks_itself():
    """bench.py's own launcher and the RCCL test share ONE limit on self-started rank processes (streamkit_amd/dist.py): `--gpus 8` without a launcher is refused
    with a pointer to the driver's torchrun line, before anything counts devices; under a launcher (RANK set) bench.py is one of the ranks and the limit does not apply."""
    import subprocess
    from streamkit_amd import dist as skd
    assert skd.self_started_rank_limit() == skd.SELF_STARTED_RANK_LIMIT == 6
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SKW_SELF_STARTED_RANK_LIMIT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "self-started jobs are limited to 6" in r.stderr and "torch.distributed.run" in r.stderr and "{" not in r.stdout
    src = open(os.path.join(ROOT, "tests", "test_gpu_dist.py")).read()
    assert "self_started_rank_limit()" in src and "min(R, 6)" not in src
