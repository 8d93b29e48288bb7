"""CPU: the N>1 path (batch sharding + all-gather of the output audio) with world_size 2 over gloo.
The per-rank sampler is a stand-in callable (the HIP sampler needs a GPU); what is covered is the sharding,
ragged shards, rank-order reassembly and independence of the result from the number of ranks."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from audio_diffusion_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(model, x, steps, eta):
    if x.shape[0] == 0:                          # like the real sampler (check_geo: "bad geometry B=0")
        raise ValueError("empty batch")
    return torch.tanh(x * 3.0) + steps          # any per-sample function


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    D.init(rank, world, backend="gloo")
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(total, 2, 64, generator=g)      # generated once from one seed on every rank
    out = D.sample_sharded(None, noise, 5, 0, sampler=_fake_sampler)
    # generic form: output with another trailing shape, a per-sample side input, empty shards
    side = torch.arange(total, dtype=torch.float32)
    up = D.run_sharded(lambda x, s: x.repeat_interleave(3, dim=2)[:, :1] * s[:, None, None], noise, side)
    assert torch.equal(up, noise.repeat_interleave(3, dim=2)[:, :1] * side[:, None, None])
    mx = D.max_over_ranks(float(rank + 1), world, torch.device("cpu"))
    D.barrier(world)
    if rank == 0:
        q.put((out, mx))
    D.barrier(world)
    D.shutdown()


@pytest.mark.parametrize("total", [8, 7, 1])
def test_sharded_sampling_world2(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out, mx = q.get(timeout=120)            # a failing worker must not hang the suite
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(total, 2, 64, generator=g)
    assert torch.equal(out, _fake_sampler(None, noise, 5, 0))       # same as the 1-rank result, rank order kept
    assert mx == 2.0


def test_bench_launcher_world2_gloo():
    """`python bench.py --gpus 2` exactly as the driver calls it (no launcher around it): the parent starts the two
    ranks itself, they rendezvous on 127.0.0.1 (gloo here, RCCL on GPUs), shard configs[4]'s global batch, all-gather
    inside the timed region, and rank 0 prints ONE JSON line.  --standin replaces only the HIP sampler."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--standin", "--global-batch", "7"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["backend"] == "gloo"
    assert rec["scaling"] == "strong" and rec["config"]["global_batch"] == 7 and rec["config"]["batch_per_gpu"] == 4
    assert rec["steps"] == 2 and rec["warmup"] == 1 and rec["value"] > 0


def test_shard_bounds_cover_everything():
    for total in (1, 7, 16, 256):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
