"""Batch-parallel multi-GPU sampling: one process per GPU, independent samples sharded across ranks,
ONE collective at the end (all-gather of the output audio over RCCL/xGMI, or gloo on CPU for tests).

The sampling path itself has no exchange step (GroupNorm(1,C) and attention are per-sample; eta=0 draws
no RNG), so nothing is communicated inside the step loop.  Reference: sampling is single-process there
(train_uncond.py:165 ``@rank_zero_only``); this is the MI355X scale-out of BASELINE.json configs[4].
"""
import os

import torch
import torch.distributed as dist


def init(rank, world, backend=None):
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    dist.init_process_group(backend, rank=rank, world_size=world)


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def barrier(world):
    if world > 1 and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, world, device):
    if world <= 1 or not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_bounds(total, rank, world):
    """Contiguous, balanced split of `total` samples; the first total % world ranks get one more."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_audio(local):
    """[b_r, C, T] per rank -> [sum b_r, C, T] on every rank (rank order).  Equal shards use one
    all_gather_into_tensor (a single ring pass over xGMI); ragged shards pad to the largest."""
    world = dist.get_world_size()
    if dist.get_backend() != "nccl" and local.is_cuda:      # gloo rehearsal on a GPU box: gather through host memory
        return all_gather_audio(local.cpu()).to(local.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    dist.all_gather(sizes, mine)
    counts = [int(s.item()) for s in sizes]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    big = max(counts)
    padded = torch.zeros((big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    return torch.cat([b[:n] for b, n in zip(bufs, counts)], dim=0)


@torch.no_grad()
def sample_sharded(model, noise, steps, eta, sampler):
    """Every rank passes the SAME global `noise` [B,C,T] (generated once from one seed); each rank runs
    its contiguous shard through `sampler(model, x, steps, eta)` and the full batch is all-gathered.  A rank whose
    shard is empty (more ranks than samples) never calls the sampler (it rejects B = 0, like the reference's convs)."""
    return run_sharded(lambda x: sampler(model, x, steps, eta), noise)


@torch.no_grad()
def run_sharded(fn, batch, *per_sample):
    """Generic form for the other per-sample entry points (``lambda x: sample_heun(denoiser, x, sigmas)``, a
    SoundStream decoder, a conditioned sampler with ``per_sample`` = (cond,)): every rank passes the SAME global
    tensors; rank r runs ``fn(batch[lo:hi], *[a[lo:hi] for a in per_sample])`` on its contiguous shard and the
    results are all-gathered in rank order.  The output may have any trailing shape; a rank whose shard is empty
    (more ranks than samples) skips ``fn`` and learns the trailing shape from rank 0."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    local = fn(batch[lo:hi], *[a[lo:hi] for a in per_sample]) if hi > lo else None
    meta = [None if local is None else (tuple(local.shape[1:]), local.dtype)]
    dist.broadcast_object_list(meta, src=0)              # rank 0 always owns at least one sample
    if local is None:
        local = torch.empty((0,) + meta[0][0], dtype=meta[0][1], device=batch.device)
    return all_gather_audio(local)
