"""Sharding the particle filter over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for
the tests).

Particles are independent during a scan (Algorithm/FastSlam.py:25-27), so the data
path has no collective.  The only exchange is the weight normaliser
(``normalizeWeights`` / ``weightUnbalanced``, Algorithm/FastSlam.py:30-48): an
all-reduce(MAX) of the local log-weight maximum followed by one all-reduce(SUM) of
the vector [sum w, sum w^2] -- 8 + 16 bytes per scan, latency-bound.  Resampling
(rare, Algorithm/FastSlam.py:50-62) all-gathers the N weights, draws the indices
from the shared seeded stream on every rank, and moves the surviving particles' maps
point-to-point.

Everything here is device-agnostic tensor plumbing: it runs on CUDA/HIP tensors with
nccl and on CPU tensors with gloo.
"""
import math

import torch
import torch.distributed as dist


def _stage(t, group=None):
    """gloo has no device-tensor point-to-point (and only some device collectives): with that backend
    (CPU tests, or two ranks sharing one GPU) device tensors make the hop through host memory."""
    if t.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo":
        return t.cpu()
    return t


def shard_range(total, world, rank):
    """Contiguous slice [first, first + count) of ``total`` particles owned by ``rank``."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def owner_of(index, total, world):
    base, extra = divmod(total, world)
    cut = extra * (base + 1)
    if index < cut:
        return index // (base + 1)
    return extra + (index - cut) // base


def normalize_sharded(logw_local, total, group=None):
    """Normalise log-weights sharded over the ranks of ``group``.

    Returns (w_local, logw_local_normalised, variance) where variance is
    sum_i (w_i - 1/N)^2 over ALL N particles (Algorithm/FastSlam.py:32-35), computed as
    sum w^2 - 1/N.  Two tiny all-reduces; no gather of the weights themselves."""
    dev = logw_local.device
    m = logw_local.max().reshape(1) if logw_local.numel() else logw_local.new_full((1,), -math.inf)
    if dist.is_initialized():
        m = _stage(m, group)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        m = m.to(dev)
    e = torch.exp(logw_local - m)
    sums = torch.stack((e.sum(), (e * e).sum()))
    if dist.is_initialized():
        sums = _stage(sums, group)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        sums = sums.to(dev)
    w = e / sums[0]
    variance = sums[1] / (sums[0] * sums[0]) - 1.0 / total
    logw = logw_local - (m + torch.log(sums[0]))
    return w, logw, variance


def gather_weights(w_local, total, world, group=None):
    """All N normalised weights on every rank, in particle order (ragged shards allowed)."""
    if not dist.is_initialized() or world == 1:
        return w_local.clone()
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    cap = max(counts)
    pad = w_local.new_zeros(cap)
    pad[:w_local.numel()] = w_local
    pad = _stage(pad, group)
    parts = [pad.new_zeros(cap) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)]).to(w_local.device)


def resample_plan(indices, total, world, rank):
    """Who sends what to whom for ``particles[i] = copy(particles[indices[i]])``.

    Returns (local_copies, sends, recvs): local_copies = [(dst_local, src_local)],
    sends = [(dst_rank, src_local, tag)], recvs = [(src_rank, dst_local, tag)];
    tag = destination global index, so matching is unambiguous."""
    first, count = shard_range(total, world, rank)
    local, sends, recvs = [], [], []
    for dst, src in enumerate(int(i) for i in indices):
        dr, sr = owner_of(dst, total, world), owner_of(src, total, world)
        if dr == rank and sr == rank:
            local.append((dst - first, src - first))
        elif sr == rank:
            sends.append((dr, src - first, dst))
        elif dr == rank:
            recvs.append((sr, dst - first, dst))
    return local, sends, recvs


def migrate(tensors, indices, total, world, rank, group=None):
    """Apply a resample to per-particle state held as a list of same-shaped tensors
    (this rank's particles, in order).  Returns the new list.  Cross-rank moves are
    point-to-point sends of whole particle states (5-32 MB maps over xGMI)."""
    local, sends, recvs = resample_plan(indices, total, world, rank)
    new = [None] * len(tensors)
    for d, s in local:
        new[d] = tensors[s].clone()
    ops, landing = [], []
    dev = tensors[0].device
    for dst_rank, s, tag in sends:
        ops.append(dist.P2POp(dist.isend, _stage(tensors[s].contiguous(), group), dst_rank, group=group, tag=tag))
    for src_rank, d, tag in recvs:
        buf = _stage(torch.empty_like(tensors[0]), group)
        landing.append((d, buf))
        ops.append(dist.P2POp(dist.irecv, buf, src_rank, group=group, tag=tag))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for d, buf in landing:
        new[d] = buf.to(dev)
    assert all(t is not None for t in new)
    return new
