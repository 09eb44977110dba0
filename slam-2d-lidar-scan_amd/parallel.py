"""Sharding the particle filter over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for
the tests).

Particles are independent during a scan (Algorithm/FastSlam.py:25-27), so the data
path has no collective.  The only exchange is the weight normaliser
(``normalizeWeights`` / ``weightUnbalanced``, Algorithm/FastSlam.py:30-48): one
all-gather of [max log-weight, sum w, sum w^2] per rank (24 bytes), merged in rank order
on every rank -- latency-bound, one collective per scan.  Resampling
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
    sum w^2 - 1/N.  ONE collective per scan: an all-gather of each rank's
    [max log-weight, sum exp(lw - max), sum exp(2 (lw - max))] (24 bytes per rank), merged
    identically on every rank in rank order -- so the result does not depend on the
    reduction order of the network, and the per-scan cost is one xGMI latency."""
    dev = logw_local.device
    m = logw_local.max() if logw_local.numel() else logw_local.new_tensor(-math.inf)
    e = torch.exp(logw_local - m)
    mine = torch.stack((m, e.sum(), (e * e).sum()))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        mine = _stage(mine, group)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        allp = torch.stack(parts).to(dev)                       # [world, 3]
        gm = allp[:, 0].max()
        scale = torch.exp(allp[:, 0] - gm)                      # rescale every rank's sums to the global max
        s1 = (allp[:, 1] * scale).sum()
        s2 = (allp[:, 2] * scale * scale).sum()
    else:
        gm, s1, s2 = m, mine[1].to(dev), mine[2].to(dev)
    w = torch.exp(logw_local - gm) / s1
    variance = s2 / (s1 * s1) - 1.0 / total
    logw = logw_local - (gm + torch.log(s1))
    return w, logw, variance


class DirectRccl:
    """The normaliser's all-gather as ONE ``ncclAllGather`` call on the caller's own HIP stream, straight from librccl -- an
    OPTION (SLAM2D_DIRECT_RCCL=1), off by default.

    Measured on one rank (round 4, ``bench.py`` ``variants.sharded_normaliser_probe``): the sharded step costs 1.3 us more than
    the unsharded one with this call and 2.1 us more through ``torch.distributed.all_gather_into_tensor`` -- the collective is
    not what a scan waits for either way.  (An earlier measurement of +100 us "through c10d" turned out to be the placement of
    freshly created streams on the hardware queues, see ``engine.group_streams``.)  The direct call saves the hand-over to and
    from the process group's private stream and ~30 us of host time per scan, which matters only to a host-bound loop; it has
    never run with two ranks (RCCL refuses two ranks on one GPU), so ``torch.distributed`` stays the default.  When enabled:
    ``torch.distributed`` (backend nccl) still forms the job and carries the unique id to the ranks once, and ``create`` returns
    None -- c10d stays in charge -- whenever anything here fails, does not finish within its time limit or fails its self-check
    (every rank contributes f(rank); every rank must see all of them)."""

    _DOUBLE = 8                                            # ncclFloat64 (rccl.h)

    @classmethod
    def create(cls, device, group=None, timeout=60.0):
        import os
        import threading
        if os.environ.get("SLAM2D_DIRECT_RCCL", "0") != "1" or not dist.is_initialized() or dist.get_backend(group) != "nccl":
            return None
        box = {}

        def work():
            try:
                box["obj"] = cls(device, group)
            except Exception as exc:                       # any failure: the c10d path stands
                box["err"] = repr(exc)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            # a broadcast or ncclCommInitRank of the setup is still in flight on the helper thread: falling back to c10d now would
            # let it pair with the job's next collective.  Fail hard; the option is opt-in.
            raise RuntimeError(f"DirectRccl: the communicator did not form within {timeout:.0f} s (unset SLAM2D_DIRECT_RCCL)")
        # every rank must take the same all-gather path: agree on the outcome over c10d, use the direct call only if ALL succeeded
        ok = torch.tensor([1 if "obj" in box else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            cls.last_error = box.get("err") or "another rank could not form the direct communicator"
            return None
        return box["obj"]

    last_error = None

    def __init__(self, device, group=None):
        import ctypes as C
        import os
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        torch.cuda.set_device(device)                      # (this may be a helper thread: the device is per thread)
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")      # the RCCL torch itself runs on
        self.L = L = C.CDLL(path if os.path.exists(path) else "librccl.so")

        class Uid(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        L.ncclGetUniqueId.argtypes, L.ncclGetUniqueId.restype = [C.POINTER(Uid)], C.c_int
        L.ncclCommInitRank.argtypes, L.ncclCommInitRank.restype = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int], C.c_int
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllGather.restype = C.c_int
        uid = Uid()
        if self.rank == 0 and L.ncclGetUniqueId(C.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        self.comm = C.c_void_p()
        rc = L.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank: error {rc}")
        self._as_ptr = C.c_void_p
        # self-check on the caller's device: every rank contributes (rank + 1) * [1.5, 2.5, 3.5]
        mine = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64, device=device) * (self.rank + 1)
        got = torch.zeros(3 * self.world, dtype=torch.float64, device=device)
        st = torch.cuda.current_stream(device)
        st.synchronize()
        self.all_gather(mine.data_ptr(), got.data_ptr(), 3, st.cuda_stream)
        st.synchronize()
        want = torch.cat([torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64) * (r + 1) for r in range(self.world)])
        if not torch.equal(got.cpu(), want):
            raise RuntimeError("direct RCCL all-gather failed its self-check")

    def all_gather(self, send_ptr, recv_ptr, count, stream_handle):
        """recv[rank * count ...] = send[0 .. count) of every rank, float64, enqueued on ``stream_handle`` (a hipStream_t)."""
        rc = self.L.ncclAllGather(self._as_ptr(send_ptr), self._as_ptr(recv_ptr), count, self._DOUBLE, self.comm,
                                  stream_handle if isinstance(stream_handle, self._as_ptr) else self._as_ptr(stream_handle))
        if rc != 0:
            raise RuntimeError(f"ncclAllGather: error {rc}")


class ShardedNormalizer:
    """``normalize_sharded`` for device-resident particles as two HIP launches around the one
    collective: ``slam2d_weights_local`` (log-weights += log-confidence, this rank's three partials),
    an all-gather of the 24 bytes of every rank, ``slam2d_weights_merge`` (fold in rank order,
    normalise this rank's particles, variance over all N).  The torch-op version above costs a dozen
    small launches per scan (measured 77 us on one MI355X -- a quarter of a config-2 step)."""

    def __init__(self, lib, check, device, total, group=None, overlap=False, slots=1):
        """slots: partials per rank in the all-gather -- a filter whose run() steps its particles in G groups gathers G partials
        per rank and scan (filter._sharded_gather_merge); its step-by-step scans must take part in collectives of the same shape:
        the rank's one partial goes into slot 0, the other slots hold the empty partial (max -inf, sums 0: it adds nothing)."""
        self.lib, self.check, self.total, self.group = lib, check, int(total), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.slots = int(slots)
        self._part_all = torch.zeros(3 * self.slots, dtype=torch.float64, device=device)
        for k in range(1, self.slots):
            self._part_all[3 * k] = float("-inf")
        self.part = self._part_all[:3]
        self.parts = torch.zeros(3 * self.slots * self.world, dtype=torch.float64, device=device)
        self.via_host = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.rccl = DirectRccl.create(device, group) if dist.is_initialized() and not self.via_host else None
        # overlap: the collective and the merge run on a side stream, so the launch stream goes straight on to the next
        # scan's match (which needs no weight); the results are ordered by events -- the next call waits for this merge
        # before it touches logw, readers of w / stats call wait() first
        self.overlap = bool(overlap) and dist.is_initialized() and not self.via_host
        if self.overlap:
            self.side = torch.cuda.Stream(device)
            self.ev_local, self.ev_merged = torch.cuda.Event(), torch.cuda.Event()
            self.pending = False

    def wait(self):
        """Order the current stream behind the last overlapped merge (no-op otherwise)."""
        if self.overlap and self.pending:
            torch.cuda.current_stream(self.part.device).wait_event(self.ev_merged)
            self.pending = False

    def pre_local(self):
        """To be called BEFORE anything on the launch stream writes ``logw`` or ``self.part`` for the next scan -- i.e. before
        the fused slam2d_grid_update_weights_local launch (``ParticleEngine.grid_update_weights_local`` does it when it is
        handed the normaliser): with ``overlap`` the previous scan's all-gather (reads ``part``) and merge (reads and writes
        ``logw``) may still be running on the side stream."""
        self.wait()

    def __call__(self, logw, logconf_ptr, logconf_stride, w, stats, local_done=False):
        """In place on ``logw`` (this rank's log-weights); writes ``w`` and ``stats`` =
        [sum over all particles of (w - 1/N)^2, log of the pre-normalisation sum].  ``local_done``: the rank-local half
        already ran and filled ``self.part`` (slam2d_grid_update_weights_local: it rides in the map update's launch)."""
        main = torch.cuda.current_stream(logw.device)
        stream = main.cuda_stream
        n = logw.numel()
        if self.overlap:
            if local_done and self.pending:
                # the rank-local half already ran on the launch stream: it must have been ordered behind the previous merge
                # (pre_local) BEFORE it wrote logw / part -- too late to wait now
                raise RuntimeError("ShardedNormalizer(overlap=True): call pre_local() (or pass the normaliser to "
                                   "ParticleEngine.grid_update_weights_local) before the fused local launch")
            self.wait()                                       # logw, part: the previous scan's merge is done with them
        if not local_done:
            self.check(self.lib.slam2d_weights_local(logw.data_ptr(), logconf_ptr, logconf_stride, n,
                                                     self.part.data_ptr(), stream), "slam2d_weights_local")
        if self.overlap:
            self.ev_local.record(main)
            for t in (logw, w, stats):                        # used on the side stream: keep the allocator from recycling them early
                t.record_stream(self.side)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_local)
                dist.all_gather_into_tensor(self.parts, self._part_all, group=self.group)
                self.check(self.lib.slam2d_weights_merge(logw.data_ptr(), n, self.parts.data_ptr(), self.world * self.slots, self.total,
                                                         w.data_ptr(), stats.data_ptr(), self.side.cuda_stream),
                           "slam2d_weights_merge")
                self.ev_merged.record(self.side)
            self.pending = True
            return
        if not dist.is_initialized():
            self.parts.copy_(self._part_all)
        elif self.via_host:                                   # gloo: the 24 bytes hop through host memory
            mine = self._part_all.cpu()
            got = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(got, mine, group=self.group)
            self.parts.copy_(torch.cat(got))
        elif self.rccl is not None:                           # one RCCL call on the launch stream itself (no c10d, no stream hop)
            self.rccl.all_gather(self._part_all.data_ptr(), self.parts.data_ptr(), 3 * self.slots, stream)
        else:
            dist.all_gather_into_tensor(self.parts, self._part_all, group=self.group)
        self.check(self.lib.slam2d_weights_merge(logw.data_ptr(), n, self.parts.data_ptr(), self.world * self.slots, self.total,
                                                 w.data_ptr(), stats.data_ptr(), stream), "slam2d_weights_merge")


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


def _gather_sizes(sizes, total, world, rank, group=None):
    """[total, k] int64 table of every particle's size record, from this rank's [count, k] rows
    (one small all-gather, padded to the largest shard)."""
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    cap, k = max(counts), sizes.shape[1]
    mine = torch.zeros((cap, k), dtype=torch.int64)
    mine[:sizes.shape[0]] = sizes
    if not dist.is_initialized() or world == 1:
        return mine[:counts[0]]
    dev = None
    if dist.get_backend(group) != "gloo":                   # nccl moves device tensors only
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = mine.to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)]).cpu()


def migrate_ragged(cells, aux, indices, total, world, rank, group=None):
    """``migrate`` for particles whose state differs in size: the reference's resample deep-copies whatever
    shape a particle's map has grown to (Algorithm/FastSlam.py:50-62, Utils/OccupancyGrid.py:59-100).

    cells[i]: 2-D tensor (this rank's particle i's map, any [rows, pitch]); aux[i]: 1-D float64 tensor of any
    length (coordinate vectors, growth log, pose, trajectory -- whatever else has to travel with the map).
    Every rank first learns every particle's (rows, pitch, len(aux)) through one small all-gather, so a
    receiver allocates its landing buffers from the SENDER's shapes; then two point-to-point messages per
    moved particle (map, aux), issued in ascending destination order on both sides (NCCL matches by order, not
    by tag).  Returns (new_cells, new_aux) for this rank's slots."""
    first, count = shard_range(total, world, rank)
    assert len(cells) == count and len(aux) == count
    # (the element width travels too: a map promoted to 64-bit cells, MapState.promote, keeps them on its new rank)
    sizes = torch.tensor([[c.shape[0], c.shape[1], a.numel(), c.element_size()] for c, a in zip(cells, aux)], dtype=torch.int64).reshape(count, 4)
    table = _gather_sizes(sizes, total, world, rank, group)
    local, sends, recvs = resample_plan(indices, total, world, rank)
    new_cells, new_aux = [None] * count, [None] * count
    for d, s in local:
        new_cells[d], new_aux[d] = cells[s].clone(), aux[s].clone()
    ops, landing = [], []
    dev = cells[0].device if count else None
    for dst_rank, s, tag in sends:
        ops.append(dist.P2POp(dist.isend, _stage(cells[s].contiguous(), group), dst_rank, group=group, tag=2 * tag))
        ops.append(dist.P2POp(dist.isend, _stage(aux[s].contiguous(), group), dst_rank, group=group, tag=2 * tag + 1))
    for src_rank, d, tag in recvs:
        rows, pitch, naux, esize = (int(v) for v in table[int(indices[tag])])
        like = cells[0] if count else None
        cbuf = _stage(torch.empty((rows, pitch), dtype=torch.int64 if esize == 8 else like.dtype if like.element_size() == esize else torch.int32,
                                  device=like.device), group)
        abuf = _stage(torch.empty(naux, dtype=torch.float64, device=like.device), group)
        landing.append((d, cbuf, abuf))
        ops.append(dist.P2POp(dist.irecv, cbuf, src_rank, group=group, tag=2 * tag))
        ops.append(dist.P2POp(dist.irecv, abuf, src_rank, group=group, tag=2 * tag + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for d, cbuf, abuf in landing:
        new_cells[d], new_aux[d] = cbuf.to(dev), abuf.to(dev)
    assert all(t is not None for t in new_cells)
    return new_cells, new_aux


def pack_particle(X, Y, growth_log, pose, heading, trajectory, count_bound=2):
    """Everything of one particle except its count map, as one float64 vector (the ``aux`` of
    ``migrate_ragged``): [cols, rows, len(growth_log), T, count bound, pose(3), heading, trajectory (T x 2), X (cols),
    Y (rows), growth log (n x 2)].  Small integers are exact in float64."""
    import numpy as np
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    log = np.asarray(growth_log, dtype=np.float64).reshape(-1, 2)
    traj = np.asarray(trajectory, dtype=np.float64).reshape(-1, 2)
    head = np.array([len(X), len(Y), len(log), len(traj), float(count_bound)], dtype=np.float64)
    return torch.from_numpy(np.concatenate((head, np.asarray(pose, dtype=np.float64).reshape(3),
                                            [float(heading)], traj.ravel(), X, Y, log.ravel())))


def unpack_particle(aux):
    """Inverse of ``pack_particle``: dict(X, Y, growth_log, pose, heading, trajectory) of NumPy values."""
    a = aux.detach().cpu().numpy()
    cols, rows, nlog, T, bound = (int(v) for v in a[:5])
    o = 5
    pose = a[o:o + 3].copy(); o += 3
    heading = float(a[o]); o += 1
    traj = a[o:o + 2 * T].reshape(T, 2).copy(); o += 2 * T
    X = a[o:o + cols].copy(); o += cols
    Y = a[o:o + rows].copy(); o += rows
    log = [(int(s), int(n)) for s, n in a[o:o + 2 * nlog].reshape(nlog, 2)]
    assert o + 2 * nlog == a.size
    return dict(X=X, Y=Y, growth_log=log, pose=pose, heading=heading, trajectory=traj, count_bound=bound)
