"""The N > 1 path on CPU: world_size-2 (and 3, ragged) gloo groups exercising the weight
normaliser all-reduce, the weight gather and the resample migration plan/transport."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

par = importlib.import_module("slam-2d-lidar-scan_amd.parallel")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, seed, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(seed)
        logw_all = torch.from_numpy(rs.uniform(-400, -20, total))
        maps_all = [torch.from_numpy(rs.randint(0, 1 << 30, (5, 7)).astype(np.int32)) for _ in range(total)]
        idx = rs.choice(total, total)
        first, count = par.shard_range(total, world, rank)
        w, logw, var = par.normalize_sharded(logw_all[first:first + count].clone(), total)
        w_all = par.gather_weights(w, total, world)
        new = par.migrate(maps_all[first:first + count], idx, total, world, rank)
        torch.save(dict(w=w, logw=logw, var=var, w_all=w_all, new=new, first=first, count=count),
                   os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (3, 10)])
def test_sharded_normaliser_and_migration(tmp_path, world, total):
    seed = 11
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, seed, str(tmp_path)), nprocs=world, join=True)
    rs = np.random.RandomState(seed)
    logw_all = rs.uniform(-400, -20, total)
    maps_all = [rs.randint(0, 1 << 30, (5, 7)).astype(np.int32) for _ in range(total)]
    idx = rs.choice(total, total)
    w_ref = np.exp(logw_all - logw_all.max()); w_ref /= w_ref.sum()
    var_ref = ((w_ref - 1 / total) ** 2).sum()
    seen = 0
    for r in range(world):
        o = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        first, count = o["first"], o["count"]
        assert (first, count) == par.shard_range(total, world, r)
        np.testing.assert_allclose(o["w"].numpy(), w_ref[first:first + count], rtol=1e-12)
        np.testing.assert_allclose(np.exp(o["logw"].numpy()), w_ref[first:first + count], rtol=1e-10)
        np.testing.assert_allclose(float(o["var"]), var_ref, rtol=1e-9)
        np.testing.assert_allclose(o["w_all"].numpy(), w_ref, rtol=1e-12)
        for k, t in enumerate(o["new"]):
            assert np.array_equal(t.numpy(), maps_all[idx[first + k]])
        seen += count
    assert seen == total


def _uneven_particle(i, seed):
    """Particle i of a seeded population whose maps have grown unevenly: its own extent, coordinate vectors
    (high-side growth compresses the spacing, Utils/OccupancyGrid.py:79-80), growth log, pose, heading, trajectory."""
    rs = np.random.RandomState(seed * 100003 + i)
    rows, cols = 6 + int(rs.randint(0, 9)), 5 + int(rs.randint(0, 11))
    pitch = -(-cols // 16) * 16
    cells = rs.randint(0, 1 << 30, (rows, pitch)).astype(np.int32)
    X = np.sort(rs.uniform(-30, 30, cols)); Y = np.sort(rs.uniform(-30, 30, rows))
    log = [(int(rs.randint(1, 5)), int(rs.randint(1, 600))) for _ in range(int(rs.randint(0, 7)))]
    pose = rs.uniform(-20, 20, 3)
    heading = float("nan") if i % 7 == 0 else float(rs.uniform(-3, 3))
    traj = rs.uniform(-20, 20, (4, 2))
    return dict(cells=cells, X=X, Y=Y, log=log, pose=pose, heading=heading, traj=traj)


def _ragged_worker(rank, world, port, total, seed, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(seed)
        # a near-degenerate weight vector, like the one that triggers the reference's resample: a handful of
        # survivors are copied all over the node, so most copies cross ranks
        w = rs.uniform(0, 1, total) ** 40
        w /= w.sum()
        idx = rs.choice(total, total, p=w)
        first, count = par.shard_range(total, world, rank)
        mine = [_uneven_particle(first + k, seed) for k in range(count)]
        cells = [torch.from_numpy(m["cells"]) for m in mine]
        aux = [par.pack_particle(m["X"], m["Y"], m["log"], m["pose"], m["heading"], m["traj"]) for m in mine]
        new_cells, new_aux = par.migrate_ragged(cells, aux, idx, total, world, rank)
        ok = True
        for k, (c, a) in enumerate(zip(new_cells, new_aux)):
            src = _uneven_particle(int(idx[first + k]), seed)
            o = par.unpack_particle(a)
            ok &= np.array_equal(c.numpy(), src["cells"]) and np.array_equal(o["X"], src["X"]) and np.array_equal(o["Y"], src["Y"])
            ok &= o["growth_log"] == src["log"] and np.array_equal(o["pose"], src["pose"])
            ok &= np.array_equal(o["trajectory"], src["traj"])
            ok &= (np.isnan(o["heading"]) and np.isnan(src["heading"])) or o["heading"] == src["heading"]
        moved = sum(par.owner_of(int(idx[first + k]), total, world) != rank for k in range(count))
        open(os.path.join(out_dir, f"r{rank}"), "w").write(f"{bool(ok)} {moved}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(8, 512), (3, 10)])
def test_ragged_resample_migration(tmp_path, world, total):
    """BASELINE config 4's resample on CPU: 512 particles on 8 ranks (64 per rank) whose maps have grown
    unevenly; every destination slot must receive its source particle whole -- map of the SENDER's shape,
    coordinate vectors, growth log, pose, heading, trajectory (Algorithm/FastSlam.py:50-62 deep-copies
    whatever a particle holds)."""
    mp.spawn(_ragged_worker, args=(world, _free_port(), total, 5, str(tmp_path)), nprocs=world, join=True)
    res = [open(os.path.join(str(tmp_path), f"r{r}")).read().split() for r in range(world)]
    assert all(r[0] == "True" for r in res), res
    assert sum(int(r[1]) for r in res) > total // 2          # the plan really moved particles across ranks


def test_shard_bookkeeping():
    for total in (1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [par.shard_range(total, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == total
            assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            for i in range(total):
                r = par.owner_of(i, total, world) if total >= world else None
                if r is not None:
                    assert spans[r][0] <= i < spans[r][0] + spans[r][1]


def test_single_process_normaliser_matches_numpy():
    rs = np.random.RandomState(2)
    lw = rs.uniform(-50, 0, 33)
    w, logw, var = par.normalize_sharded(torch.from_numpy(lw), 33)
    ref = np.exp(lw - lw.max()); ref /= ref.sum()
    np.testing.assert_allclose(w.numpy(), ref, rtol=1e-13)
    np.testing.assert_allclose(float(var), ((ref - 1 / 33) ** 2).sum(), rtol=1e-10)


@pytest.mark.parametrize("world,total", [(8, 512), (8, 1024), (8, 509), (4, 64), (3, 10)])
def test_point_to_point_order_is_identical_on_both_sides(world, total):
    """NCCL / RCCL matches the point-to-point operations of a group call BY ORDER per (sender, receiver) pair, not by tag (gloo,
    which the multi-process tests here run on, matches by tag and would hide a mismatch).  For random resample draws at the
    8-rank sizes of BASELINE configs 4 and 5: the sequence of particles rank a sends to rank b is exactly the sequence rank b
    receives from rank a -- and the receiver sizes every landing buffer from the SENDER's particle (migrate_ragged's size
    table), so ragged maps fit.  Pure plan arithmetic: no process group needed.  (Algorithm/FastSlam.py:50-62.)"""
    rs = np.random.RandomState(world * 1000 + total)
    for trial in range(6):
        w = rs.dirichlet(np.full(total, 0.05 if trial % 2 else 1.0))           # degenerate and flat weight vectors
        idx = rs.choice(total, total, p=w)
        rows = rs.randint(100, 200, total)                                      # every particle's map has its own extent
        plans = [par.resample_plan(idx, total, world, r) for r in range(world)]
        moved = 0
        for a in range(world):
            fa, _ = par.shard_range(total, world, a)
            for b in range(world):
                if a == b:
                    continue
                fb, _ = par.shard_range(total, world, b)
                sent = [(tag, rows[fa + src]) for dst_rank, src, tag in plans[a][1] if dst_rank == b]
                # the receiver looks the size up under the SOURCE particle's global index, indices[tag] (migrate_ragged)
                recv = [(tag, rows[int(idx[tag])]) for src_rank, dst, tag in plans[b][2] if src_rank == a]
                assert sent == recv, f"pair {a}->{b}: send and receive sequences differ"
                assert [t for t, _ in sent] == sorted(t for t, _ in sent)
                moved += len(sent)
        # every destination slot is filled exactly once: locally or by one receive
        for r in range(world):
            first, count = par.shard_range(total, world, r)
            filled = sorted([d for d, _ in plans[r][0]] + [d for _, d, _ in plans[r][2]])
            assert filled == list(range(count))
        assert moved == sum(par.owner_of(int(idx[d]), total, world) != par.owner_of(d, total, world) for d in range(total))


def test_direct_rccl_is_only_for_nccl_groups():
    """parallel.DirectRccl (the normaliser's all-gather straight from librccl) must decline -- and leave torch.distributed in
    charge -- without a process group, and it can be switched off."""
    assert par.DirectRccl.create(torch.device("cpu")) is None              # off by default
    os.environ["SLAM2D_DIRECT_RCCL"] = "1"
    try:
        assert par.DirectRccl.create(torch.device("cpu")) is None          # asked for, but there is no nccl group to ride on
    finally:
        os.environ.pop("SLAM2D_DIRECT_RCCL")
