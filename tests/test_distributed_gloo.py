"""N > 1 path on CPU: two gloo ranks shard the sample-index stream, label their shards (CPU oracle as the
stand-in for the per-GPU kernels), compact, all-gather fixed-capacity blocks -- and every rank ends up
with exactly the valid set a single process computes for the union range."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import common
import oracle_py as O
from art_planner_amd.distributed import (EdgeResultGatherer, ValidBitmapGatherer, ValidIndexGatherer,
                                         ValidStateGatherer, agree_capacity, shard_first_index)
from synthetic import make_map

BATCH, STEPS, WORLD = 512, 3, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_valid(gm, rob, step, rank, world):
    smp = O.OracleSampler(gm)
    se3, _ = smp.sample(rob, 42, shard_first_index(step, rank, world, BATCH), BATCH)
    valid = O.OracleMap(gm).states_valid(rob, se3)
    return se3, valid


def _shard_edges(gm, rob, se3, valid):
    """Edges of a shard as SURVEY.md 8d pairs them: accepted state i with accepted states i+1 .. i+3 when their
    lateral distance is below 2 m; verdict = the 0.5 m interpolation rule (oracle), cost = a deterministic
    stand-in triple (length, |dz|, |dyaw| proxy) -- the exchange does not care what the three floats are."""
    pos = np.flatnonzero(valid)
    acc = se3[pos]
    ei, ej = [], []
    for d in (1, 2, 3):
        a, b = acc[:-d], acc[d:]
        near = np.hypot(a[:, 0] - b[:, 0], a[:, 1] - b[:, 1]) < 2.0
        ei.append(pos[:-d][near])
        ej.append(pos[d:][near])
    ei, ej = np.concatenate(ei), np.concatenate(ej)
    ok, _ = O.OracleMap(gm).edges_interp_valid(rob, se3[ei], se3[ej])
    d3 = se3[ej, :3] - se3[ei, :3]
    cost = np.stack([np.linalg.norm(d3, axis=1), np.abs(d3[:, 2]), np.abs(se3[ej, 6] - se3[ei, 6])], 1).astype(np.float32)
    return ei.astype(np.uint32), ej.astype(np.uint32), ok, cost


def _worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    gm = make_map(96, 0.04, seed=3)
    rob = O.robot("yaml")
    dev = torch.device("cpu")
    # warm-up step sizes the blocks
    _, v0 = _shard_valid(gm, rob, 1000, rank, WORLD)
    cap = agree_capacity(int(v0.sum()), BATCH, dev)
    g = ValidStateGatherer(WORLD, cap, dev)
    gi = ValidIndexGatherer(WORLD, cap, dev)
    smp = O.OracleSampler(gm)
    merged = []
    gb = ValidBitmapGatherer(WORLD, BATCH, dev)
    ge = EdgeResultGatherer(WORLD, 3 * BATCH, dev)
    all_ij, all_cost = [], []
    for step in range(STEPS):
        se3, valid = _shard_valid(gm, rob, step, rank, WORLD)
        # the second exchange: edge results {u32 i, u32 j, f32 cost[3]} of the valid edges, fixed-capacity blocks
        ei, ej, ok_e, cost = _shard_edges(gm, rob, se3, valid)
        keep = ok_e != 0
        rec = np.zeros((3 * BATCH, 5), np.int32)
        n_e = int(keep.sum())
        rec[:n_e, 0] = ei[keep].view(np.int32)
        rec[:n_e, 1] = ej[keep].view(np.int32)
        rec[:n_e, 2:] = cost[keep].view(np.int32)
        ge.gather(torch.from_numpy(rec), torch.tensor([n_e], dtype=torch.int64))
        ij, c, ok3 = ge.global_records(step, BATCH)
        assert ok3
        all_ij.append(ij.numpy().copy())
        all_cost.append(c.numpy().copy())
        comp = torch.zeros((BATCH, 7), dtype=torch.float64)
        sel = torch.from_numpy(se3[valid != 0])
        comp[:len(sel)] = sel
        g.gather(comp, torch.tensor([len(sel)], dtype=torch.int64))
        m, ok = g.merged()
        assert ok
        # the index exchange (4 B per accepted state) + local re-materialisation gives the same states
        idx = torch.zeros(BATCH, dtype=torch.int32)
        pos = np.flatnonzero(valid)
        idx[:len(pos)] = torch.from_numpy(pos.astype(np.int32))
        gi.gather(idx, torch.tensor([len(pos)], dtype=torch.int64))
        gidx, ok2 = gi.global_indices(step, BATCH)
        assert ok2
        regen = np.stack([smp.sample(rob, 42, int(k), 1)[0][0] for k in gidx.tolist()]) if len(gidx) else np.zeros((0, 7))
        assert np.array_equal(regen, m.numpy())
        # the bitmap exchange (1 bit per candidate: what bench.py gathers) names the same global indices
        bits = np.packbits(valid.astype(np.uint8), bitorder="little")
        bits = np.concatenate([bits, np.zeros((-len(bits)) % 8, np.uint8)]).view(np.int64)
        gb.gather(torch.from_numpy(bits.copy()))
        assert torch.equal(gb.global_indices(step), gidx)
        merged.append(m.numpy().copy())
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate(merged, 0))
    np.save(os.path.join(out_dir, f"edges_ij_rank{rank}.npy"), np.concatenate(all_ij, 0))
    np.save(os.path.join(out_dir, f"edges_cost_rank{rank}.npy"), np.concatenate(all_cost, 0))
    dist.barrier()
    dist.destroy_process_group()


def test_shards_tile_the_stream():
    seen = []
    for world in (1, 2, 4, 8):
        idx = sorted(shard_first_index(s, r, world, BATCH) for s in range(4) for r in range(world))
        assert idx == [k * BATCH for k in range(4 * world)]
        seen.append(idx)


def test_two_rank_gather_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)          # every rank holds the same gathered set
    # single process over the union range, in (step, rank) order
    gm = make_map(96, 0.04, seed=3)
    rob = O.robot("yaml")
    ref = []
    for step in range(STEPS):
        for rank in range(WORLD):
            se3, valid = _shard_valid(gm, rob, step, rank, WORLD)
            ref.append(se3[valid != 0])
    ref = np.concatenate(ref, 0)
    assert np.array_equal(r0, ref)
    assert len(ref) > 0
    # edge results: every rank holds the same records, equal to a single process's over the union range, with
    # GLOBAL sample indices (the endpoints can be re-materialised anywhere from (seed, index))
    ij0, ij1 = np.load(tmp_path / "edges_ij_rank0.npy"), np.load(tmp_path / "edges_ij_rank1.npy")
    c0, c1 = np.load(tmp_path / "edges_cost_rank0.npy"), np.load(tmp_path / "edges_cost_rank1.npy")
    assert np.array_equal(ij0, ij1) and np.array_equal(c0, c1)
    ref_ij, ref_c = [], []
    for step in range(STEPS):
        for rank in range(WORLD):
            se3, valid = _shard_valid(gm, rob, step, rank, WORLD)
            ei, ej, ok_e, cost = _shard_edges(gm, rob, se3, valid)
            base = shard_first_index(step, rank, WORLD, BATCH)
            ref_ij.append(np.stack([ei[ok_e != 0].astype(np.int64) + base, ej[ok_e != 0].astype(np.int64) + base], 1))
            ref_c.append(cost[ok_e != 0])
    ref_ij, ref_c = np.concatenate(ref_ij), np.concatenate(ref_c)
    assert len(ref_ij) > 8
    assert np.array_equal(ij0, ref_ij) and np.array_equal(c0, ref_c)
    smp = O.OracleSampler(gm)
    k = int(ref_ij[7, 1])
    st = smp.sample(rob, 42, k, 1)[0][0]
    assert O.OracleMap(gm).states_valid(rob, st[None])[0] == 1     # an endpoint id is a valid state of the stream


def _synthetic_worker(rank, port, world, out_dir):
    """Capacity agreement, rank offsets and edge-record packing for any world size, on synthetic labels (rank r accepts
    every (r + 2)-th candidate; its edges connect consecutive accepted candidates, ids above 2^31 included)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    batch = 4096
    pos = np.arange(rank, batch, rank + 2)                   # accepted in-batch indices of this rank
    cap = agree_capacity(len(pos), batch, dev, slack=1.0)
    caps = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(caps, torch.tensor([cap], dtype=torch.int64))
    assert len({int(c.item()) for c in caps}) == 1            # the SAME capacity on every rank ...
    assert cap == min(batch, max(len(np.arange(r, batch, r + 2)) for r in range(world)) + 1024)   # ... = the largest count + slack
    gi = ValidIndexGatherer(world, cap, dev)
    ge = EdgeResultGatherer(world, cap, dev)
    gb = ValidBitmapGatherer(world, batch, dev)
    for step in (0, 5):
        idx = torch.zeros(cap, dtype=torch.int32)
        idx[:len(pos)] = torch.from_numpy(pos.astype(np.int32))
        gi.gather(idx, torch.tensor([len(pos)], dtype=torch.int64))
        g, ok = gi.global_indices(step, batch)
        want = np.concatenate([np.arange(r, batch, r + 2) + shard_first_index(step, r, world, batch) for r in range(world)])
        assert ok and np.array_equal(g.numpy(), want)
        valid = np.zeros(batch, np.uint8)
        valid[pos] = 1
        gb.gather(torch.from_numpy(np.packbits(valid, bitorder="little").view(np.int64).copy()))
        assert torch.equal(gb.global_indices(step), g)
        # 20-byte records {u32 i, u32 j, f32 cost[3]} in int32 storage
        n_e = len(pos) - 1
        rec = np.zeros((cap, 5), np.int32)
        rec[:n_e, 0] = pos[:-1].astype(np.uint32).view(np.int32)
        rec[:n_e, 1] = (pos[1:].astype(np.uint32) | (np.uint32(0x80000000) if rank % 2 else np.uint32(0))).view(np.int32)
        rec[:n_e, 2:] = (np.arange(3 * n_e, dtype=np.float32).reshape(n_e, 3) + 1000.0 * rank).view(np.int32)
        ge.gather(torch.from_numpy(rec), torch.tensor([n_e], dtype=torch.int64))
        ij, cost, ok3 = ge.global_records(step, batch)
        assert ok3
        at = 0
        for r in range(world):
            p_r = np.arange(r, batch, r + 2)
            base = shard_first_index(step, r, world, batch)
            hi = 0x80000000 if r % 2 else 0
            assert np.array_equal(ij[at:at + len(p_r) - 1, 0].numpy(), p_r[:-1] + base)
            assert np.array_equal(ij[at:at + len(p_r) - 1, 1].numpy(), (p_r[1:] | hi) + base)      # unsigned 32-bit ids survive
            assert np.array_equal(cost[at:at + len(p_r) - 1].numpy(),
                                  np.arange(3 * (len(p_r) - 1), dtype=np.float32).reshape(-1, 3) + 1000.0 * r)
            at += len(p_r) - 1
        assert at == len(ij)
    # a block that does not fit is reported, not truncated silently
    small = ValidIndexGatherer(world, 8, dev)
    small.gather(torch.zeros(8, dtype=torch.int32), torch.tensor([9 if rank == world - 1 else 3], dtype=torch.int64))
    assert small.global_indices(0, batch)[1] is False
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_capacity_agreement_rank_offsets_and_edge_records_for_any_world_size(tmp_path, world):
    """the exchange's host logic at W = 2, 4, 8 (gloo, CPU): one agreed block capacity, every rank's block
    at its rank offset with the shard's first index added, unsigned 32-bit ids and bit-cast float costs through the 20-byte
    records, overflow reported."""
    mp.spawn(_synthetic_worker, args=(_free_port(), world, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
