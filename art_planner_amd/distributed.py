"""Multi-GPU plumbing of the sampling + validity path: one process per GPU (torch.distributed; backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards over independent sample-index ranges: rank r of W owns the states
[ (step*W + r)*S, (step*W + r + 1)*S ) of the global counter-based sample stream, so the union over ranks
and steps is a gap-free, overlap-free prefix of the stream whatever W is.  The only exchange step is the
all-gather of the ACCEPTED states (every rank's planner front end needs all of them).  A state is a pure
function of (seed, sample index), so only the 4-byte in-batch indices cross xGMI (fixed-capacity blocks
{count, idx[cap]}: static shape, runs on a side stream) and every rank re-materialises the states it needs
with the sampler (ValidIndexGatherer); ValidStateGatherer moves whole 56-byte states instead.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_first_index(step: int, rank: int, world: int, batch: int) -> int:
    """First global sample index of (step, rank) -- the same numbers as artp_shard_first_index of the C ABI."""
    return (step * world + rank) * batch


class _DeviceArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr: int, shape, typestr: str, device) -> torch.Tensor:
    """A torch tensor over device memory somebody else owns (a device group's exchange buffers): no copy.
    typestr as in numpy ("<f8", "<i8", "<u1", "<i4")."""
    return torch.as_tensor(_DeviceArray(ptr, shape, typestr), device=device)


class DeviceGroup:
    """Python view of an artp_group (include/artp_c.h "multi-GPU"): the exchange steps behind the C ABI, over RCCL
    bound directly by libartp.so -- no torch.distributed in the data path.  Two ways in, like the C entry points:
    DeviceGroup.single_process(devices) and DeviceGroup.from_rank(device, rank, world, unique_id)."""

    def __init__(self, handle, params):
        import ctypes as C
        from . import _capi
        from .context import Context
        self.L, self.h, self.params = _capi.load(), handle, params
        self.world = self.L.artp_group_world_size(self.h)
        self.n_local = self.L.artp_group_local_count(self.h)
        self.ranks = [self.L.artp_group_rank(self.h, l) for l in range(self.n_local)]
        self.contexts = []
        for l in range(self.n_local):
            ctx_h = self.L.artp_group_ctx(self.h, l)
            self.contexts.append(Context.borrowed(ctx_h, -1, params))
        self.batch = 0

    @staticmethod
    def _params(params):
        from . import _capi
        from .context import make_params
        return params if isinstance(params, _capi.Params) else make_params(params)

    @classmethod
    def single_process(cls, devices, params="yaml", transport: int = 0) -> "DeviceGroup":
        import ctypes as C
        from . import _capi
        L, p = _capi.load(), cls._params(params)
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = L.artp_group_create(arr, len(devices), C.byref(p), transport, C.byref(h))
        if rc != 0:
            raise _capi.ArtpError(f"artp_group_create failed: {L.artp_status_string(rc).decode()} ({rc})")
        g = cls(h, p)
        for l, d in enumerate(devices):
            g.contexts[l].device = d
        return g

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _capi
        L = _capi.load()
        buf = (C.c_uint8 * 128)()
        rc = L.artp_group_unique_id(buf)
        if rc != 0:
            raise _capi.ArtpError(f"artp_group_unique_id failed: {L.artp_status_string(rc).decode()} ({rc})")
        return bytes(buf)

    @classmethod
    def from_rank(cls, device: int, rank: int, world: int, unique_id: bytes, params="yaml") -> "DeviceGroup":
        import ctypes as C
        from . import _capi
        L, p = _capi.load(), cls._params(params)
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        rc = L.artp_group_create_rank(device, rank, world, buf, C.byref(p), C.byref(h))
        if rc != 0:
            raise _capi.ArtpError(f"artp_group_create_rank failed: {L.artp_status_string(rc).decode()} ({rc})")
        g = cls(h, p)
        g.contexts[0].device = device
        return g

    def _chk(self, rc, what):
        if rc != 0:
            from . import _capi
            raise _capi.ArtpError(f"{what} failed: {self.L.artp_status_string(rc).decode()} ({rc}) "
                                  f"{self.L.artp_group_last_error(self.h).decode()}")

    def ranks_seen(self) -> int:
        import ctypes as C
        n = C.c_int(0)
        self._chk(self.L.artp_group_ranks_seen(self.h, C.byref(n)), "artp_group_ranks_seen")
        return n.value

    def configure(self, seed: int, batch: int, materialise_cap: int = 0, prefix: int = 0):
        self._chk(self.L.artp_group_configure(self.h, seed, batch, materialise_cap, prefix), "artp_group_configure")
        self.batch, self.mat_cap = batch, min(materialise_cap, prefix or batch)

    def step(self, step: int):
        self._chk(self.L.artp_group_sample_and_validate_step(self.h, step), "artp_group_sample_and_validate_step")

    def synchronize(self, timeout_ms: int = -1):
        self._chk(self.L.artp_group_synchronize(self.h, timeout_ms), "artp_group_synchronize")

    def abort(self):
        self.L.artp_group_abort(self.h)

    def step_pointers(self, local: int, step: int):
        """(se3, valid, bits, states, counts) device addresses of member `local` for `step`."""
        import ctypes as C
        ptrs = [C.c_void_p() for _ in range(5)]
        self._chk(self.L.artp_group_step_buffers(self.h, local, step, *[C.byref(p) for p in ptrs]),
                  "artp_group_step_buffers")
        return tuple(p.value for p in ptrs)

    def exchange_edges(self, per_local, cap: int):
        """per_local: one (valid, edge_i, edge_j, cost, n) tuple of device addresses per local member."""
        from . import _capi
        arr = (_capi.GroupEdges * self.n_local)()
        for l, (v, i, j, c, n) in enumerate(per_local):
            arr[l].valid, arr[l].edge_i, arr[l].edge_j, arr[l].cost, arr[l].n = v, i, j, c, n
        self._chk(self.L.artp_group_exchange_edges(self.h, arr, cap), "artp_group_exchange_edges")

    def edge_pointers(self, local: int):
        import ctypes as C
        rec, cnt = C.c_void_p(), C.c_void_p()
        self._chk(self.L.artp_group_edge_buffers(self.h, local, C.byref(rec), C.byref(cnt)), "artp_group_edge_buffers")
        return rec.value, cnt.value

    def close(self):
        if getattr(self, "h", None):
            for c in self.contexts:
                c.close()
            self.L.artp_group_destroy(self.h)
            self.h = None


class ValidStateGatherer:
    """All-gather of compacted valid states with a fixed per-rank capacity."""

    def __init__(self, world: int, cap: int, device, dtype=torch.float64, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.cap, self.group = world, cap, group
        self.gathered = torch.empty((world, cap, 7), dtype=dtype, device=device)
        self.counts = torch.empty(world, dtype=torch.int64, device=device)

    def gather(self, compact: torch.Tensor, count: torch.Tensor):
        """compact: [>=cap, 7] states with the valid ones first; count: int64[1].  Asynchronous on the
        current stream for NCCL."""
        self.dist.all_gather_into_tensor(self.counts, count, group=self.group)
        self.dist.all_gather_into_tensor(self.gathered.view(-1), compact[:self.cap].reshape(-1),
                                         group=self.group)

    def merged(self) -> Tuple[torch.Tensor, bool]:
        """Valid states of all ranks in rank order (host sync).  Second value: no block overflowed."""
        counts = self.counts.tolist()
        ok = all(c <= self.cap for c in counts)
        parts = [self.gathered[r, :min(c, self.cap)] for r, c in enumerate(counts)]
        return torch.cat(parts, 0), ok


def agree_capacity(local_max_count: int, batch: int, device, slack: float = 1.1, group=None) -> int:
    """Common block capacity: max over ranks of the observed valid count, plus slack."""
    import torch.distributed as dist
    t = torch.tensor([int(local_max_count * slack) + 1024], device=device, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return min(int(t.item()), batch)


class ValidIndexGatherer:
    """All-gather of the in-batch indices (int32) of the accepted states, fixed per-rank capacity."""

    def __init__(self, world: int, cap: int, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.cap, self.group = world, cap, group
        self.gathered = torch.empty((world, cap), dtype=torch.int32, device=device)
        self.counts = torch.empty(world, dtype=torch.int64, device=device)

    def gather(self, idx: torch.Tensor, count: torch.Tensor):
        self.dist.all_gather_into_tensor(self.counts, count, group=self.group)
        self.dist.all_gather_into_tensor(self.gathered.view(-1), idx[:self.cap].reshape(-1), group=self.group)

    def global_indices(self, step: int, batch: int) -> Tuple[torch.Tensor, bool]:
        """Global sample indices of every accepted state of this step, in rank order (host sync)."""
        counts = self.counts.tolist()
        ok = all(c <= self.cap for c in counts)
        parts = [self.gathered[r, :min(c, self.cap)].to(torch.int64) + shard_first_index(step, r, self.world, batch)
                 for r, c in enumerate(counts)]
        return torch.cat(parts, 0), ok


class EdgeResultGatherer:
    """The second exchange of SURVEY.md 8e: all-gather of the edge results of every rank as fixed-capacity blocks
    of 20-byte records {u32 i, u32 j, f32 cost[3]} (artp_pack_edge_results_dev packs the valid edges of a rank;
    int32 storage, the cost floats bit-cast).  i / j are vertex ids in the owner's numbering (in-batch sample
    indices); global_records() turns them into global sample indices."""

    def __init__(self, world: int, cap: int, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.cap, self.group = world, cap, group
        self.gathered = torch.empty((world, cap, 5), dtype=torch.int32, device=device)
        self.counts = torch.empty(world, dtype=torch.int64, device=device)

    def gather(self, records: torch.Tensor, count: torch.Tensor):
        """records: int32 [>= cap, 5] with the valid records first; count: int64[1].  Asynchronous on the current
        stream for NCCL."""
        self.dist.all_gather_into_tensor(self.counts, count, group=self.group)
        self.dist.all_gather_into_tensor(self.gathered.view(-1), records[:self.cap].reshape(-1), group=self.group)

    def global_records(self, step: int, batch: int) -> Tuple[torch.Tensor, torch.Tensor, bool]:
        """(ij [E, 2] int64 global sample indices, cost [E, 3] float32) of every rank's edges in rank order (host
        sync).  Third value: no block overflowed."""
        counts = self.counts.tolist()
        ok = all(c <= self.cap for c in counts)
        ij, cost = [], []
        for r, c in enumerate(counts):
            blk = self.gathered[r, :min(c, self.cap)]
            # in-batch indices are unsigned 32 bit in int32 storage
            ij.append((blk[:, :2].to(torch.int64) & 0xffffffff) + shard_first_index(step, r, self.world, batch))
            cost.append(blk[:, 2:].contiguous().view(torch.float32))
        return torch.cat(ij, 0), torch.cat(cost, 0), ok


class ValidBitmapGatherer:
    """All-gather of the validity BITMAP of every rank's batch: one bit per candidate state (batch / 8 bytes per
    rank: 512 KiB for 2^22 candidates) -- 20x less xGMI traffic than the index lists of ValidIndexGatherer and a
    fixed size, whatever the acceptance rate.  bits: int64 [ceil(batch / 64)], bit k of word w = candidate 64 w + k
    (artp_pack_valid_bits_dev); the receiving rank expands what it needs with artp_indices_from_bits_dev."""

    def __init__(self, world: int, batch: int, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.batch, self.group = world, batch, group
        self.words = (batch + 63) // 64
        self.gathered = torch.empty((world, self.words), dtype=torch.int64, device=device)

    def gather(self, bits: torch.Tensor):
        self.dist.all_gather_into_tensor(self.gathered.view(-1), bits[:self.words].reshape(-1), group=self.group)

    def global_indices(self, step: int) -> torch.Tensor:
        """Global sample indices of every accepted state of this step, in rank order (host sync; numpy)."""
        import numpy as np
        parts = []
        for r in range(self.world):
            w = self.gathered[r].cpu().numpy().view(np.uint64)
            b = np.unpackbits(w.view(np.uint8), bitorder="little")[:self.batch]
            parts.append(torch.from_numpy(np.flatnonzero(b).astype(np.int64)) +
                         shard_first_index(step, r, self.world, self.batch))
        return torch.cat(parts, 0)
