"""Candidate-sharded MPC step: one process per GPU, K/G candidates per rank, ONE
all-reduce(min) per step over xGMI (RCCL through ``torch.distributed``, backend "nccl").

Protocol: every rank maps its local result record
[J*, k*, u(3), (theta,gamma)_0..N] to order-preserving int64 keys and writes them into row
``rank`` of a [world][R] buffer whose other rows are INT64_MAX; after all-reduce(min) every
rank holds every rank's record and takes the lexicographic (cost, global index) minimum,
which is exactly ``np.argmin`` over the un-sharded candidate set.  No other collective.

``pack_record`` / ``select_record`` are the host (torch, any device) statement of the two tiny
kernels ``finalize_kernel`` / ``select_kernel``; the gloo CPU tests run them, the GPU path
runs the kernels.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

INT64_MAX = np.iinfo(np.int64).max
_LOW63 = 0x7FFFFFFFFFFFFFFF


def ordered_keys(x: torch.Tensor) -> torch.Tensor:
    """float64 -> int64 keys whose signed order equals the IEEE order of the values."""
    b = x.contiguous().view(torch.int64)
    return b ^ ((b >> 63) & _LOW63)


def ordered_values(k: torch.Tensor) -> torch.Tensor:
    b = k ^ ((k >> 63) & _LOW63)
    return b.contiguous().view(torch.float64)


def pack_record(record: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """[world][R] int64 slots: own row = keys of ``record`` (float64, R), others INT64_MAX."""
    R = record.numel()
    slots = torch.full((world, R), INT64_MAX, dtype=torch.int64, device=record.device)
    slots[rank] = ordered_keys(record.to(torch.float64))
    return slots


def select_record(slots: torch.Tensor) -> torch.Tensor:
    """Lexicographic (cost, index) minimum over the rows of the reduced slot buffer."""
    vals = ordered_values(slots)
    best = 0
    for r in range(1, vals.shape[0]):
        if vals[r, 0] < vals[best, 0] or (vals[r, 0] == vals[best, 0] and vals[r, 1] < vals[best, 1]):
            best = r
    return vals[best].clone()


def shard_bounds(K_total: int, rank: int, world: int):
    """Contiguous K/G slice of rank (the last ranks take the remainder one each)."""
    base, rem = divmod(K_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedMPC:
    """Per-rank driver.  GPU path: ``Engine`` + torch CUDA tensors + nccl.  The ``local_solver``
    hook exists so the host logic can be exercised with gloo on CPU in tests; it must return
    the local record (float64 tensor of length R with a GLOBAL index in [1])."""

    def __init__(self, engine=None, rank: Optional[int] = None, world: Optional[int] = None,
                 K_total: Optional[int] = None, local_solver: Optional[Callable] = None, group=None,
                 force_collective: bool = False, host_staged: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.engine = engine
        self.local_solver = local_solver
        self.force_collective = force_collective      # run the all-reduce even at world == 1 (rehearsal)
        self.host_staged = host_staged                # gloo rehearsal: the slot image crosses the host (blocking)
        if engine is None and local_solver is None:
            raise ValueError("ShardedMPC needs an Engine (GPU) or a local_solver")
        if engine is not None:
            K_local = engine.cfg.K
            self.K_total = K_total if K_total is not None else K_local * self.world
            self.k_offset = shard_bounds(self.K_total, self.rank, self.world)[0]
            dev = torch.device("cuda", engine.cfg.device)
            self.R = engine.result_len
            # double-buffered slots / results so step i's collective overlaps step i+1's rollout
            self.slots = [torch.empty((self.world, self.R), dtype=torch.int64, device=dev) for _ in range(2)]
            self.results = [torch.empty(self.R, dtype=torch.float64, device=dev) for _ in range(2)]
            self.comm_stream = torch.cuda.Stream(device=dev)
            self._rolled = [torch.cuda.Event() for _ in range(2)]      # rollout(i) finished writing slots[i%2]
            self._selected = [torch.cuda.Event() for _ in range(2)]    # select(i) finished reading slots[i%2]
            self._used = [False, False]
            self._flip = 0
            self._graphs = {}

    # -- GPU path ----------------------------------------------------------------------------
    def _enqueue(self, d_state: torch.Tensor, d_U: torch.Tensor, i: int, cur, comm):
        eng = self.engine
        eng.step_device_sharded(d_state.data_ptr(), d_U.data_ptr(), self.k_offset, self.rank, self.world,
                                self.slots[i].data_ptr(), cur.cuda_stream)
        if self.host_staged:
            # rehearsal of the N > 1 step without RCCL (gloo reduces host tensors): same kernels, same slot image,
            # same select -- only the transport differs, and it blocks
            cur.synchronize()
            image = self.slots[i].cpu()
            if self.world > 1 or self.force_collective:
                self.dist.all_reduce(image, op=self.dist.ReduceOp.MIN, group=self.group)
            self.slots[i].copy_(image)
            eng.select_device(self.slots[i].data_ptr(), self.world, self.results[i].data_ptr(), cur.cuda_stream)
            self._selected[i].record(cur)
            return
        self._rolled[i].record(cur)
        comm.wait_event(self._rolled[i])
        with torch.cuda.stream(comm):
            if self.world > 1 or self.force_collective:
                self.dist.all_reduce(self.slots[i], op=self.dist.ReduceOp.MIN, group=self.group)
            eng.select_device(self.slots[i].data_ptr(), self.world, self.results[i].data_ptr(), comm.cuda_stream)
        self._selected[i].record(comm)

    def step_device(self, d_state: torch.Tensor, d_U: torch.Tensor) -> torch.Tensor:
        """Enqueue one sharded step; returns the device tensor that will hold the global record
        (valid after ``synchronize()`` or on the side stream).  The rollout runs on the current
        stream, the all-reduce and the select on a side stream: step i's collective overlaps step
        i+1's rollout; only step i+2 (same slot buffer) waits for select(i)."""
        i = self._flip
        self._flip ^= 1
        cur = torch.cuda.current_stream()
        if self._used[i]:
            cur.wait_event(self._selected[i])      # slots[i] / results[i] are free again
        self._used[i] = True
        self._enqueue(d_state, d_U, i, cur, self.comm_stream)
        return self.results[i]

    def join(self):
        """The current stream waits (on the GPU, no host block) for every enqueued step's global record."""
        if not self.host_staged:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def synchronize(self):
        torch.cuda.current_stream().wait_stream(self.comm_stream)
        torch.cuda.current_stream().synchronize()

    # -- host-logic path (tests, gloo) ---------------------------------------------------------
    def step_host(self, *solver_args) -> torch.Tensor:
        record = self.local_solver(*solver_args)
        slots = pack_record(record, self.rank, self.world)
        if self.world > 1:
            self.dist.all_reduce(slots, op=self.dist.ReduceOp.MIN, group=self.group)
        return select_record(slots)


class NativeShardedMPC:
    """Candidate-sharded step with the collective issued by the library itself: the fused rollout
    kernel, ONE ``ncclAllReduce(ncclMin, ncclInt64)`` over xGMI and the select kernel are enqueued
    by a single C call (``rovmpc_step_device_allreduce``), the collective on the handle's side
    stream so it overlaps the next step's rollout.  ``torch.distributed`` is used once, to hand the
    128-byte RCCL id from rank 0 to the other ranks."""

    SLOTS = 4          # ROVMPC_COMM_SLOTS: collectives in flight = result buffers in rotation

    def __init__(self, engine, rank: Optional[int] = None, world: Optional[int] = None,
                 K_total: Optional[int] = None, group=None):
        import torch.distributed as dist
        self.engine = engine
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        K_local = engine.cfg.K
        self.K_total = K_total if K_total is not None else K_local * self.world
        self.k_offset = shard_bounds(self.K_total, self.rank, self.world)[0]
        dev = torch.device("cuda", engine.cfg.device)
        # Every rank must take the same path: a rank-0 failure is broadcast (None) instead of
        # leaving the others in the broadcast, and the outcome of ncclCommInitRank is agreed by
        # one all-reduce(MIN) before anybody uses the communicator.
        box, err = [None], None
        if self.rank == 0:
            try:
                box[0] = engine.comm_unique_id()
            except Exception as exc:                      # noqa: BLE001
                err = exc
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            raise RuntimeError(f"RCCL id not available on rank 0: {err}")
        try:
            engine.comm_init(box[0], self.rank, self.world)
        except Exception as exc:                          # noqa: BLE001
            err = exc
        if self.world > 1:
            ok = torch.tensor([0 if err else 1], dtype=torch.int32,
                              device=dev if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                if err is None:
                    engine.comm_destroy()
                raise RuntimeError(f"ncclCommInitRank failed on at least one rank (this rank: {err})")
        elif err is not None:
            raise err
        self.R = engine.result_len
        self.results = [torch.empty(self.R, dtype=torch.float64, device=dev) for _ in range(self.SLOTS)]
        self._flip = 0

    def step_device(self, d_state: torch.Tensor, d_U: torch.Tensor) -> torch.Tensor:
        i = self._flip
        self._flip = (self._flip + 1) % self.SLOTS
        self.engine.step_device_allreduce(d_state.data_ptr(), d_U.data_ptr(), self.k_offset,
                                          self.results[i].data_ptr(), torch.cuda.current_stream().cuda_stream)
        return self.results[i]

    def join(self):
        """The current stream waits (on the GPU, no host block) for every enqueued step's global record."""
        self.engine.comm_join(torch.cuda.current_stream().cuda_stream)

    def synchronize(self):
        """Blocks until every enqueued step's global record is final; raises ``RovmpcError`` when a GPU-side hand-off
        (rollout -> collective -> select) of any of them gave up -- that step's record then carries a NaN cost."""
        cur = torch.cuda.current_stream()
        self.engine.comm_sync(cur.cuda_stream)

    def abort(self):
        """From another thread: make collectives that will never complete let go (``rovmpc_comm_abort``); the stuck
        ``synchronize`` then raises.  The object is good for ``close()`` only afterwards."""
        self.engine.comm_abort()

    def close(self):
        self.engine.comm_destroy()
