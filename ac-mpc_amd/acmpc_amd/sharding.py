"""Multi-GPU form of one solve: candidates are independent, so each rank (one process per
GPU) rolls out its own contiguous slice and the ranks meet once, in an all-reduce(MIN) over the per-problem packed
(cost, global index) keys - RCCL over xGMI when the tensors live on the GPU (`backend="nccl"` is RCCL on ROCm),
gloo on the CPU in tests.  The rank that owns the winning candidate then contributes its record (selected
controls + predicted states) to an all-reduce(SUM) in which every other rank adds zeros, so the result is exact
and every rank ends with the same plan.  Payloads are O(P) keys and O(P n) floats: latency-bound, so a single
flat collective each, no ring pipelining or bucketing.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [offset, offset + count) of `total` candidates owned by `rank`; the first
    `total % world_size` ranks hold one extra."""
    base, extra = divmod(total, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def global_select(local_keys: torch.Tensor, make_records: Callable[[torch.Tensor], torch.Tensor],
                  group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """local_keys: int64 [P], this rank's best key per problem.  Returns (global_keys [P], records [P, R]).

    `make_records(global_keys)` must return this rank's [P, R] float32 records: the winner's record where this
    rank owns the winning index, zeros elsewhere (column 2 always carries the local feasible count)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_keys, op=dist.ReduceOp.MIN, group=group)
        records = make_records(local_keys)
        dist.all_reduce(records, op=dist.ReduceOp.SUM, group=group)
    else:
        records = make_records(local_keys)
    return local_keys, records


def global_select_sampled(local_keys: torch.Tensor, regenerate_records: Callable[[torch.Tensor], torch.Tensor],
                          group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Single-collective form for counter-based candidates: all-reduce(MIN) the keys, then EVERY rank rebuilds the
    winners' records from the global indices in the keys (`regenerate_records(global_keys) -> [P, R]`)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_keys, op=dist.ReduceOp.MIN, group=group)
    return local_keys, regenerate_records(local_keys)


def softmin_payload(mean: torch.Tensor, weight_sum: torch.Tensor, count: int) -> torch.Tensor:
    """What one shard contributes to the softmin mean of all shards: [P][2n + 2] float64 = its own weighted mean, the
    sum of its weights (relative to the GLOBAL minimum cost) and its candidate count."""
    P = mean.shape[0]
    return torch.cat([mean.double().reshape(P, -1), weight_sum.double().reshape(P, 1),
                      torch.full((P, 1), float(count), dtype=torch.float64, device=mean.device)], dim=1)


def combine_softmin(payloads, n_steps: int):
    """Shard payloads in rank order -> (mean [P][n][2] float32, weight_sum [P] float64).  mean = sum_r w_r mean_r /
    sum_r w_r in float64, accumulated in the order given (deterministic); a problem whose weights are all zero (no
    finite cost on any shard) gets the plain average of the candidates, as the single-GPU kernel gives it."""
    num = torch.zeros_like(payloads[0][:, :-2])
    uni = torch.zeros_like(num)
    den = torch.zeros_like(payloads[0][:, -2:-1])
    cnt = torch.zeros_like(den)
    for t in payloads:
        m, w, c = t[:, :-2], t[:, -2:-1], t[:, -1:]
        num = num + m * w
        den = den + w
        uni = uni + m * c
        cnt = cnt + c
    usable = den > 0
    mean = torch.where(usable, num / torch.where(usable, den, torch.ones_like(den)), uni / cnt)
    return mean.float().reshape(-1, n_steps, 2), den[:, 0].clone()


class ShardedRollout:
    """Binds an `Engine` to this rank's slice of the candidates.  All tensors are torch CUDA tensors; the engine
    is handed raw pointers and the current stream, RCCL runs on the same stream through torch.distributed."""

    def __init__(self, engine, n_problems: int, n_local: int, n_steps: int, layout: int, index_offset: int,
                 device: torch.device, group: Optional[dist.ProcessGroup] = None, want_costs: bool = True,
                 host_collectives: bool = False, want_keys: bool = False):
        """`want_keys=True` makes a single-rank rollout export its reduced keys too (softmin() needs them; with several
        ranks they are exported anyway).  `host_collectives=True` stages the (tiny) collective payloads through the CPU - for process groups whose
        backend cannot reduce GPU tensors (gloo rehearsals of the multi-rank path on a single GPU)."""
        from ._capi import record_floats
        self.host_collectives = host_collectives
        self.want_keys = want_keys
        self._soft = None
        self._records_b, self._stream_count = None, 0   # step_stream(): the second record buffer, batches so far

        self.engine, self.group = engine, group
        self.P, self.N, self.n, self.layout, self.offset = n_problems, n_local, n_steps, layout, index_offset
        self.keys = torch.empty(n_problems, dtype=torch.int64, device=device)
        self.records = torch.empty(n_problems, record_floats(n_steps), dtype=torch.float32, device=device)
        self.costs = torch.empty(n_problems, n_local, dtype=torch.float32, device=device) if want_costs else None
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.sampler = None
        self.rccl_comm = 0     # use_library_collective()

    def use_library_collective(self, rccl_comm: int):
        """Carry the step's one collective - the all-reduce(MIN) of the packed keys - through the library's own entry point,
        `acmpc_reduce_across_ranks` on `rccl_comm` (an ncclComm_t as an integer: `LibraryCommunicator` below, or the
        host's own), enqueued on the step's stream, instead of torch.distributed.  The rank then takes the multi-rank path
        (rollout -> keys -> reduce -> finalize) even when it is the only one: a one-rank communicator is a valid one."""
        self.rccl_comm = int(rccl_comm)
        self.distributed = True

    def use_sampler(self, centre: torch.Tensor, u_ref: Optional[torch.Tensor], sigma, seed: int, round_: int = 0):
        """Declare that the control matrices handed to rollout()/select() were produced by `sample()` with these
        parameters.  select() then re-draws each winner from the global index in its key on EVERY rank
        (acmpc_finalize_sampled_device): ONE all-reduce(MIN) of the keys per step, no record exchange."""
        self.sampler = dict(centre=centre, u_ref=u_ref, sigma=tuple(sigma), seed=int(seed), round=int(round_))

    def sample(self, U: torch.Tensor, stream: int, seed: Optional[int] = None, round_: Optional[int] = None):
        """Fill `U` with this rank's slice of the candidates (global indices offset .. offset + N)."""
        sp = self.sampler
        self.engine.sample_device(sp["centre"].data_ptr(), sp["centre"].shape[-2] * 2 if sp["centre"].dim() == 3
                                  else sp["centre"].shape[-1], sp["u_ref"].data_ptr() if sp["u_ref"] is not None else 0,
                                  self.P, self.N, self.n, self.layout, self.offset, sp["sigma"],
                                  sp["seed"] if seed is None else seed, sp["round"] if round_ is None else round_,
                                  U.data_ptr(), stream)

    def rollout(self, x0: torch.Tensor, U: torch.Tensor, stream: int):
        """The dominant kernel alone: controls in, costs and per-workgroup partial keys out."""
        self.engine.rollout_device(x0.data_ptr(), U.data_ptr(), self.P, self.N, self.n, self.layout, self.offset,
                                   self.costs.data_ptr() if self.costs is not None else 0,
                                   self.keys.data_ptr() if (self.distributed or self.want_keys) else 0, stream)

    def select(self, x0: torch.Tensor, U: torch.Tensor, stream: int, seed: Optional[int] = None,
               round_: Optional[int] = None):
        """argmin across workgroups (and across ranks), then the winner's record."""
        if self.sampler is not None:
            sp = self.sampler
            if self.distributed:
                self._reduce_keys(stream)   # the only collective
            centre = sp["centre"]
            stride = centre.shape[-2] * 2 if centre.dim() == 3 else centre.shape[-1]
            self.engine.finalize_sampled_device(self.keys.data_ptr() if self.distributed else 0, x0.data_ptr(),
                                                centre.data_ptr(), stride,
                                                sp["u_ref"].data_ptr() if sp["u_ref"] is not None else 0, self.P, self.N,
                                                self.n, sp["sigma"], sp["seed"] if seed is None else seed,
                                                sp["round"] if round_ is None else round_, self.records.data_ptr(),
                                                stream)
            return self.records
        if self.distributed:
            self._reduce_keys(stream)
            self.engine.finalize_device(self.keys.data_ptr(), x0.data_ptr(), U.data_ptr(), self.P, self.N, self.n,
                                        self.layout, self.offset, self.records.data_ptr(), stream)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                self._all_reduce(self.records, dist.ReduceOp.SUM)
        else:
            self.engine.finalize_device(0, x0.data_ptr(), U.data_ptr(), self.P, self.N, self.n, self.layout,
                                        self.offset, self.records.data_ptr(), stream)
        return self.records

    def step(self, x0: torch.Tensor, U: torch.Tensor, stream: int, seed: Optional[int] = None,
             round_: Optional[int] = None):
        """rollout + select.  A rank that holds ALL the candidates has no collective between the two: it makes ONE call
        into the library (acmpc_solve_sampled_device / acmpc_solve_device), which is one launch where the shape allows it."""
        if self.distributed or self.offset != 0:   # (a slice with an index offset belongs to a wider selection)
            self.rollout(x0, U, stream)
            return self.select(x0, U, stream, seed=seed, round_=round_)
        costs = self.costs.data_ptr() if self.costs is not None else 0
        keys = self.keys.data_ptr() if self.want_keys else 0
        if self.sampler is not None:
            sp = self.sampler
            centre = sp["centre"]
            stride = centre.shape[-2] * 2 if centre.dim() == 3 else centre.shape[-1]
            self.engine.solve_sampled_device(x0.data_ptr(), U.data_ptr(), centre.data_ptr(), stride,
                                             sp["u_ref"].data_ptr() if sp["u_ref"] is not None else 0, self.P, self.N,
                                             self.n, self.layout, sp["sigma"], sp["seed"] if seed is None else seed,
                                             sp["round"] if round_ is None else round_, costs, keys,
                                             self.records.data_ptr(), stream)
        else:
            self.engine.solve_device(x0.data_ptr(), U.data_ptr(), self.P, self.N, self.n, self.layout, costs, keys,
                                     self.records.data_ptr(), stream)
        return self.records

    def step_stream(self, x0: torch.Tensor, U: torch.Tensor, stream: int, seed: Optional[int] = None,
                    round_: Optional[int] = None) -> torch.Tensor:
        """One batch of a STREAM of batches on a rank that holds all the candidates (acmpc_solve_stream_device): the rollout
        now, argmin and records inside the next batch's launch (or behind `flush`).  Returns the tensor that will hold this
        batch's records - one of two, alternating, so that a consumer can read batch k while batch k + 1's are written;
        `x0`, the sampler's centre / reference and - without a sampler - `U` stay as they are until then."""
        if self.distributed or self.offset != 0:
            raise RuntimeError("step_stream() is for a rank that holds all the candidates: use step()")
        if self._records_b is None:
            self._records_b = torch.empty_like(self.records)
        self._stream_count += 1
        out = self.records if self._stream_count % 2 else self._records_b
        costs = self.costs.data_ptr() if self.costs is not None else 0
        keys = self.keys.data_ptr() if self.want_keys else 0
        if self.sampler is not None:
            sp = self.sampler
            centre = sp["centre"]
            stride = centre.shape[-2] * 2 if centre.dim() == 3 else centre.shape[-1]
            centre_ptr, ref_ptr = centre.data_ptr(), sp["u_ref"].data_ptr() if sp["u_ref"] is not None else 0
            sigma, seed_, rnd = sp["sigma"], sp["seed"] if seed is None else seed, sp["round"] if round_ is None else round_
        else:
            centre_ptr, ref_ptr, stride, sigma, seed_, rnd = 0, 0, 2 * self.n, (0.0, 0.0), 0, 0
        self.engine.solve_stream_device(x0.data_ptr(), U.data_ptr(), centre_ptr, stride, ref_ptr, self.P, self.N, self.n,
                                        self.layout, sigma, seed_, rnd, costs, keys, out.data_ptr(), stream)
        return out

    def flush(self, stream: int):
        """The last batch of a stream: its argmin and records, a launch of their own."""
        self.engine.solve_stream_flush(stream)

    def softmin(self, U: torch.Tensor, stream: int):
        """Softmin-weighted mean control sequence over the candidates of ALL ranks (the weighted-reduction form of
        localiser.py:572-579 across shards; SURVEY 8e's softmin variant).  Call after select(): the weights are
        exp(-(cost - min) / lambda) with the GLOBAL minimum, which the all-reduced keys carry.  Every rank runs the
        softmin kernels on its own slice, ONE all-gather moves [P][2n + 2] float64 per rank (local mean, local weight sum,
        local count), and every rank combines them in rank order - the same bits on every rank, run after run.
        Returns (mean [P][n][2] float32, weight_sum [P] float64)."""
        if self.costs is None:
            raise RuntimeError("softmin() needs the per-candidate costs: construct with want_costs=True")
        if not (self.distributed or self.want_keys):
            raise RuntimeError("softmin() needs the reduced keys: construct with want_keys=True")
        if self._soft is None:
            dev = self.keys.device
            self._soft = (torch.empty(self.P, self.n, 2, dtype=torch.float32, device=dev),
                          torch.empty(self.P, dtype=torch.float64, device=dev))
        mean, wsum = self._soft
        self.engine.softmin_device(self.costs.data_ptr(), self.keys.data_ptr(), U.data_ptr(), self.P, self.N, self.n,
                                   self.layout, mean.data_ptr(), wsum.data_ptr(), stream)
        if not self.distributed:
            return mean, wsum
        mine = softmin_payload(mean, wsum, self.N)
        world = dist.get_world_size(self.group)
        if self.host_collectives:
            staged = mine.cpu()                # synchronises with the stream that produced it
            parts = [torch.empty_like(staged) for _ in range(world)]
            dist.all_gather(parts, staged, group=self.group)
            parts = [t.to(mine.device) for t in parts]
        else:
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=self.group)
        return combine_softmin(parts, self.n)

    def _reduce_keys(self, stream: int):
        if self.rccl_comm:
            self.engine.reduce_across_ranks(self.rccl_comm, self.keys.data_ptr(), self.P, stream)
        else:
            self._all_reduce(self.keys, dist.ReduceOp.MIN)

    def _all_reduce(self, tensor: torch.Tensor, op):
        if self.host_collectives:
            staged = tensor.cpu()              # synchronises with the stream that produced `tensor`
            dist.all_reduce(staged, op=op, group=self.group)
            tensor.copy_(staged)
        else:
            dist.all_reduce(tensor, op=op, group=self.group)


class LibraryCommunicator:
    """An RCCL communicator made by the library itself (acmpc_rccl_unique_id / acmpc_rccl_comm_create: the copy of RCCL that
    `acmpc_reduce_across_ranks` resolves), one rank per GPU.  Rank 0 draws the ncclUniqueId; with more than one rank it
    travels through the torch.distributed group that is already up (any backend - it is 128 bytes of host memory).
    `ShardedRollout.use_library_collective(communicator.handle)` then routes the step's all-reduce through the C ABI."""

    def __init__(self, device_index: int, group: Optional[dist.ProcessGroup] = None):
        from . import _capi
        self._capi = _capi
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if multi else 1
        self.rank = dist.get_rank(group) if multi else 0
        unique = [_capi.rccl_unique_id() if self.rank == 0 else None]
        if multi:
            dist.broadcast_object_list(unique, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.handle = _capi.rccl_comm_create(unique[0], self.world, self.rank, device_index)

    def close(self):
        if getattr(self, "handle", 0):
            self._capi.rccl_comm_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipelinedRollout:
    """Two-deep software pipeline over independent batches: the dominant rollout kernels run back to back on the
    caller's stream while argmin / winner record (and, with several ranks, the two small collectives) of the
    previous batch run on a side stream.  Each pipeline slot has its own `Engine`, so the per-workgroup partial
    keys of batch i are never overwritten before batch i's finalize has read them (the main stream waits for the
    slot's previous finalize before reusing it)."""

    def __init__(self, engines, n_problems: int, n_local: int, n_steps: int, layout: int, index_offset: int,
                 device: torch.device, group: Optional[dist.ProcessGroup] = None, want_costs: bool = True):
        self.slots = [ShardedRollout(e, n_problems, n_local, n_steps, layout, index_offset, device, group, want_costs)
                      for e in engines]
        self.side = torch.cuda.Stream(device=device)
        self.rolled = [torch.cuda.Event() for _ in self.slots]
        self.selected = [torch.cuda.Event() for _ in self.slots]
        self.count = 0
        self._main = None

    def bind_stream(self, stream: "torch.cuda.Stream"):
        """Fix the launch stream once instead of looking it up every step."""
        self._main = stream

    def step(self, x0: torch.Tensor, U: torch.Tensor, before_rollout=None, after_rollout=None) -> "ShardedRollout":
        """Enqueue one batch; returns the slot whose `.records` / `.costs` will hold its results once the side
        stream has drained (call `drain()` before reading them on the host or from the main stream).
        `before_rollout` / `after_rollout` are optional timing events recorded on the launch stream around the
        rollout kernel.  The host side of a step is seven driver calls, kept lean on purpose: at ~80 us of GPU
        work per step the Python launch path is otherwise the bottleneck."""
        i = self.count % len(self.slots)
        slot = self.slots[i]
        main = self._main if self._main is not None else torch.cuda.current_stream()
        if self.count >= len(self.slots):
            main.wait_event(self.selected[i])          # slot free again: its previous finalize has read the partials
        if before_rollout is not None:
            before_rollout.record(main)
        slot.rollout(x0, U, main.cuda_stream)
        done = after_rollout if after_rollout is not None else self.rolled[i]
        done.record(main)
        self.side.wait_event(done)
        if slot.distributed:
            with torch.cuda.stream(self.side):         # torch.distributed enqueues on the current stream
                slot.select(x0, U, self.side.cuda_stream)
        else:
            slot.select(x0, U, self.side.cuda_stream)
        self.selected[i].record(self.side)
        self.count += 1
        return slot

    def drain(self):
        torch.cuda.current_stream().wait_stream(self.side)


class ShardedOptimizer:
    """The closed-loop solve (`acmpc_optimize`'s rounds) with the candidates of every round spread over the ranks:
    each rank draws and rolls out its own slice of global candidate indices, ONE all-reduce(MIN) of the keys per
    round picks the global winner, every rank re-draws that winner from its index and uses it as the next round's
    centre.  All ranks end every round with identical records - except column 2 (`n_feasible`), which is each rank's
    count over its own slice - so no other exchange is needed.  `stream` must be the stream torch.distributed enqueues
    on (torch's current stream): the engine calls and the collective are ordered by stream order alone."""

    def __init__(self, engine, n_problems: int, n_local: int, n_steps: int, index_offset: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None, host_collectives: bool = False):
        from ._capi import LAYOUT_STEP_MAJOR, REC_HEADER, record_floats

        self.shard = ShardedRollout(engine, n_problems, n_local, n_steps, LAYOUT_STEP_MAJOR, index_offset, device, group,
                                    want_costs=False, host_collectives=host_collectives)
        self.shard.distributed = self.shard.distributed or host_collectives   # keys must be produced for the reduce
        self.U = torch.empty(n_problems, n_steps, 2, n_local, dtype=torch.float32, device=device)
        self._rec_header, self._rec_floats = REC_HEADER, record_floats(n_steps)

    def solve(self, x0: torch.Tensor, centre: torch.Tensor, u_ref: Optional[torch.Tensor], rounds: int, sigma,
              shrink: float = 0.5, seed: int = 0, stream: int = 0) -> torch.Tensor:
        """x0 [P,3], centre / u_ref [P,n,2] device tensors -> records [P, R] (identical on every rank)."""
        shard, n = self.shard, self.shard.n
        if stream not in (0, torch.cuda.current_stream().cuda_stream) and not shard.host_collectives:
            raise ValueError("ShardedOptimizer.solve: `stream` must be torch's current stream (the collective runs there)")
        scale = 1.0
        for r in range(rounds):
            if r == 0:
                centre_ptr, stride = centre.data_ptr(), 2 * n
            else:   # the u block of the previous round's records
                centre_ptr, stride = shard.records.data_ptr() + 4 * self._rec_header, self._rec_floats
            sig = (sigma[0] * scale, sigma[1] * scale)
            ref_ptr = u_ref.data_ptr() if u_ref is not None else 0
            shard.engine.sample_device(centre_ptr, stride, ref_ptr, shard.P, shard.N, n, shard.layout, shard.offset, sig,
                                       seed, r, self.U.data_ptr(), stream)
            shard.rollout(x0, self.U, stream)
            if shard.distributed and dist.is_initialized() and dist.get_world_size(shard.group) > 1:
                shard._all_reduce(shard.keys, dist.ReduceOp.MIN)
            shard.engine.finalize_sampled_device(shard.keys.data_ptr() if shard.distributed else 0, x0.data_ptr(),
                                                 centre_ptr, stride, ref_ptr,
                                                 shard.P, shard.N, n, sig, seed, r, shard.records.data_ptr(), stream)
            scale *= shrink
        return shard.records
