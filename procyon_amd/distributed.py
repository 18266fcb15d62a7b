"""Data-parallel retrieval embedding (BASELINE config 3) and row-split generation / pair scoring (configs 4, 5); SURVEY.md section 8e.

Proteins are independent units, so the target set is cut into contiguous chunks of ceil(N/W) per rank (the tail
padded by wrap-around so every rank holds the same count), each rank embeds its chunk with no data-path
communication, and ONE all-gather of the [N/W, D] bf16 blocks (RCCL over xGMI when the backend is "nccl")
rebuilds the [N, D] matrix in the original order on every rank.  These are the semantics of the reference's
`SequentialDistributedSampler` (/root/reference/procyon/data/samplers.py:154-196) + gather at
/root/reference/procyon/training/trainIT.py:1594-1610 (whose receive list aliases ONE buffer W times, :1604;
the evident intent -- W distinct slots in rank order -- is what is built).

Generation and pair scoring do not shard a row: the B rows of a batch are split across the ranks the same way (contiguous chunks in
rank order, wrap-around tail), every rank runs its rows with the 8B weights replicated, no cross-GPU traffic inside the loop, and ONE
final all-gather collects the token ids / probabilities (`run_sharded_rows`; SURVEY.md section 8e "split the B rows across GPUs ...
optional final gather of token ids").
"""
from __future__ import annotations

import ctypes as C
import math

import torch


class PcyComm:
    """RCCL communicator behind the C ABI (`pcy_comm_init` / `pcy_allgather`, include/pcy.h), bootstrapped over an existing
    torch.distributed group: rank 0 draws the unique id, the group's host channel broadcasts it."""

    def __init__(self, group=None):
        import torch.distributed as td
        from . import _lib as L
        from .engine import Context
        self.ctx = Context.get()
        self.lib = self.ctx.lib
        self.rank, self.world = td.get_rank(group), td.get_world_size(group)
        idbuf = C.create_string_buffer(128)
        if self.rank == 0:
            L.check(self.lib.pcy_comm_unique_id(idbuf), "pcy_comm_unique_id")
        box = [idbuf.raw]
        td.broadcast_object_list(box, src=td.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.h = C.c_void_p()
        L.check(self.lib.pcy_comm_init(self.ctx.h, self.world, self.rank, C.create_string_buffer(box[0], 128), C.byref(self.h)), "pcy_comm_init")

    def all_gather(self, local):
        """[n, D] on this rank -> [world * n, D] in rank order (one ncclAllGather on the engine's stream)"""
        from . import _lib as L
        local = local.contiguous()
        full = torch.empty(self.world * local.shape[0], *local.shape[1:], dtype=local.dtype, device=local.device)
        L.check(self.lib.pcy_allgather(self.ctx.h, self.h, C.c_void_p(local.data_ptr()), C.c_void_p(full.data_ptr()),
                                       local.numel() * local.element_size()), "pcy_allgather")
        return full

    def close(self):
        if self.h:
            self.lib.pcy_comm_destroy(self.h)
            self.h = C.c_void_p()


_comms = {}


def _comm_for(group):
    key = id(group)
    if key not in _comms:
        _comms[key] = PcyComm(group)
    return _comms[key]


def shard_indices(n_total: int, rank: int, world: int):
    """Indices rank `rank` embeds: samplers.py:178-196 (pad by wrap-around, contiguous slice)."""
    per = int(math.ceil(n_total / world))
    idx = list(range(n_total))
    idx += idx[: per * world - n_total]
    return idx[rank * per:(rank + 1) * per]


def embed_sharded(model_or_fn, token_fn, n_total, batch_size=None, rank=None, world=None, group=None):
    """Embed proteins [0, n_total) across the process group and return the [n_total, D] matrix on every rank.

    model_or_fn: a `UnifiedProCyon` (uses forward_sequences(...)["shared"], evaluate/framework/procyon.py:318-319)
    or any callable tokens -> [b, D].  token_fn(list_of_indices) -> token matrix for those proteins.
    batch_size=None lets the engine choose (`EsmEngine.preferred_batch` on the first protein's token count; 16 for a
    plain callable).
    """
    import torch.distributed as td
    dist = td.is_available() and td.is_initialized()
    if world is None:
        world = td.get_world_size(group) if dist else 1
    if rank is None:
        rank = td.get_rank(group) if dist else 0
    fn = model_or_fn if callable(model_or_fn) and not hasattr(model_or_fn, "forward_sequences") else \
        (lambda toks: model_or_fn.forward_sequences(toks)["shared"])
    mine = shard_indices(n_total, rank, world)
    if batch_size is None:
        batch_size = 16
        eng = getattr(getattr(model_or_fn, "protein_seq_encoder", None), "engine", None)
        if eng is not None and hasattr(eng, "preferred_batch") and len(mine):
            batch_size = eng.preferred_batch(int(token_fn(mine[:1]).shape[1]))
    outs = []
    for s in range(0, len(mine), batch_size):
        idx = mine[s:s + batch_size]
        outs.append(fn(token_fn(idx)))
    local = torch.cat(outs, 0).contiguous()
    if world == 1 and not dist:
        return local[:n_total]
    if local.is_cuda and td.get_backend(group) == "nccl":
        # the one collective of the path, through the C ABI (RCCL over xGMI) on the engine's stream
        return _comm_for(group).all_gather(local)[:n_total]
    full = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    td.all_gather_into_tensor(full, local, group=group)   # host tensors (gloo): torch's collective
    return full[:n_total]


def all_gather_rows(local, n_total, group=None):
    """[n_local, ...] on every rank (equal n_local) -> [n_total, ...] in rank order on every rank: the path's one collective.
    Device tensors under backend "nccl" go through the C ABI (`pcy_allgather`: RCCL over xGMI on the engine's stream, any dtype --
    it moves bytes); host tensors (gloo) through torch's collective; without a process group the input is returned."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()):
        return local[:n_total]
    local = local.contiguous()
    if local.is_cuda and td.get_backend(group) == "nccl":
        return _comm_for(group).all_gather(local)[:n_total]
    world = td.get_world_size(group)
    dev = local.device
    if local.is_cuda:          # a host backend (gloo) does not take device tensors: through host memory (advisor finding, round 4)
        local = local.cpu()
    full = torch.empty(world * local.shape[0], *local.shape[1:], dtype=local.dtype, device=local.device)
    td.all_gather_into_tensor(full, local, group=group)
    return full[:n_total].to(dev)


def run_sharded_rows(row_fn, n_rows, rank=None, world=None, group=None):
    """Split rows [0, n_rows) of a batch across the process group (`shard_indices`: contiguous, rank order, wrap-around tail so
    every rank runs the same count), call row_fn(list_of_row_indices) -> tensor [len(idx), ...] on this rank's rows, and return
    the [n_rows, ...] result on every rank after ONE all-gather.  BASELINE configs[3]: the 32 generation rows over 1 -> 8 GPUs."""
    import torch.distributed as td
    dist = td.is_available() and td.is_initialized()
    if world is None:
        world = td.get_world_size(group) if dist else 1
    if rank is None:
        rank = td.get_rank(group) if dist else 0
    mine = shard_indices(n_rows, rank, world)
    local = row_fn(mine)
    assert local.shape[0] == len(mine), (local.shape, len(mine))
    if world == 1 and not dist:
        return local[:n_rows]
    return all_gather_rows(local, n_rows, group)
