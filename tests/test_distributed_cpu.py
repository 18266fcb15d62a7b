"""CPU, world_size 2 over gloo: the sharded retrieval path (contiguous chunks, wrap-around padding, one all-gather,
original order restored) -- procyon_amd/distributed.py."""
import os
import socket

import torch
import torch.multiprocessing as mp

from procyon_amd.distributed import embed_sharded, shard_indices


def test_shard_indices_match_sequential_sampler():
    # samplers.py:178-196: pad by wrap-around, contiguous slices in rank order
    assert shard_indices(10, 0, 4) == [0, 1, 2] and shard_indices(10, 3, 4) == [9, 0, 1]
    assert shard_indices(8, 1, 2) == [4, 5, 6, 7]
    assert sum((shard_indices(101, r, 8) for r in range(8)), [])[:101] == list(range(101))


def _fake_embed(tokens):
    # deterministic "embedding": a function of the token row only
    t = tokens.float()
    return torch.stack([t.sum(1), (t * torch.arange(t.shape[1])).sum(1), t[:, 1], t[:, -2]], 1).bfloat16()


def _token_fn(idx):
    g = torch.Generator().manual_seed(1234)
    table = torch.randint(4, 24, (64, 12), generator=g)
    return table[torch.tensor(idx)]


def _worker(rank, world, port, n_total, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    out = embed_sharded(_fake_embed, _token_fn, n_total, batch_size=3)
    ref = _fake_embed(_token_fn(list(range(n_total))))
    ok = torch.equal(out, ref)
    ret[rank] = ok
    td.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_world2_gloo_allgather_restores_order():
    for n_total in (11, 16):
        mgr = mp.get_context("spawn").Manager()
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), n_total, ret), nprocs=2, join=True)
        assert dict(ret) == {0: True, 1: True}, (n_total, dict(ret))


def test_single_process_path():
    out = embed_sharded(_fake_embed, _token_fn, 7, batch_size=4, rank=0, world=1)
    assert torch.equal(out, _fake_embed(_token_fn(list(range(7)))))
