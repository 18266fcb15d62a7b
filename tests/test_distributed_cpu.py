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


# ---- row-split generation (BASELINE configs[3]: the 32 rows of a batch across 1 -> 8 GPUs, one gather of the token ids) ----
def _fake_generate(idx):
    # "token ids" that are a function of the row index only: [len(idx), 1, 6] int32
    base = torch.tensor(idx, dtype=torch.int32).view(-1, 1, 1)
    return base * 100 + torch.arange(6, dtype=torch.int32).view(1, 1, 6)


def _rows_worker(rank, world, port, n_rows, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    from procyon_amd.distributed import run_sharded_rows
    td.init_process_group("gloo", rank=rank, world_size=world)
    seen = []
    out = run_sharded_rows(lambda idx: (seen.append(list(idx)), _fake_generate(idx))[1], n_rows)
    ret[rank] = (bool(torch.equal(out, _fake_generate(list(range(n_rows))))), seen[0])
    td.destroy_process_group()


def test_world2_gloo_row_split_generation_gathers_token_ids_in_row_order():
    """`run_sharded_rows` (the reference's pattern: /root/reference/procyon/data/samplers.py:154-196 contiguous chunks in rank order +
    the gather of /root/reference/procyon/training/trainIT.py:1594-1610): every rank runs ceil(n / W) rows, the gathered matrix is the
    single-process result, also when n does not divide (wrap-around tail)."""
    for n_rows in (32, 7):
        mgr = mp.get_context("spawn").Manager()
        ret = mgr.dict()
        mp.spawn(_rows_worker, args=(2, _free_port(), n_rows, ret), nprocs=2, join=True)
        r = dict(ret)
        assert r[0][0] and r[1][0], (n_rows, r)
        per = -(-n_rows // 2)
        assert r[0][1] == list(range(per)) and r[1][1][: n_rows - per] == list(range(per, n_rows))


def test_config4_rows_of_a_rank_are_the_rows_of_the_whole_batch():
    """A rank's sub-batch of config 4 holds exactly the proteins and prompts the single-GPU batch gives those row indices."""
    from procyon_amd.workloads import config4_inputs
    full, lens, plen = config4_inputs(ragged=True, rows=8)
    sub, lens2, plen2 = config4_inputs(ragged=True, rows=8, subset=[4, 5, 6, 7])
    a, b = full(), sub()
    assert lens == lens2 and plen == plen2
    assert torch.equal(a["data"]["seq"][4:], b["data"]["seq"]) and a["instructions"][4:] == b["instructions"]
    assert b["input"]["seq"] == [[0], [1], [2], [3]]
