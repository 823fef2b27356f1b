"""Block-granular KV paging on the device (extension -- the reference's pool is token-granular, TODO at
executor/kv_cache_manager.py:211).  Held to (1) the plain-python model oracle/kv_paged.py, exactly, over random
admit / append / release sequences incl. exhaustion, and (2) the reference's observable behaviour: a decode that gets
its rows from the paged pool generates the tokens of the decode on the reference's bump-allocated rows, eager and
inside the captured step."""

import numpy as np
import pytest
import torch

from lite_llama_amd.executor import DecodeEngine
from lite_llama_amd.executor.paged_kv import PagedKVPool
from lite_llama_amd.model import CausalLM, tiny_geometry
from lite_llama_amd.quantization import QuantConfig
from oracle.kv_paged import PagedPoolModel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pool(num_blocks, bs, max_req, max_len):
    return PagedKVPool(1, num_blocks, bs, 1, 32, DEV, max_req, max_len), PagedPoolModel(num_blocks, bs, max_req, max_len)


def _same(pool, table, model):
    st = pool.state.cpu().numpy()
    assert int(st[0]) == model.free and int(st[1]) == model.err
    assert np.array_equal(pool.req_blocks.cpu().numpy(), model.req_blocks)
    got_bt, want_bt = pool.block_table.cpu().numpy(), model.block_table
    for r in range(len(model.req_blocks)):  # only the held prefix of a row is defined
        n = int(model.req_blocks[r])
        assert np.array_equal(got_bt[r, :n], want_bt[r, :n]), r
    assert np.array_equal(pool.free_stack.cpu().numpy()[: model.free], np.array(model.stack[: model.free], dtype=np.int32))
    assert np.array_equal(table.cpu().numpy(), model.token_table)


@pytest.mark.parametrize("bs", [1, 16, 64])
def test_fresh_pool_hands_out_ascending_blocks(bs):
    pool, model = _pool(40, bs, 8, 4 * bs + 3)
    table = torch.zeros(8, 4 * bs + 3, dtype=torch.int32, device=DEV)
    req = torch.tensor([3, 0, 5], dtype=torch.int32, device=DEV)
    lens = torch.tensor([2 * bs + 1, 1, bs], dtype=torch.int32, device=DEV)
    lp = 2 * bs + 2
    sel = pool.admit(req, lens, lp, table)
    want = model.extend([3, 0, 5], lens.tolist(), lp, 0)
    assert np.array_equal(sel.cpu().numpy(), want)
    _same(pool, table, model)
    assert pool.block_table[3, :3].tolist() == [0, 1, 2] and pool.block_table[0, 0].item() == 3
    assert pool.free_blocks == 39 - 5 and pool.error == 0
    # pad positions of the grid name rows of the junk block
    s = sel.view(3, lp)
    assert (s[1, 1:] >= 39 * bs).all() and (s[0, : 2 * bs + 1] < 39 * bs).all()


def test_random_sequences_match_the_model():
    rng = np.random.default_rng(4)
    bs, max_req, max_len, nb = 8, 12, 70, 60
    pool, model = _pool(nb, bs, max_req, max_len)
    table = torch.zeros(max_req, max_len, dtype=torch.int32, device=DEV)
    lens = np.zeros(max_req, dtype=np.int64)
    live = set()
    for step in range(120):
        op = rng.integers(0, 10)
        idle = [r for r in range(max_req) if r not in live]
        if op < 3 and idle:                                  # admit a few requests with a padded grid
            k = int(rng.integers(1, min(4, len(idle)) + 1))
            reqs = [int(r) for r in rng.choice(idle, k, replace=False)]
            ln = [int(rng.integers(1, 30)) for _ in reqs]
            lp = max(ln) + int(rng.integers(0, 3))
            sel = pool.admit(torch.tensor(reqs, dtype=torch.int32, device=DEV), torch.tensor(ln, dtype=torch.int32, device=DEV),
                             lp, table)
            err_before = model.err
            want = model.extend(reqs, ln, lp, 0)
            assert np.array_equal(sel.cpu().numpy(), want), step
            if model.err == err_before or True:
                for r, n in zip(reqs, ln):
                    if model.req_blocks[r] * bs >= n:
                        live.add(r)
                        lens[r] = n
        elif op < 8 and live:                                # one decode step for a subset of the live requests
            reqs = sorted(int(r) for r in rng.choice(sorted(live), int(rng.integers(1, len(live) + 1)), replace=False))
            reqs = [r for r in reqs if lens[r] + 1 <= max_len]
            if not reqs:
                continue
            new = [int(lens[r] + 1) for r in reqs]
            sel = pool.append(torch.tensor(reqs, dtype=torch.int32, device=DEV), torch.tensor(new, dtype=torch.int32, device=DEV), table)
            want = model.extend(reqs, new, 1, 1)
            assert np.array_equal(sel.cpu().numpy(), want), step
            for r, n in zip(reqs, new):
                if model.req_blocks[r] * bs >= n:
                    lens[r] = n
        elif live:                                           # finished requests give their blocks back
            reqs = [int(r) for r in rng.choice(sorted(live), int(rng.integers(1, min(3, len(live)) + 1)), replace=False)]
            pool.release(torch.tensor(reqs, dtype=torch.int32, device=DEV))
            model.release(reqs)
            for r in reqs:
                live.discard(r)
                lens[r] = 0
        _same(pool, table, model)
    assert model.free + int(model.req_blocks.sum()) == nb - 1   # blocks are conserved


def test_exhaustion_is_all_or_nothing_and_flagged():
    pool, model = _pool(6, 4, 4, 40)   # 5 usable blocks
    table = torch.zeros(4, 40, dtype=torch.int32, device=DEV)
    req = torch.tensor([0, 1], dtype=torch.int32, device=DEV)
    sel = pool.admit(req, torch.tensor([9, 13], dtype=torch.int32, device=DEV), 13, table)   # needs 3 + 4 = 7 > 5
    assert np.array_equal(sel.cpu().numpy(), model.extend([0, 1], [9, 13], 13, 0))
    assert pool.error == 1 and pool.free_blocks == 5 and pool.req_blocks.sum().item() == 0
    assert (sel >= 5 * 4).all()                      # everything points at the junk block: the kernels stay in bounds
    sel = pool.admit(req[:1], torch.tensor([9], dtype=torch.int32, device=DEV), 9, table)   # 3 blocks fit
    assert pool.free_blocks == 2 and sel.tolist() == list(range(9))
    # a request that outgrows its block-table row is clamped and flagged
    pool2, _ = _pool(40, 4, 2, 10)                   # 3 blocks per request
    t2 = torch.zeros(2, 16, dtype=torch.int32, device=DEV)
    pool2.admit(torch.tensor([0], dtype=torch.int32, device=DEV), torch.tensor([14], dtype=torch.int32, device=DEV), 14, t2)
    assert pool2.error == 2 and pool2.req_blocks[0].item() == 3


def _model():
    geo = tiny_geometry(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
                        vocab_size=512, qkv_bias=True)
    quant = QuantConfig.int4_groupwise(128)
    return CausalLM(geo, quant).init_synthetic(seed=21, quant=quant, device=DEV)


@pytest.mark.parametrize("block_size", [1, 16])
def test_paged_decode_generates_the_bump_allocators_tokens(block_size):
    """Prefill of ragged prompts + 40 decode steps (crossing block boundaries at different steps per sequence): the
    paged engine, eager and captured, against the engine on the reference's bump-allocated rows."""
    model = _model()
    ids = torch.randint(0, 512, (3, 21), generator=torch.Generator().manual_seed(8)).to(DEV)
    lens = torch.tensor([21, 9, 16], device=DEV)
    ref_eng = DecodeEngine(model, max_batch=3, max_seq_len=80)
    first = ref_eng.prefill(ids, lens)
    want = ref_eng.decode(first, 40, use_graph=False).cpu()
    for use_graph in (False, True):
        eng = DecodeEngine(model, max_batch=3, max_seq_len=80, kv_block_size=block_size)
        f2 = eng.prefill(ids, lens)
        assert torch.equal(f2, first)
        got = eng.decode(f2, 40, use_graph=use_graph).cpu()
        assert torch.equal(got, want), (block_size, use_graph)
        assert eng.pool.error == 0
        held = eng.pool.req_blocks.cpu().tolist()
        # after the last step the engine has already named the row of the NEXT token (decode_alloc_kv_cache runs after
        # every forward): prompt + 40 generated + 1
        assert held == [-(-(int(n) + 41) // block_size) for n in lens.tolist()]
        # a sequence's rows are consecutive inside a block
        t = eng.info.b_req_tokens_table.cpu()
        for r, n in enumerate(lens.tolist()):
            rows = t[r, : n + 41]
            for b0 in range(0, n + 41 - block_size + 1, block_size):
                blk = rows[b0 : b0 + block_size]
                assert torch.equal(blk, blk[0] + torch.arange(block_size, dtype=blk.dtype))
