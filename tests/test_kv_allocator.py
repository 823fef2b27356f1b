"""KV row allocator: the device-side search and reference counts against an op
sequence recorded from the reference's KVCacheManager, and against the oracle at sizes that cross
the scan's word (64 rows) and chunk (4096 rows) boundaries.  Integer work: exact.

Also the properties the reference's tests/executor/test_kv_cache_manager.py checks (fresh pool,
over-capacity, contiguous vs fragmented, second reference, bump fast path and its retirement).
"""

import json

import numpy as np
import pytest
import torch

from oracle.kv_alloc import OracleKVAllocator
from tests import _golden as G


def _gold():
    d = np.load(G.GOLDEN_DIR + "/kv_alloc_sequence.npz")
    return d, json.loads(str(d["script"])), int(d["rows"])


def _replay(make, to_np, tensor):
    """Run the recorded script on a manager; compare returns, free counter and use counts per op."""
    d, script, rows = _gold()
    m = make(rows)
    allocs = []
    for i, (op, arg) in enumerate(script):
        ret = None
        if op == "index":
            ret = m.alloc_kvcache_index(arg)
        elif op == "contiguous":
            got = m.alloc_contiguous_kvcache(arg)
            ret = None if got is None else got[0]
        elif op == "scattered":
            ret = m.alloc_kvcache(arg)
        elif op == "free" and allocs[arg] is not None and len(allocs[arg]):
            m.release_ref(tensor(allocs[arg]))
        elif op == "free_part" and allocs[arg] is not None and len(allocs[arg]) > 1:
            m.release_ref(tensor(allocs[arg][::2]))
            allocs[arg] = allocs[arg][1::2]
        elif op == "add_ref" and allocs[arg] is not None and len(allocs[arg]):
            uniq = np.unique(allocs[arg])
            m.add_ref(tensor(uniq))
            allocs[arg] = np.concatenate([allocs[arg], uniq])
        elif op == "free_all":
            m.free_all()
        if op in ("index", "contiguous", "scattered"):
            allocs.append(None if ret is None else to_np(ret).astype(np.int64))
            assert (ret is None) == bool(int(d[f"{i}.none"])), (i, op, arg)
            if ret is not None:
                assert np.array_equal(allocs[-1], d[f"{i}.ret"]), (i, op, arg)
        free, state = m.free_rows(), m.counts()
        assert free == int(d[f"{i}.free"]), (i, op, arg)
        assert np.array_equal(state, d[f"{i}.state"]), (i, op, arg)


class _OracleAdapter(OracleKVAllocator):
    def free_rows(self):
        return self.free

    def counts(self):
        return self.state


def test_oracle_allocator_matches_reference_sequence():
    _replay(_OracleAdapter, lambda r: np.asarray(r), lambda a: np.asarray(a))


def _manager(rows):
    from lite_llama_amd.executor import KVCacheManager

    class M(KVCacheManager):
        def free_rows(self):
            return self.can_use_mem_size

        def counts(self):
            return self.kv_mem_use_state.cpu().numpy()

    return M(num_layers=2, num_kv_heads=2, head_dim=8, gpu_num_blocks=rows, device="cuda")


@pytest.mark.gpu
def test_device_allocator_matches_reference_sequence():
    _replay(_manager, lambda r: r.cpu().numpy(), lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,seed", [(20000, 0), (9000, 1), (4096 * 3, 2), (70, 3)])
def test_device_search_across_word_and_chunk_boundaries(rows, seed):
    """Random fragmentation with long free stretches; every general allocation must pick the rows
    the oracle picks (first run, else first free rows), incl. needs larger than a chunk."""
    rng = np.random.default_rng(seed)
    m, o = _manager(rows), OracleKVAllocator(rows)
    # occupy everything with the cursor, then punch holes of assorted lengths
    assert np.array_equal(m.alloc_kvcache_index(rows).cpu().numpy(), o.alloc_kvcache_index(rows))
    pos = 0
    holes = []
    while pos < rows:
        gap = int(rng.choice([1, 2, 3, 63, 64, 65, 130, 500, 4096, 5000])) if rng.random() < 0.5 else int(rng.integers(1, 40))
        used = int(rng.integers(1, 80))
        holes.append(np.arange(pos, min(pos + gap, rows)))
        pos += gap + used
    hole_rows = np.concatenate(holes)
    m.release_ref(torch.from_numpy(hole_rows).cuda())
    o.release_ref(hole_rows)
    assert m.can_use_mem_size == o.free
    needs = [1, 2, 3, 5, 17, 63, 64, 65, 100, 129, 500, 4095, 4096, 4097, 5000, 1, 7, 64]
    rng.shuffle(needs)
    for need in needs:
        for kind in ("index", "scattered", "contiguous"):
            if kind == "index":
                a, b = m.alloc_kvcache_index(need), o.alloc_kvcache_index(need)
            elif kind == "scattered":
                a, b = m.alloc_kvcache(need), o.alloc_kvcache(need)
            else:
                a, b = m.alloc_contiguous_kvcache(need), o.alloc_contiguous_kvcache(need)
                assert (a is None) == (b is None), (need, kind)
                if a is not None:
                    assert (a[1], a[2]) == (b[1], b[2])
                    a, b = a[0], b[0]
            assert (a is None) == (b is None), (need, kind)
            if a is not None:
                assert np.array_equal(a.cpu().numpy().astype(np.int64), np.asarray(b).astype(np.int64)), (need, kind)
            assert m.can_use_mem_size == o.free
        # give half of what was taken back so that later needs still find room
        back = np.nonzero(o.state == 1)[0]
        back = back[rng.random(back.size) < 0.3]
        if back.size:
            m.release_ref(torch.from_numpy(back).cuda())
            o.release_ref(back)
    assert np.array_equal(m.kv_mem_use_state.cpu().numpy(), o.state)


@pytest.mark.gpu
def test_pool_properties():
    m = _manager(32)
    assert m.can_use_mem_size == 32 and int(m.kv_mem_use_state.sum()) == 0
    assert len(m.gpu_kv_buffer) == 2 and tuple(m.gpu_kv_buffer[0].shape) == (32, 4, 8)
    assert m.alloc_kvcache(33) is None and m.alloc_contiguous_kvcache(33) is None and m.can_use_mem_size == 32
    a = m.alloc_kvcache_index(8)
    assert a.dtype == torch.int32 and a.tolist() == list(range(8)) and m.can_use_mem_size == 24
    b = m.alloc_kvcache_index(8)
    assert b.tolist() == list(range(8, 16))
    m.free(a[2:5])                                   # a partial free retires the cursor
    assert m.can_use_mem_size == 19
    c = m.alloc_kvcache_index(3)                     # the hole fits exactly: first run wins
    assert c.tolist() == [2, 3, 4]
    m.free(torch.tensor([0, 5], device="cuda"))
    d = m.alloc_kvcache_index(17)                    # 16..31 is the only run of 16; 17 needs scattered rows
    assert d.tolist() == [0, 5] + list(range(16, 31))
    assert m.can_use_mem_size == 1
    m.add_ref(b)                                     # second reference keeps rows alive through one release
    m.free(b)
    assert m.can_use_mem_size == 1 and int(m.kv_mem_use_state[8]) == 1
    m.free(b)
    assert m.can_use_mem_size == 9
    m.free_all()
    assert m.can_use_mem_size == 32 and m.alloc_kvcache_index(4).tolist() == [0, 1, 2, 3]
    m.free_all()
    m.claim(10)
    assert m.can_use_mem_size == 22 and m.alloc_kvcache_index(2).tolist() == [10, 11]
    with pytest.raises(ValueError):
        m.claim(100)


@pytest.mark.gpu
def test_general_allocation_does_not_read_back():
    """After a partial free the hot entry decides and fills on the device: the host bound admits the
    request, and the only synchronising calls are the explicit reads in this test."""
    m = _manager(4096)
    m.alloc_kvcache_index(4096)
    m.free(torch.arange(100, 400, device="cuda"))
    assert m.can_use_mem_size == 300                 # one read-back: bound exact again
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        rows = m.alloc_kvcache_index(50)
        rows2 = m.alloc_kvcache_index(250)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert rows.tolist() == list(range(100, 150)) and rows2.tolist() == list(range(150, 400))
    assert m.alloc_kvcache_index(1) is None
