"""KV allocator golden sequence from the REFERENCE class (build container only).

    python tests/golden/gen_golden_kv_alloc.py

Drives lite_llama.executor.kv_cache_manager.KVCacheManager (CPU tensors) through a seeded random
script -- bump allocations, partial frees (fragmentation), general allocations that land on a
contiguous run or fall back to scattered rows, extra references, free_all -- and records, after
every call, the rows returned, the free-row counter and the whole use-count vector.
Saved: plain integer arrays + the script as JSON.
"""

import json
import os
import sys

sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = 300


def build_script(seed=5):
    """(op, arg) list; allocations are numbered in order so frees can name them."""
    rng = np.random.default_rng(seed)
    script, live, n_alloc = [], [], 0
    for phase in range(3):
        for _ in range(6):                                   # append-only phase
            script.append(("index", int(rng.integers(1, 24))))
            live.append(n_alloc)
            n_alloc += 1
        for _ in range(40):                                  # fragmented phase
            r = rng.random()
            if r < 0.35 and live:
                script.append(("free", live.pop(int(rng.integers(0, len(live))))))
            elif r < 0.45 and live:
                a = live[int(rng.integers(0, len(live)))]
                script.append(("free_part", a))              # release every other row of allocation a
            elif r < 0.55 and live:
                script.append(("add_ref", live[int(rng.integers(0, len(live)))]))
            elif r < 0.65:
                script.append(("contiguous", int(rng.integers(1, 40))))
                live.append(n_alloc)
                n_alloc += 1
            elif r < 0.72:
                script.append(("scattered", int(rng.integers(1, 30))))
                live.append(n_alloc)
                n_alloc += 1
            else:
                script.append(("index", int(rng.integers(1, 60))))
                live.append(n_alloc)
                n_alloc += 1
        script.append(("index", 400))                        # larger than the pool
        script.append(("free_all", 0))
        live = []
    return script


def main():
    from lite_llama.executor.kv_cache_manager import KVCacheManager

    m = KVCacheManager(num_layers=1, num_kv_heads=1, head_dim=8, gpu_num_blocks=ROWS, device="cpu")
    script = build_script()
    arrays = {"script": np.array(json.dumps(script)), "rows": np.array(ROWS)}
    allocs = []          # rows of allocation i (or None)
    released = {}        # allocation -> already (partly) released
    for i, (op, arg) in enumerate(script):
        ret = None
        if op in ("index", "contiguous", "scattered"):
            if op == "index":
                try:
                    ret = m.alloc_kvcache_index(arg)
                except AttributeError:       # the reference dereferences None when the pool is short
                    ret = None
            elif op == "contiguous":
                got = m.alloc_contiguous_kvcache(arg)
                ret = None if got is None else got[0]
            else:
                ret = m.alloc_kvcache(arg)
            allocs.append(None if ret is None else ret.clone().long())
        elif op == "free":
            rows = allocs[arg]
            if rows is not None and rows.numel():
                m.free(rows)
        elif op == "free_part":
            rows = allocs[arg]
            if rows is not None and rows.numel() > 1:
                part = rows[::2]
                m.free(part)
                allocs[arg] = rows[1::2]
        elif op == "add_ref":
            rows = allocs[arg]
            if rows is not None and rows.numel():
                uniq = torch.unique(rows)                    # each row named once (duplicates are outside
                m.add_ref(uniq)                              # what add_ref defines, see kv_cache_manager.py)
                allocs[arg] = torch.cat([rows, uniq])        # one more release needed from now on
        elif op == "free_all":
            m.free_all()
        arrays[f"{i}.ret"] = np.array([-1]) if ret is None else ret.numpy().astype(np.int64)
        arrays[f"{i}.none"] = np.array(int(ret is None))
        arrays[f"{i}.free"] = np.array(int(m.can_use_mem_size))
        arrays[f"{i}.state"] = m.kv_mem_use_state.numpy().astype(np.int32).copy()
    path = os.path.join(HERE, "kv_alloc_sequence.npz")
    np.savez_compressed(path, **arrays)
    kinds = {}
    for i, (op, _) in enumerate(script):
        if op in ("index", "contiguous", "scattered"):
            r = arrays[f"{i}.ret"]
            contig = int(arrays[f"{i}.none"]) == 0 and r.size > 0 and bool(np.all(np.diff(r) == 1))
            key = (op, "none" if int(arrays[f"{i}.none"]) else ("run" if contig else "scattered"))
            kinds[key] = kinds.get(key, 0) + 1
    print("wrote", path, os.path.getsize(path), "bytes;", len(script), "ops;", kinds)


if __name__ == "__main__":
    main()
