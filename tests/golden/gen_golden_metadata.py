"""Per-step metadata golden vectors from the REFERENCE executor code (build container only).

    python tests/golden/gen_golden_metadata.py

Two scripted scenarios, every tensor the attention kernels would read recorded after each call:

* ``slot``   -- lite_llama.executor.slot_batch.SlotBatch on a 5-slot x 16-row table with captured
  batch sizes (1, 2, 4): prefill of two requests, steady-state decode, a third request joining
  (padding 3 -> 4 with the filler slot), one leaving, and a single survivor.
* ``oneshot`` -- ModelRunner.prefill_alloc_kv_cache / decode_alloc_kv_cache (bump allocator +
  ``update_kv_index``) for prompt lengths (5, 3, 4) and four decode steps; the two methods are run
  unbound on a stand-in object carrying only the attributes they touch.

Saved: plain integer arrays only.
"""

import os
import sys
import types

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SLOT_SCRIPT = [
    ("prefill", [0, 1], [5, 3]),
    ("decode", [0, 1], [6, 4]),
    ("decode", [0, 1], [7, 5]),
    ("prefill", [2], [4]),
    ("decode", [0, 1, 2], [8, 6, 5]),
    ("decode", [0, 1, 2], [9, 7, 6]),
    ("decode", [0, 2], [10, 7]),
    ("decode", [0, 2], [11, 8]),
    ("decode", [2], [9]),
    ("decode", [2], [10]),
]
GRAPH_SIZES = (1, 2, 4)
ONESHOT_LENS = [5, 3, 4]
ONESHOT_STEPS = 4


def _record(arrays, tag, info):
    arrays[f"{tag}.b_req_idx"] = info.b_req_idx.numpy().astype(np.int64)
    arrays[f"{tag}.b_seq_len"] = info.b_seq_len.numpy().astype(np.int64)
    arrays[f"{tag}.cur_select_index"] = info.cur_select_index.numpy().astype(np.int64)
    arrays[f"{tag}.max_actual_seq_len"] = np.array(int(info.max_actual_seq_len))
    if info.b_start_loc is not None:
        arrays[f"{tag}.b_start_loc"] = info.b_start_loc.numpy().astype(np.int64)


def slot_scenario(arrays):
    from lite_llama.executor.attention_metadata import AttentionMetadata
    from lite_llama.executor.kv_cache_manager import KVCacheManager
    from lite_llama.executor.slot_batch import SlotBatch

    slots, row_len = 5, 16
    kvm = KVCacheManager(num_layers=1, num_kv_heads=1, head_dim=8, gpu_num_blocks=slots * row_len, device="cpu")
    info = AttentionMetadata()
    info.kv_buffer = kvm.gpu_kv_buffer
    info.b_req_tokens_table = torch.zeros(slots, row_len, dtype=torch.int32)

    def graph_batch_size(n):
        return next((b for b in GRAPH_SIZES if b >= n), n)

    runner = types.SimpleNamespace(atten_info=info, device="cpu", max_seq_len=row_len, b_req_tokens_table=info.b_req_tokens_table,
                                   kv_cache_manager=kvm, graph_batch_size=graph_batch_size)
    sb = SlotBatch(runner)
    arrays["slot.table"] = info.b_req_tokens_table.numpy().copy()
    arrays["slot.num_slots"] = np.array(sb.num_slots)
    arrays["slot.free_rows_after_claim"] = np.array(kvm.can_use_mem_size)
    for i, (kind, s, lens) in enumerate(SLOT_SCRIPT):
        if kind == "prefill":
            sb.begin_prefill(s, lens)
            padded = len(s)
        else:
            padded = sb.begin_decode(s, lens)
            arrays[f"slot.{i}.positions"] = (sb.seq_lens.view(-1, 1) - 1).numpy().astype(np.int64)
        arrays[f"slot.{i}.padded"] = np.array(padded)
        _record(arrays, f"slot.{i}", info)


def oneshot_scenario(arrays):
    from lite_llama.executor.attention_metadata import AttentionMetadata
    from lite_llama.executor.kv_cache_manager import KVCacheManager
    from lite_llama.executor.model_runner import ModelRunner

    b, lp, max_seq = len(ONESHOT_LENS), max(ONESHOT_LENS), 16
    kvm = KVCacheManager(num_layers=1, num_kv_heads=1, head_dim=8, gpu_num_blocks=64, device="cpu")
    info = AttentionMetadata()
    info.kv_buffer = kvm.gpu_kv_buffer
    info.b_req_tokens_table = torch.zeros(b, max_seq, dtype=torch.int32)
    stub = types.SimpleNamespace(atten_info=info, kv_cache_manager=kvm, device="cpu")
    stub._init_req_tokens_table = types.MethodType(ModelRunner._init_req_tokens_table, stub)
    kvm.free_all()
    ModelRunner.prefill_alloc_kv_cache(stub, lp, torch.tensor(ONESHOT_LENS, dtype=torch.int32),
                                       torch.arange(b, dtype=torch.int32))
    _record(arrays, "oneshot.prefill", info)
    arrays["oneshot.prefill.table"] = info.b_req_tokens_table.numpy().copy()
    for step in range(ONESHOT_STEPS):
        ModelRunner.decode_alloc_kv_cache(stub, b)
        _record(arrays, f"oneshot.{step}", info)
        arrays[f"oneshot.{step}.table"] = info.b_req_tokens_table.numpy().copy()
    arrays["oneshot.free_rows"] = np.array(kvm.can_use_mem_size)


if __name__ == "__main__":
    import json

    arrays = {"script": np.array(json.dumps(dict(slot=SLOT_SCRIPT, graph_sizes=GRAPH_SIZES, oneshot_lens=ONESHOT_LENS,
                                                 oneshot_steps=ONESHOT_STEPS)))}
    slot_scenario(arrays)
    oneshot_scenario(arrays)
    path = os.path.join(HERE, "step_metadata.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes,", len(arrays), "arrays")
    for k in sorted(arrays):
        if k.startswith("slot.4") or k.startswith("oneshot.1"):
            print(k, arrays[k].tolist())
