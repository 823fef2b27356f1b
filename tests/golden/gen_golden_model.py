"""Whole-step golden vectors from the REFERENCE model code (build container only).

    TRITON_INTERPRET=1 python tests/golden/gen_golden_model.py

Imports lite_llama.models from /root/reference, builds a tiny random-weight Qwen2-shaped model on
CPU, runs a padded-grid prefill of two sequences, one ``update_kv_index`` + decode step in fp16,
and the same decode step after ``quantize_`` to int4 / int8 / smoothquant / fp8-per-channel.
Saved (plain arrays only): parameters, inputs, per-step metadata, logits, KV pool contents,
greedy tokens.  This pins the per-layer call ORDER and the cache/table side effects of
CausalLM.forward (models/base.py:447-489), not just single kernels.
"""

import copy
import math
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import transformers
    from lite_llama.executor.attention_metadata import AttentionMetadata
    from lite_llama.kernels import update_kv_index
    from lite_llama.models.config import ModelConfig
    from lite_llama.models.qwen2 import Qwen2Model
    from lite_llama.models.quantization import QuantConfig

    torch.manual_seed(7)
    H, I, L, HQ, HKV, V = 256, 512, 2, 4, 2, 512
    D = H // HQ
    cfg = transformers.Qwen2Config(hidden_size=H, intermediate_size=I, num_hidden_layers=L,
                                   num_attention_heads=HQ, num_key_value_heads=HKV, vocab_size=V,
                                   max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=10000.0,
                                   tie_word_embeddings=False)
    model = Qwen2Model(ModelConfig(cfg))
    sd = {}
    for k, v in model.state_dict().items():
        if "norm" in k:
            t = (1 + 0.1 * torch.randn(v.shape)).half()
        elif k.endswith("bias"):
            t = (0.02 * torch.randn(v.shape)).half()
        else:
            t = (0.05 * torch.randn(v.shape)).half()
        sd[k] = t
    model.load_state_dict(sd)
    model.eval()

    lens = [5, 3]
    B, LP, MAXTOK, MAXSEQ = 2, 5, 64, 16
    info = AttentionMetadata()
    info.kv_buffer = [torch.zeros(MAXTOK, 2 * HKV, D, dtype=torch.float16) for _ in range(L)]
    info.b_req_tokens_table = torch.zeros(B, MAXSEQ, dtype=torch.int32)
    info.b_req_idx = torch.arange(B, dtype=torch.int32)
    info.cur_select_index = torch.arange(B * LP, dtype=torch.int32)
    info.b_seq_len = torch.tensor(lens, dtype=torch.int32)
    info.max_actual_seq_len = LP
    info.b_start_loc = torch.arange(B, dtype=torch.int32) * LP
    for i, n in enumerate(lens):
        info.b_req_tokens_table[i, :n] = info.cur_select_index[i * LP : i * LP + n]
    ids = torch.randint(0, V, (B, LP))
    pos = torch.arange(LP).unsqueeze(0).expand(B, LP).contiguous()
    with torch.no_grad():
        logits_p = model(ids, pos, info)
    last = torch.stack([logits_p[i, n - 1] for i, n in enumerate(lens)])
    tok = torch.argmax(last, dim=-1)
    kv_after_prefill = [k.clone() for k in info.kv_buffer]

    # decode_alloc_kv_cache (model_runner.py:200-218)
    info.cur_select_index = torch.arange(B * LP, B * LP + B, dtype=torch.int32)
    info.b_seq_len = info.b_seq_len + 1
    info.max_actual_seq_len += 1
    update_kv_index(info.b_req_tokens_table, info.b_req_idx, info.b_seq_len, info.cur_select_index)
    dpos = torch.tensor(lens).view(B, 1)
    state = copy.deepcopy((info.kv_buffer, info.b_req_tokens_table))
    out = {}
    with torch.no_grad():
        logits_d = model(tok.view(B, 1), dpos, info)
    out["fp16"] = logits_d.clone()
    kv_after_decode = [k.clone() for k in info.kv_buffer]

    for name, q in [("int4", QuantConfig.int4_groupwise(128)), ("int8", QuantConfig.int8_per_channel()),
                    ("smoothquant", QuantConfig.smoothquant_per_channel()), ("fp8", QuantConfig.fp8_per_channel())]:
        m2 = Qwen2Model(ModelConfig(cfg))
        m2.load_state_dict(sd)
        m2.eval()
        m2.quantize_(q)
        info.kv_buffer = [k.clone() for k in state[0]]
        info.b_req_tokens_table = state[1].clone()
        with torch.no_grad():
            out[name] = m2(tok.view(B, 1), dpos, info).clone()

    arrays = {f"param.{k}": v.numpy() for k, v in sd.items()}
    arrays.update(
        geometry=np.array([H, I, L, HQ, HKV, D, V]), lens=np.array(lens), prompt_ids=ids.numpy(),
        logits_prefill_last=last.numpy(), first_tokens=tok.numpy(),
        table_after=info.b_req_tokens_table.numpy() if False else state[1].numpy(),
        decode_positions=dpos.numpy(),
    )
    for i in range(L):
        arrays[f"kv_prefill.{i}"] = kv_after_prefill[i].numpy()
        arrays[f"kv_decode.{i}"] = kv_after_decode[i].numpy()
    for k, v in out.items():
        arrays[f"logits_decode.{k}"] = v.numpy()
    path = os.path.join(HERE, "model_step_qwen2_tiny.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    print("fp16 decode argmax", out["fp16"].argmax(-1).flatten().tolist(), "int4", out["int4"].argmax(-1).flatten().tolist())


if __name__ == "__main__":
    main()
