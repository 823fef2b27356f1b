"""Whole-step golden vectors for the other model families, from the REFERENCE model code
(build container only).

    TRITON_INTERPRET=1 python tests/golden/gen_golden_families.py

Same protocol as gen_golden_model.py (padded-grid prefill of two sequences, ``update_kv_index``,
one decode step in fp16 and after ``quantize_``), for three more shapes of the decode-step caller:

* ``llama``      -- no qkv bias, rope theta 5e5 (models/llama.py)
* ``qwen3``      -- per-head q/k RMSNorm, head_dim decoupled from hidden/heads (models/qwen3.py)
* ``qwen3_moe``  -- qwen3 attention + top-k routed experts (models/qwen3_moe.py:60-111)

Saved: plain arrays only (parameters, inputs, logits, KV pool contents, greedy tokens).
"""

import copy
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _families():
    import transformers as T

    common = dict(num_hidden_layers=2, vocab_size=384, max_position_embeddings=4096, rms_norm_eps=1e-6,
                  tie_word_embeddings=False)
    return {
        "llama": dict(
            cfg=T.LlamaConfig(hidden_size=128, intermediate_size=256, num_attention_heads=4, num_key_value_heads=2,
                              rope_theta=500000.0, attention_bias=False, **common),
            quants=["int4", "int8", "smoothquant", "fp8"], seed=11),
        "qwen3": dict(
            cfg=T.Qwen3Config(hidden_size=128, intermediate_size=256, num_attention_heads=4, num_key_value_heads=2,
                              head_dim=64, rope_theta=1000000.0, attention_bias=False, **common),
            quants=["int4", "fp8"], seed=12),
        "qwen3_moe": dict(
            cfg=T.Qwen3MoeConfig(hidden_size=128, intermediate_size=256, num_attention_heads=4, num_key_value_heads=2,
                                 head_dim=64, rope_theta=1000000.0, attention_bias=False, num_experts=8,
                                 num_experts_per_tok=2, moe_intermediate_size=128, norm_topk_prob=True,
                                 decoder_sparse_step=1, mlp_only_layers=[], **common),
            quants=["int8", "smoothquant", "fp8"], seed=13),
    }


def _rope_theta(cfg):
    rp = getattr(cfg, "rope_parameters", None)
    if isinstance(rp, dict) and "rope_theta" in rp:
        return float(rp["rope_theta"])
    return float(getattr(cfg, "rope_theta", 10000.0))


def generate(name, spec):
    from lite_llama.executor.attention_metadata import AttentionMetadata
    from lite_llama.kernels import update_kv_index
    from lite_llama.models.config import ModelConfig
    from lite_llama.models.quantization import QuantConfig
    from lite_llama.models.registry import ModelRegistry

    cfg = spec["cfg"]
    mc = ModelConfig(cfg)
    Model = ModelRegistry.resolve(cfg.model_type).load_class()

    torch.manual_seed(spec["seed"])
    model = Model(mc)
    sd = {}
    for k, v in model.state_dict().items():
        if "norm" in k:
            t = (1 + 0.1 * torch.randn(v.shape)).half()
        elif k.endswith("bias"):
            t = (0.02 * torch.randn(v.shape)).half()
        elif "experts" in k:
            t = (torch.randn(v.shape) / (v.shape[-1] ** 0.5)).half()
        elif k.endswith("gate_weight"):
            t = (0.2 * torch.randn(v.shape)).half()  # spread router logits so top-k is unambiguous
        else:
            t = (0.05 * torch.randn(v.shape)).half()
        sd[k] = t
    model.load_state_dict(sd)
    model.eval()

    L, HKV, D, V = mc.num_layers, mc.num_kv_heads, mc.head_dim, mc.vocab_size
    lens = [6, 4]
    B, LP, MAXTOK, MAXSEQ = 2, 6, 64, 16
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

    info.cur_select_index = torch.arange(B * LP, B * LP + B, dtype=torch.int32)
    info.b_seq_len = info.b_seq_len + 1
    info.max_actual_seq_len += 1
    update_kv_index(info.b_req_tokens_table, info.b_req_idx, info.b_seq_len, info.cur_select_index)
    dpos = torch.tensor(lens).view(B, 1)
    state = copy.deepcopy((info.kv_buffer, info.b_req_tokens_table))
    out = {}
    with torch.no_grad():
        out["fp16"] = model(tok.view(B, 1), dpos, info).clone()
    kv_after_decode = [k.clone() for k in info.kv_buffer]

    qcfg = {"int4": QuantConfig.int4_groupwise(128), "int8": QuantConfig.int8_per_channel(),
            "smoothquant": QuantConfig.smoothquant_per_channel(), "fp8": QuantConfig.fp8_per_channel()}
    for q in spec["quants"]:
        m2 = Model(mc)
        m2.load_state_dict(sd)
        m2.eval()
        m2.quantize_(qcfg[q])
        info.kv_buffer = [k.clone() for k in state[0]]
        info.b_req_tokens_table = state[1].clone()
        with torch.no_grad():
            out[q] = m2(tok.view(B, 1), dpos, info).clone()

    moe = [int(getattr(cfg, "num_experts", 0) or 0), int(getattr(cfg, "num_experts_per_tok", 0) or 0),
           int(getattr(cfg, "moe_intermediate_size", 0) or 0), int(bool(getattr(cfg, "norm_topk_prob", False)))]
    if name != "qwen3_moe":
        moe = [0, 0, 0, 0]
    arrays = {f"param.{k}": v.numpy() for k, v in sd.items()}
    arrays.update(
        geometry=np.array([mc.hidden_size, mc.intermediate_size, L, mc.num_heads, HKV, D, V]),
        rope_theta=np.array(_rope_theta(mc.text_config)), eps=np.array(float(cfg.rms_norm_eps)),
        qk_norm=np.array(int(name != "llama")), moe=np.array(moe),
        lens=np.array(lens), prompt_ids=ids.numpy(), logits_prefill_last=last.numpy(), first_tokens=tok.numpy(),
        table_after=state[1].numpy(), decode_positions=dpos.numpy(),
    )
    for i in range(L):
        arrays[f"kv_prefill.{i}"] = kv_after_prefill[i].numpy()
        arrays[f"kv_decode.{i}"] = kv_after_decode[i].numpy()
    for k, v in out.items():
        arrays[f"logits_decode.{k}"] = v.numpy()
    path = os.path.join(HERE, f"model_step_{name}_tiny.npz")
    np.savez_compressed(path, **arrays)
    print(name, "wrote", path, os.path.getsize(path) // 1024, "KiB; theta", arrays["rope_theta"],
          "argmax", {k: v.argmax(-1).flatten().tolist() for k, v in out.items()})


if __name__ == "__main__":
    want = sys.argv[1:]
    for name, spec in _families().items():
        if not want or name in want:
            generate(name, spec)
