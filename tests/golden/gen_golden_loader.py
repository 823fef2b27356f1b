"""Checkpoint-loader golden data from the REFERENCE loader (build container only).

    python tests/golden/gen_golden_loader.py

Writes two tiny synthetic HuggingFace checkpoints (data: seeded random tensors under the key names the published
checkpoints use) and what the reference's production load path makes of them:

* ``ckpt/qwen2_tiny/``      fp16, q/k/v biases, two safetensors shards, untied lm_head;
* ``ckpt/qwen3_moe_fp8/``   block-fp8 (e4m3 + ``weight_scale_inv`` per 128 x 128 block) with per-expert keys, loaded
                            raw into 8-bit layers (``dequantize_fp8=False``: config.quant is set).

For each, ``ModelConfig.from_pretrained -> registry class -> materialise_parameters -> load_weights(
hf_weights_iterator(dir))`` (tests/models/test_weight_parity.py:175-181) at TP = 1 and on both ranks of TP = 2 (rank
and world size set in lite_llama.distributed.parallel_state).  ``loader_expected.json`` records every parameter's
dtype, shape and the SHA-256 of its bytes -- bit-exact expectations without a second copy of the tensors -- and the
reference's answers for the key-translation / shard-dimension tables.
"""

import hashlib
import json
import os
import sys

sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "ckpt")

QWEN2 = {
    "model_type": "qwen2", "hidden_size": 64, "intermediate_size": 128, "num_hidden_layers": 2,
    "num_attention_heads": 4, "num_key_value_heads": 2, "vocab_size": 96, "max_position_embeddings": 256,
    "rms_norm_eps": 1e-6, "rope_theta": 10000.0, "tie_word_embeddings": False,
}
MOE_FP8 = {
    "model_type": "qwen3_moe", "hidden_size": 256, "intermediate_size": 512, "num_hidden_layers": 1,
    "num_attention_heads": 8, "num_key_value_heads": 8, "head_dim": 32, "vocab_size": 64,
    "max_position_embeddings": 256, "rms_norm_eps": 1e-6, "rope_theta": 10000.0, "tie_word_embeddings": False,
    "num_experts": 2, "num_experts_per_tok": 2, "moe_intermediate_size": 256, "decoder_sparse_step": 1,
    "mlp_only_layers": [], "norm_topk_prob": True,
    "quantization_config": {"quant_method": "fp8", "fmt": "e4m3", "activation_scheme": "dynamic",
                            "weight_block_size": [128, 128]},
}


def rnd(g, *shape, std=0.05):
    return (torch.randn(*shape, generator=g) * std).to(torch.float16)


def write_qwen2(directory):
    os.makedirs(directory, exist_ok=True)
    g = torch.Generator().manual_seed(11)
    c = QWEN2
    h, i, hq, hkv = c["hidden_size"], c["intermediate_size"], c["num_attention_heads"], c["num_key_value_heads"]
    d = h // hq
    shards = [{}, {}]
    shards[0]["model.embed_tokens.weight"] = rnd(g, c["vocab_size"], h)
    for layer in range(c["num_hidden_layers"]):
        s = shards[layer % 2]
        p = f"model.layers.{layer}."
        for name, n in (("q", hq * d), ("k", hkv * d), ("v", hkv * d)):
            s[f"{p}self_attn.{name}_proj.weight"] = rnd(g, n, h)
            s[f"{p}self_attn.{name}_proj.bias"] = rnd(g, n, std=0.02)
        s[f"{p}self_attn.o_proj.weight"] = rnd(g, h, hq * d)
        s[f"{p}mlp.gate_proj.weight"] = rnd(g, i, h)
        s[f"{p}mlp.up_proj.weight"] = rnd(g, i, h)
        s[f"{p}mlp.down_proj.weight"] = rnd(g, h, i)
        s[f"{p}input_layernorm.weight"] = (1 + rnd(g, h, std=0.1).float()).half()
        s[f"{p}post_attention_layernorm.weight"] = (1 + rnd(g, h, std=0.1).float()).half()
    shards[1]["model.norm.weight"] = (1 + rnd(g, h, std=0.1).float()).half()
    shards[1]["lm_head.weight"] = rnd(g, c["vocab_size"], h)
    for n, s in enumerate(shards):
        save_file(s, os.path.join(directory, f"model-{n + 1:05d}-of-00002.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(directory, "config.json"), "w") as f:
        json.dump(c, f, indent=1)


def fp8_pair(g, n, k):
    """Random e4m3 bytes (no NaN codes) + one fp32 scale per 128 x 128 block."""
    codes = torch.randint(0, 256, (n, k), generator=g, dtype=torch.int32)
    codes = torch.where((codes & 0x7F) == 0x7F, codes & 0x80, codes).to(torch.uint8)
    scale = torch.rand((n + 127) // 128, (k + 127) // 128, generator=g) * 0.01 + 0.001
    return codes.view(torch.float8_e4m3fn), scale.float()


def write_moe_fp8(directory):
    os.makedirs(directory, exist_ok=True)
    g = torch.Generator().manual_seed(12)
    c = MOE_FP8
    h, hq, hkv, d = c["hidden_size"], c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"]
    mi = c["moe_intermediate_size"]
    s = {"model.embed_tokens.weight": rnd(g, c["vocab_size"], h), "model.norm.weight": (1 + rnd(g, h, std=0.1).float()).half(),
         "lm_head.weight": rnd(g, c["vocab_size"], h)}
    p = "model.layers.0."

    def put(key, n, k):
        s[key + ".weight"], s[key + ".weight_scale_inv"] = fp8_pair(g, n, k)

    put(p + "self_attn.q_proj", hq * d, h)
    put(p + "self_attn.k_proj", hkv * d, h)
    put(p + "self_attn.v_proj", hkv * d, h)
    put(p + "self_attn.o_proj", h, hq * d)
    s[p + "self_attn.q_norm.weight"] = (1 + rnd(g, d, std=0.1).float()).half()
    s[p + "self_attn.k_norm.weight"] = (1 + rnd(g, d, std=0.1).float()).half()
    s[p + "input_layernorm.weight"] = (1 + rnd(g, h, std=0.1).float()).half()
    s[p + "post_attention_layernorm.weight"] = (1 + rnd(g, h, std=0.1).float()).half()
    s[p + "mlp.gate.weight"] = rnd(g, c["num_experts"], h)
    for e in range(c["num_experts"]):
        put(f"{p}mlp.experts.{e}.gate_proj", mi, h)
        put(f"{p}mlp.experts.{e}.up_proj", mi, h)
        put(f"{p}mlp.experts.{e}.down_proj", h, mi)
    save_file(s, os.path.join(directory, "model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(directory, "config.json"), "w") as f:
        json.dump(c, f, indent=1)


def digest(t: torch.Tensor):
    raw = t.detach().contiguous().view(torch.uint8).numpy().tobytes()
    return [str(t.dtype).replace("torch.", ""), list(t.shape), hashlib.sha256(raw).hexdigest()]


def reference_load(directory, world, rank):
    import lite_llama.distributed.parallel_state as ps
    from lite_llama.executor.loader import materialise_parameters
    from lite_llama.executor.weight_utils import hf_weights_iterator
    from lite_llama.models.config import ModelConfig
    from lite_llama.models.registry import ModelRegistry

    ps._TP_WORLD_SIZE, ps._TP_RANK = world, rank
    try:
        config = ModelConfig.from_pretrained(directory, max_seq_len=128)
        model = ModelRegistry.resolve(config.model_type).load_class()(config)
        materialise_parameters(model, "cpu")
        model.load_weights(hf_weights_iterator(directory, "cpu", dequantize_fp8=config.quant is None))
        return {name: digest(p.data) for name, p in model.named_parameters()}
    finally:
        ps._TP_WORLD_SIZE, ps._TP_RANK = 1, 0


KEYS = [
    "embed_tokens.weight", "norm.weight", "lm_head.weight", "layers.3.input_layernorm.weight",
    "layers.3.post_attention_layernorm.weight", "layers.3.self_attn.q_proj.weight", "layers.3.self_attn.q_proj.bias",
    "layers.3.self_attn.k_proj.weight", "layers.3.self_attn.v_proj.weight", "layers.3.self_attn.k_proj.bias",
    "layers.3.self_attn.v_proj.bias", "layers.3.self_attn.v_proj.weight_scale_inv", "layers.3.self_attn.o_proj.weight",
    "layers.3.self_attn.q_norm.weight", "layers.3.self_attn.k_norm.weight", "layers.3.mlp.gate.weight",
    "layers.3.mlp.gate_proj.weight", "layers.3.mlp.up_proj.weight", "layers.3.mlp.down_proj.weight",
    "layers.3.mlp.down_proj.weight_scale_inv", "layers.3.mlp.experts.7.gate_proj.weight",
    "layers.3.mlp.experts.7.up_proj.weight", "layers.3.mlp.experts.7.down_proj.weight",
    "layers.3.mlp.experts.7.up_proj.weight_scale_inv", "layers.3.mlp.experts.7.down_proj.weight_scale_inv",
    "layers.11.mlp.experts.0.gate_proj.weight_scale_inv", "rotary_emb.inv_freq", "something.else.weight",
]
SHARD_NAMES = [
    "embed_tokens.weight", "norm_weight", "lm_head_weight", "layers.0.input_layernorm_weight",
    "layers.0.self_attn.q_proj.weight", "layers.0.self_attn.q_proj.bias", "layers.0.self_attn.q_proj.weight_scale_inv",
    "layers.0.self_attn.kv_proj.weight", "layers.0.self_attn.kv_proj.bias", "layers.0.self_attn.o_proj.weight",
    "layers.0.self_attn.o_proj.weight_scale_inv", "layers.0.self_attn.q_norm_weight", "layers.0.mlp.gate_weight",
    "layers.0.mlp.gate_proj.weight", "layers.0.mlp.up_proj.weight", "layers.0.mlp.down_proj.weight",
    "layers.0.mlp.experts.gate_up_proj", "layers.0.mlp.experts.gate_up_proj_scale_inv", "layers.0.mlp.experts.down_proj",
    "layers.0.mlp.experts.down_proj_scale_inv", "vision_tower.layers.0.self_attn.q_proj.weight",
]


def translation_tables():
    from lite_llama.models import weights

    # a destination is identified by the shape and the first element of the view it cuts out of a probe parameter
    probe_stack = torch.arange(12 * 8 * 3, dtype=torch.float32).view(12, 8, 3)  # [experts, rows, cols]
    probe_flat = torch.arange(8 * 3, dtype=torch.float32).view(8, 3)
    table = {}
    for key in KEYS:
        name, dest = weights.translate_text_key(key)
        region = dest(probe_stack if ".experts." in key else probe_flat)
        table[key] = [name, list(region.shape), float(region.reshape(-1)[0])]
    return table, {n: weights.shard_dim(n) for n in SHARD_NAMES}


def main():
    write_qwen2(os.path.join(CKPT, "qwen2_tiny"))
    write_moe_fp8(os.path.join(CKPT, "qwen3_moe_fp8"))
    expected = {"cases": {}}
    for case in ("qwen2_tiny", "qwen3_moe_fp8"):
        runs = {}
        for world, rank in ((1, 0), (2, 0), (2, 1)):
            runs[f"tp{world}_rank{rank}"] = reference_load(os.path.join(CKPT, case), world, rank)
        expected["cases"][case] = runs
        print(case, {k: len(v) for k, v in runs.items()})
    expected["translate_text_key"], expected["shard_dim"] = translation_tables()
    with open(os.path.join(HERE, "loader_expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
