"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's decode-step caller
(lite_llama/models/base.py:81-129,204-245,263-264,299-319,447-489) on top of the oracle kernels.

Pinned by ``tests/golden/model_step_{qwen2,llama,qwen3,qwen3_moe}_tiny.npz`` (generated from the
reference model code by tests/golden/gen_golden_model.py and gen_golden_families.py).  Only tests /
smoke / bench's cpu_baseline may import it.
"""

from __future__ import annotations

import math

import torch

from . import oracle as O

_LOG2E = 1.4426950408889634


class OracleModel:
    """Qwen2 / Llama / Qwen3 (per-head q/k norm, models/base.py:222-225) / Qwen3-MoE (routed experts,
    models/qwen3_moe.py:86-111) shaped causal LM over a plain ``{name: tensor}`` parameter dict.
    ``moe`` = (num_experts, top_k, moe_intermediate, norm_topk_prob) or None for a dense MLP."""

    def __init__(self, params: dict, hidden, inter, layers, hq, hkv, head_dim, vocab, eps=1e-6,
                 rope_theta=10000.0, quant: str | None = None, group_size: int = 128, qk_norm: bool = False,
                 moe: tuple | None = None):
        self.p = params
        self.H, self.I, self.L, self.HQ, self.HKV, self.D, self.V = hidden, inter, layers, hq, hkv, head_dim, vocab
        self.eps, self.theta = eps, rope_theta
        self.quant, self.gs = quant, group_size
        self.qk_norm, self.moe = qk_norm, moe
        self._q = {}

    # quantised projections are derived from the fp16 masters with the oracle quantisers
    def _linear(self, x, name, bias=None):
        w = self.p[name]
        if self.quant is None:
            y = (x.float() @ w.float().T)
            if bias is not None:
                y = y + bias.float()
            return y.to(torch.float16)
        if name not in self._q:
            if self.quant == "int4":
                self._q[name] = O.quantize_int4_groupwise(w, self.gs)
            elif self.quant in ("int8", "smoothquant"):
                self._q[name] = O.quantize_int8_per_channel(w)
            elif self.quant == "fp8":
                self._q[name] = O.quantize_fp8_per_channel(w)
        q = self._q[name]
        if self.quant == "int4":
            return O.w4a16_matmul(x, q[0], q[1], q[2], group_size=self.gs, bias=bias)
        if self.quant == "smoothquant":
            return O.smoothquant_matmul(x, q[0], q[1], bias=bias)
        return O.w8a16_matmul(x, q[0], q[1], group_n=1, group_k=w.shape[1], bias=bias)

    def _moe_block(self, x2, pre):
        """fp16 router -> fp32 softmax over ALL experts -> top-k -> renormalise -> weights in the
        activation dtype (qwen3_moe.py:86-100); experts through ``fused_moe``.  int4 has no expert
        method (methods/__init__.py:53-63); smoothquant experts run as int8 W8A16 (:28-36)."""
        e, top_k, _, norm = self.moe
        logits = (x2.float() @ self.p[pre + "mlp.gate_weight"].float().T).to(x2.dtype)
        probs = torch.softmax(logits, dim=-1, dtype=torch.float32)
        w, ids = torch.topk(probs, top_k, dim=-1)
        if norm:
            w = w / w.sum(dim=-1, keepdim=True)
        w = w.to(x2.dtype)
        w1, w2 = self.p[pre + "mlp.experts.gate_up_proj"], self.p[pre + "mlp.experts.down_proj"]
        if self.quant is None:
            return O.fused_moe(x2, w1, w2, w, ids)
        if self.quant == "int4":
            raise ValueError("int4 is not available for MoE experts")
        key = pre + "experts"
        if key not in self._q:
            fn = O.quantize_fp8_per_channel if self.quant == "fp8" else O.quantize_int8_per_channel
            self._q[key] = fn(w1) + fn(w2)
        q1, s1, q2, s2 = self._q[key]
        return O.fused_moe(x2, q1, q2, w, ids, w1_scale=s1, w2_scale=s2, group_n=1, group_k=self.H)

    def _rope_tables(self, position_ids, dtype):
        inv = 1.0 / (self.theta ** (torch.arange(0, self.D, 2, dtype=torch.float32) / self.D))
        fr = position_ids.float()[:, :, None] * inv[None, None, :]
        emb = torch.cat((fr, fr), dim=-1)
        return emb.cos().to(dtype), emb.sin().to(dtype)

    def forward(self, input_ids, position_ids, info):
        """``info``: object with kv_buffer, cur_select_index, b_req_tokens_table, b_start_loc,
        b_req_idx, b_seq_len, max_actual_seq_len (same fields as the reference struct)."""
        p = self.p
        b, s = input_ids.shape
        h = p["embed_tokens.weight"][input_ids]  # [b, s, H]
        cos, sin = self._rope_tables(position_ids, h.dtype)
        residual = None
        for li in range(self.L):
            pre = f"layers.{li}."
            h, residual = O.skip_rmsnorm(h, residual, p[pre + "input_layernorm_weight"], self.eps)
            x2 = h.reshape(-1, self.H)
            xq = self._linear(x2, pre + "self_attn.q_proj.weight", p.get(pre + "self_attn.q_proj.bias"))
            xkv = self._linear(x2, pre + "self_attn.kv_proj.weight", p.get(pre + "self_attn.kv_proj.bias"))
            kvs = self.HKV * self.D
            xk, xv = xkv[:, :kvs], xkv[:, kvs:]
            n = b * s
            xq = xq.reshape(n, self.HQ, self.D).contiguous()
            xk = xk.reshape(n, self.HKV, self.D).contiguous()
            xv = xv.reshape(n, self.HKV, self.D).contiguous()
            if self.qk_norm:
                xq, _ = O.skip_rmsnorm(xq, None, p[pre + "self_attn.q_norm_weight"], self.eps)
                xk, _ = O.skip_rmsnorm(xk, None, p[pre + "self_attn.k_norm_weight"], self.eps)
            xq, xk = O.rope_emb_forward(xq, xk, cos, sin, b, s)
            O.update_kv_buffer(torch.cat([xk, xv], dim=-2), info.cur_select_index, info.kv_buffer[li])
            scale = 1.0 / math.sqrt(self.D)
            if s > 1:
                att = O.flash_attention2_no_pad(xq, xk, xv, scale * _LOG2E, info.b_start_loc, info.b_seq_len,
                                                info.max_actual_seq_len)
            else:
                kv = info.kv_buffer[li]
                att = O.flash_decoding(xq, kv[:, : self.HKV], kv[:, self.HKV :], scale, info.b_req_tokens_table,
                                       info.b_req_idx, info.b_seq_len, info.max_actual_seq_len)
            o = self._linear(att.reshape(n, self.HQ * self.D), pre + "self_attn.o_proj.weight").reshape(b, s, self.H)
            h, residual = O.skip_rmsnorm(o, residual, p[pre + "post_attention_layernorm_weight"], self.eps)
            x2 = h.reshape(-1, self.H)
            if self.moe:
                h = self._moe_block(x2, pre).reshape(b, s, self.H)
                continue
            g = self._linear(x2, pre + "mlp.gate_proj.weight")
            u = self._linear(x2, pre + "mlp.up_proj.weight")
            h = self._linear(O.swiglu_forward(g, u), pre + "mlp.down_proj.weight").reshape(b, s, self.H)
        h, _ = O.skip_rmsnorm(h, residual, p["norm_weight"], self.eps)
        return (h.float() @ p["lm_head_weight"].float().T).to(torch.float16)
