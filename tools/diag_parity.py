import sys, json, torch, types
sys.path.insert(0, "/root/repo")
import bench
from lite_llama_amd.model import GEOMETRY
model, quant, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
geo = GEOMETRY[model]
port, sample = bench.cpu_baseline(geo, batch, 512, 1, quant)
# replicate parity_check but keep tensors
from lite_llama_amd.model import CausalLM, tiny_geometry
from lite_llama_amd.quantization import QuantConfig
p, layers = sample["params"], sample["layers"]
g1 = tiny_geometry(name="x", hidden_size=geo.hidden_size, intermediate_size=geo.intermediate_size, num_layers=layers, num_heads=geo.num_heads,
                   num_kv_heads=geo.num_kv_heads, head_dim=geo.head_dim, vocab_size=geo.vocab_size, rope_theta=geo.rope_theta,
                   rms_norm_eps=geo.rms_norm_eps, qkv_bias=geo.qkv_bias, use_qk_norm=geo.use_qk_norm, num_experts=geo.num_experts,
                   num_experts_per_tok=geo.num_experts_per_tok, moe_intermediate_size=geo.moe_intermediate_size, norm_topk_prob=geo.norm_topk_prob)
m = CausalLM(g1); m.load_state_dict(p, strict=True); m = m.to("cuda")
if quant != "none": m.quantize_(QuantConfig.for_runtime_scheme(quant))
info_c = sample["info"]; ctx = int(sample["pos"][0, 0]); m.rotary_emb.ensure(ctx + 8, "cuda")
kv = [k.clone().cuda() for k in sample["kv_before"]]
info = types.SimpleNamespace(kv_buffer=kv, cur_select_index=info_c.cur_select_index.cuda(), b_req_tokens_table=info_c.b_req_tokens_table.cuda(), b_start_loc=None,
                             b_req_idx=info_c.b_req_idx.cuda(), b_seq_len=info_c.b_seq_len.cuda(), max_actual_seq_len=info_c.max_actual_seq_len)
with torch.no_grad():
    got = m(sample["ids"].cuda(), sample["pos"].cuda(), info).float().cpu()
ref = sample["logits"].float()
err = (got - ref).abs()[:, -1]
print("logits std", float(ref.std()), "max|ref|", float(ref.abs().max()))
rowmax = err.max(-1).values
print("row max err sorted:", [round(float(v), 3) for v in rowmax.sort().values])
print("row rms err / ref rms:", [round(float(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()), 4) for e, r in zip((got - ref)[:, -1], ref[:, -1])][:16])
