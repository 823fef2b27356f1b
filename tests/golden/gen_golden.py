"""Generate golden vectors from the REFERENCE's own Triton kernels (build container only).

Run here (no GPU) with the reference mounted read-only at /root/reference:

    TRITON_INTERPRET=1 python tests/golden/gen_golden.py

Every exported reference kernel (lite_llama/kernels/__init__.py:23-39) is executed
on seeded CPU inputs under Triton's interpreter and the inputs + outputs are saved
as plain arrays in ``tests/golden/*.npz``.  Only data is stored -- no reference
source, bytecode or pickled objects.  fp16 tensors are stored as numpy float16,
bf16 tensors as their uint16 bit patterns (key suffix ``__bf16``).
"""

from __future__ import annotations

import math
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _np(t: torch.Tensor):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def save(name: str, **tensors):
    out = {}
    for key, val in tensors.items():
        if isinstance(val, torch.Tensor):
            if val.dtype == torch.bfloat16:
                out[key + "__bf16"] = _np(val.contiguous())
            else:
                out[key] = _np(val.contiguous())
        else:
            out[key] = np.asarray(val)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def main():
    from lite_llama.kernels import (
        flash_attention2_no_pad,
        flash_decoding,
        fused_moe,
        moe_align_block_size,
        rope_emb_forward,
        skip_rmsnorm,
        swiglu_forward,
        update_kv_buffer,
        update_kv_index,
        w4a16_matmul,
        w8a16_matmul,
        smoothquant_matmul,
    )
    from lite_llama.kernels.quantization.w8a8 import _quantize_activations_kernel
    from lite_llama.models.quantization.params.int4 import quantize_int4_groupwise
    from lite_llama.models.quantization.params.int8 import (
        quantize_int8_groupwise,
        quantize_int8_per_channel,
    )
    from lite_llama.models.quantization.params.fp8 import quantize_fp8_per_channel

    g = torch.Generator().manual_seed(1234)

    def randn(*shape, dtype=torch.float16, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dtype)

    # ---- skip_rmsnorm ------------------------------------------------------ #
    for tag, shape, dtype, with_res, eps in [
        ("f16_res", (4, 1, 896), torch.float16, True, 1e-6),
        ("f16_nores", (2, 3, 512), torch.float16, False, 1e-5),
        ("bf16_res", (3, 256), torch.bfloat16, True, 1e-6),
        ("f16_head128", (5, 4, 128), torch.float16, False, 1e-6),
    ]:
        x = randn(*shape, dtype=dtype)
        r = randn(*shape, dtype=dtype) if with_res else None
        w = (1 + 0.1 * torch.randn(shape[-1], generator=g)).to(dtype)
        r_in = r.clone() if with_res else None
        y, r_out = skip_rmsnorm(x.clone(), r, w, eps)
        # The interpreter's bf16 multiply (``(x*rrms).to(bf16) * w``) returns garbage
        # (numpy has no bf16); only the fp32->bf16 residual store is trustworthy there.
        y_valid = int(dtype != torch.bfloat16)
        save(f"skip_rmsnorm_{tag}", x=x, w=w, eps=eps, has_res=int(with_res), y_valid=y_valid,
             **({"r_in": r_in, "r_out": r_out} if with_res else {}), **({"y": y} if y_valid else {}))

    # ---- swiglu ------------------------------------------------------------ #
    a, b = randn(3, 300), randn(3, 300)
    save("swiglu_f16", a=a, b=b, c=swiglu_forward(a, b))
    a, b = randn(2, 1, 4864), randn(2, 1, 4864)
    save("swiglu_f16_4864", a=a, b=b, c=swiglu_forward(a, b))

    # ---- rope -------------------------------------------------------------- #
    for tag, bs, sl, hq, hk, hd in [("d64", 2, 3, 4, 2, 64), ("d128_decode", 5, 1, 28, 4, 128)]:
        q, k = randn(bs * sl, hq, hd), randn(bs * sl, hk, hd)
        pos = torch.randint(0, 2000, (bs, sl), generator=g).float()
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
        fr = pos[..., None] * inv
        emb = torch.cat([fr, fr], dim=-1)
        cos, sin = emb.cos().half(), emb.sin().half()
        q_out, k_out = rope_emb_forward(q.clone(), k.clone(), cos, sin, bs, sl)
        save(f"rope_{tag}", q=q, k=k, cos=cos, sin=sin, bs=bs, sl=sl, q_out=q_out, k_out=k_out)

    # ---- kv cache ops ------------------------------------------------------ #
    vals = randn(6, 4, 64)
    buf = randn(32, 4, 64)
    idx = torch.tensor([5, 0, 31, 7, 8, 20], dtype=torch.int32)
    buf_out = buf.clone()
    update_kv_buffer(vals, idx, buf_out)
    save("update_kv_buffer", vals=vals, idx=idx, buf_in=buf, buf_out=buf_out)

    table = torch.zeros(4, 16, dtype=torch.int32)
    req = torch.tensor([2, 0, 3], dtype=torch.int64)
    sl_ = torch.tensor([5, 16, 1], dtype=torch.int64)
    sel = torch.tensor([101, 202, 303], dtype=torch.int32)
    t_out = table.clone()
    update_kv_index(t_out, req, sl_, sel)
    save("update_kv_index", table_in=table, req=req, seq=sl_, sel=sel, table_out=t_out)

    # ---- flash_decoding ---------------------------------------------------- #
    def decode_case(tag, lens, hq, hkv, d, dtype, req_idx=None, scattered=True):
        max_tok = 512
        kc = randn(max_tok, hkv, d, dtype=dtype)
        vc = randn(max_tok, hkv, d, dtype=dtype)
        q = randn(len(lens), hq, d, dtype=dtype, scale=0.3)
        width = max(lens)
        nreq = len(lens)
        table = torch.zeros(nreq, width, dtype=torch.int32)
        perm = torch.randperm(max_tok, generator=g).to(torch.int32)
        off = 0
        for i, n in enumerate(lens):
            table[i, :n] = perm[off : off + n] if scattered else torch.arange(off, off + n)
            off += n
        ridx = torch.tensor(req_idx if req_idx is not None else list(range(nreq)), dtype=torch.int32)
        seq = torch.tensor([lens[i] for i in ridx.tolist()], dtype=torch.int32)
        out = flash_decoding(q, kc, vc, 1.0 / math.sqrt(d), table, ridx, seq, int(seq.max()))
        save(f"flash_decoding_{tag}", q=q, k_cache=kc, v_cache=vc, scale=1.0 / math.sqrt(d),
             table=table, req_idx=ridx, seq_len=seq, max_len=int(seq.max()), out=out)

    decode_case("ragged_d64", [17, 129, 64, 3], 4, 2, 64, torch.float16, req_idx=[2, 1, 0, 3])
    decode_case("gqa7_d128", [48, 130], 14, 2, 128, torch.float16)
    decode_case("bf16_d64", [1, 200], 8, 1, 64, torch.bfloat16)
    decode_case("d32_shared_slot", [33, 16], 4, 4, 32, torch.float16, req_idx=[0, 1, 1])

    # ---- flash_attention2_no_pad ------------------------------------------- #
    for tag, lens, hq, hkv, d in [("d64", [70, 33], 4, 2, 64), ("d128", [65, 5, 64], 2, 1, 128)]:
        lp = max(lens)
        bsz = len(lens)
        q, k, v = randn(bsz * lp, hq, d, scale=0.5), randn(bsz * lp, hkv, d, scale=0.5), randn(bsz * lp, hkv, d)
        start = torch.arange(bsz, dtype=torch.int32) * lp
        seq = torch.tensor(lens, dtype=torch.int32)
        scale = 1.4426950408889634 / math.sqrt(d)
        out = flash_attention2_no_pad(q, k, v, scale, start, seq, lp)
        valid = torch.zeros(bsz * lp, dtype=torch.bool)
        for i, n in enumerate(lens):
            valid[i * lp : i * lp + n] = True
        out = torch.where(valid[:, None, None], out, torch.zeros_like(out))
        save(f"fa2_nopad_{tag}", q=q, k=k, v=v, sm_scale=scale, b_start_loc=start, b_seq_len=seq,
             max_seq_len=lp, valid=valid, out=out)

    # ---- w4a16 --------------------------------------------------------------- #
    for tag, m, n, k, gs, has_bias in [("g128_bias", 8, 256, 512, 128, True), ("g32", 3, 130, 256, 32, False),
                                       ("m40_g128", 40, 192, 384, 128, False)]:
        x = randn(m, k, scale=0.5)
        w = torch.randn(n, k, generator=g) * 0.05
        qw, sc, zr = quantize_int4_groupwise(w, gs)
        bias = randn(n, scale=0.1) if has_bias else None
        y = w4a16_matmul(x, qw, sc, zr, group_size=gs, bias=bias)
        save(f"w4a16_{tag}", x=x, w_fp32=w, qweight=qw, scales=sc, zeros=zr, group_size=gs,
             **({"bias": bias} if has_bias else {}), y=y)

    # ---- w8a16 --------------------------------------------------------------- #
    m, n, k = 8, 256, 384
    x = randn(m, k, scale=0.5)
    w = torch.randn(n, k, generator=g) * 0.05
    qw = w.to(torch.float8_e4m3fn).view(torch.uint8)
    sc = torch.rand(2, 3, generator=g) + 0.5
    y = w8a16_matmul(x, qw, sc, group_n=128, group_k=128)
    save("w8a16_fp8_block", x=x, qweight=qw, scales=sc, group_n=128, group_k=128, y=y)

    m, n, k = 5, 130, 384
    x = randn(m, k, scale=0.5)
    w = torch.randn(n, k, generator=g) * 0.05
    qw, sc = quantize_int8_per_channel(w)
    bias = randn(n, scale=0.1)
    y = w8a16_matmul(x, qw, sc, group_n=1, group_k=k, bias=bias)
    save("w8a16_int8_chan", x=x, w_fp32=w, qweight=qw, scales=sc, group_n=1, group_k=k, bias=bias, y=y)

    m, n, k = 4, 64, 256
    x = randn(m, k, scale=0.5)
    w = torch.randn(n, k, generator=g) * 0.05
    qw, sc = quantize_int8_groupwise(w, 128)
    y = w8a16_matmul(x, qw, sc, group_n=1, group_k=128)
    save("w8a16_int8_group", x=x, w_fp32=w, qweight=qw, scales=sc, group_n=1, group_k=128, y=y)

    w = torch.randn(16, 64, generator=g) * 0.05
    qw, sc = quantize_fp8_per_channel(w)
    save("quantize_fp8_per_channel", w_fp32=w, qweight=qw, scales=sc)

    # ---- smoothquant --------------------------------------------------------- #
    m, n, k = 8, 256, 512
    x = randn(m, k, scale=0.5)
    x[3] = 0  # all-zero row -> scale 1.0 branch
    w = torch.randn(n, k, generator=g) * 0.05
    qw, sc = quantize_int8_per_channel(w)
    bias = randn(n, scale=0.1)
    y = smoothquant_matmul(x, qw, sc, bias=bias)
    qa = torch.empty(m, k, dtype=torch.int8)
    a_scale = torch.empty(m, dtype=torch.float32)
    _quantize_activations_kernel[(m,)](x, qa, a_scale, m, k, x.stride(0), x.stride(1),
                                       qa.stride(0), qa.stride(1), BLOCK_K=512, num_warps=4)
    save("smoothquant", x=x, qweight=qw, scales=sc, bias=bias, qa=qa, a_scale=a_scale, y=y)

    # ---- moe_align_block_size ------------------------------------------------ #
    for tag, t, topk, e, bm in [("t37_e8", 37, 2, 8, 32), ("t3_e128", 3, 8, 128, 16), ("t70_e4", 70, 2, 4, 64)]:
        ids = torch.randint(0, e, (t, topk), generator=g, dtype=torch.int32)
        s, ex, npost = moe_align_block_size(ids, bm, e)
        save(f"moe_align_{tag}", topk_ids=ids, block_size=bm, num_experts=e,
             sorted_ids=s, expert_ids=ex, num_post=npost)

    # ---- fused_moe ------------------------------------------------------------ #
    def moe_case(tag, t, e, topk, h, i, mode):
        x = randn(t, h, scale=1.0 / math.sqrt(h) * 4)
        w1 = torch.randn(e, 2 * i, h, generator=g) / math.sqrt(h)
        w2 = torch.randn(e, h, i, generator=g) / math.sqrt(i)
        ids = torch.randint(0, e, (t, topk), generator=g, dtype=torch.int64)
        wts = torch.softmax(torch.randn(t, topk, generator=g), dim=-1)
        kw = {}
        if mode == "f16":
            w1q, w2q = w1.half(), w2.half()
        elif mode == "fp8_block":
            w1q = w1.to(torch.float8_e4m3fn).view(torch.uint8)
            w2q = w2.to(torch.float8_e4m3fn).view(torch.uint8)
            gn = gk = 128
            kw = dict(w1_scale=torch.rand(e, (2 * i + gn - 1) // gn, (h + gk - 1) // gk, generator=g) + 0.5,
                      w2_scale=torch.rand(e, (h + gn - 1) // gn, (i + gk - 1) // gk, generator=g) + 0.5,
                      group_n=gn, group_k=gk)
        else:  # int8 per channel, as W8A16MoeMethod passes it
            w1q, s1 = quantize_int8_per_channel(w1)
            w2q, s2 = quantize_int8_per_channel(w2)
            kw = dict(w1_scale=s1, w2_scale=s2, group_n=1, group_k=min(1 << 30, h))
        out = fused_moe(x, w1q, w2q, wts, ids, **kw)
        save(f"fused_moe_{tag}", x=x, w1=w1q, w2=w2q, topk_weights=wts, topk_ids=ids, out=out,
             **{k_: v_ for k_, v_ in kw.items()})

    moe_case("f16", 5, 4, 2, 128, 64, "f16")
    moe_case("f16_t37", 37, 8, 2, 128, 64, "f16")
    moe_case("fp8_block", 3, 4, 2, 256, 128, "fp8_block")
    moe_case("int8_chan", 4, 4, 2, 128, 128, "int8")

    # ---- quantisers ------------------------------------------------------------ #
    w = torch.randn(6, 256, generator=g) * 0.05
    qw, sc, zr = quantize_int4_groupwise(w.half(), 128)
    save("quantize_int4_groupwise", w=w.half(), qweight=qw, scales=sc, zeros=zr, group_size=128)
    qw, sc = quantize_int8_per_channel(w.half())
    save("quantize_int8_per_channel", w=w.half(), qweight=qw, scales=sc)


if __name__ == "__main__":
    main()
