"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the lite_llama kernel hot path.

A plain torch-CPU restatement (fp32 arithmetic, explicit loops where the tiling
matters) of the algorithms behind the 16 names exported by the reference's
``lite_llama/kernels/__init__.py:23-39``.  Every function cites the reference
file:line it follows.  This module is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``lite_llama_amd``) must never route through it.

Parity pinning: ``tests/test_oracle_golden.py`` checks every function here
against fixtures under ``tests/golden/`` that were produced by running the
reference's own Triton kernels under ``TRITON_INTERPRET=1`` in the build
container (generator: ``tests/golden/gen_golden.py``).
"""

from __future__ import annotations

import math

import torch

PARTITION_SIZE = 128  # reference: kernels/flashdecoding.py:343
DECODE_BLOCK_N = 16  # reference: kernels/flashdecoding.py:179
FP8_BIT_TRICK_SCALE = 256.0  # reference: kernels/quantization/w8a16.py:39


# --------------------------------------------------------------------------- #
# a1: skip_rmsnorm  (kernels/skip_rmsnorm.py:126-189, wrapper :192-234)
# --------------------------------------------------------------------------- #
def skip_rmsnorm(X, residual, weight, eps: float = 1e-5):
    """``s = x + r`` in fp32; ``r <- s`` (rounded, in place); fp32 variance of the
    un-rounded ``s``; ``y = (s * rrms).to(dtype) * w`` with the last multiply in
    the storage dtype.  ``residual=None`` -> plain rmsnorm and ``X`` echoed back."""
    shape = X.shape
    n = shape[-1]
    x2 = X.contiguous().view(-1, n)
    s = x2.float()
    if residual is not None:
        r2 = residual.view(-1, n)  # must alias the caller's storage
        s = s + r2.float()
        r2.copy_(s.to(r2.dtype))
    var = (s * s / n).sum(dim=-1, keepdim=True)
    rrms = 1.0 / torch.sqrt(var + eps)
    y = (s * rrms).to(X.dtype) * weight
    if residual is not None:
        return y.view(shape), residual.view(shape)
    return y.view(shape), X.view(shape)


# --------------------------------------------------------------------------- #
# a2: rope_emb_forward  (kernels/rope_emb.py:14-134)
# --------------------------------------------------------------------------- #
def rope_emb_forward(q, k, cos, sin, batch_size: int, seq_len: int):
    """In-place half-split rotation; only ``cos/sin[..., :D/2]`` is read; the
    arithmetic runs in the table dtype, one rounding per elementary op."""
    n_tok, _, hd = q.shape
    assert batch_size * seq_len == n_tok
    half = hd // 2
    ct = cos.reshape(batch_size * seq_len, -1)[:, :half].unsqueeze(1)
    st = sin.reshape(batch_size * seq_len, -1)[:, :half].unsqueeze(1)
    cdt = ct.dtype
    for t in (q, k):
        x1 = t[..., :half].to(cdt)
        x2 = t[..., half : 2 * half].to(cdt)
        n1 = x1 * ct - x2 * st
        n2 = x2 * ct + x1 * st
        t[..., :half] = n1.to(t.dtype)
        t[..., half : 2 * half] = n2.to(t.dtype)
    return q, k


# --------------------------------------------------------------------------- #
# a3/a4: KV scatter and index  (kernels/update_kv_buffer.py:54-89,
#                               kernels/update_kv_index.py:50-88)
# --------------------------------------------------------------------------- #
def update_kv_buffer(KV_Values, Select_Index, KV_Buffer) -> None:
    for i in range(Select_Index.shape[0]):
        KV_Buffer[int(Select_Index[i])] = KV_Values[i]


def update_kv_index(table, b_req_idx, b_seq_len, select_index) -> None:
    for i in range(b_seq_len.shape[0]):
        table[int(b_req_idx[i]), int(b_seq_len[i]) - 1] = int(select_index[i])


# --------------------------------------------------------------------------- #
# a5: flash_decoding  (kernels/flashdecoding.py:23-161 stage 1, :224-287 stage 2)
# --------------------------------------------------------------------------- #
def slot_advance(b_seq_len, b_req_idx, cur_select_index, table):
    """Steady state of SlotBatch.begin_decode (executor/slot_batch.py:152-166), in place:
    ``b_seq_len += 1`` then ``cur_select_index = table[b_req_idx, b_seq_len - 1]``."""
    b_seq_len += 1
    cur_select_index.copy_(table[b_req_idx.long(), b_seq_len.long() - 1])


def flash_decoding(q, k_cache, v_cache, qk_scale, table, b_req_idx, b_seq_len, max_len):
    """Two-stage split-KV decode attention in fp32: 128-token partitions, 16-token
    online-softmax chunks, per-partition normalised partials + ``m + log d``, then
    an LSE merge.  Zero-length rows give NaN exactly like the reference (0/0)."""
    bsz, hq, d = q.shape
    hkv = k_cache.shape[1]
    groups = hq // hkv
    out = torch.empty_like(q)
    for b in range(bsz):  # all heads of a row advance together (vectorised over the head axis)
        length = int(b_seq_len[b])
        rows = table[int(b_req_idx[b]), :length].long()
        nparts = (length + PARTITION_SIZE - 1) // PARTITION_SIZE
        qv = q[b].float()  # [Hq, D]
        kk = k_cache[rows].float().repeat_interleave(groups, dim=1).transpose(0, 1)  # [Hq, L, D]
        vv = v_cache[rows].float().repeat_interleave(groups, dim=1).transpose(0, 1)
        part_o, part_lse = [], []
        for p in range(nparts):
            lo, hi = p * PARTITION_SIZE, min(length, (p + 1) * PARTITION_SIZE)
            m_i = torch.full((hq,), float("-inf"))
            d_i = torch.zeros(hq)
            acc = torch.zeros(hq, d)
            for c0 in range(lo, hi, DECODE_BLOCK_N):
                c1 = min(hi, c0 + DECODE_BLOCK_N)
                s = (kk[:, c0:c1] * qv[:, None, :]).sum(dim=2) * qk_scale  # [Hq, n]
                m_ij = torch.maximum(m_i, s.max(dim=1).values)
                pr = torch.exp(s - m_ij[:, None])
                alpha = torch.exp(m_i - m_ij)
                d_i = alpha * d_i + pr.sum(dim=1)
                acc = alpha[:, None] * acc + (pr[:, :, None] * vv[:, c0:c1]).sum(dim=1)
                m_i = m_ij
            part_o.append(acc / d_i[:, None])
            part_lse.append(m_i + torch.log(d_i))
        m_i = torch.full((hq,), float("-inf"))
        d_i = torch.zeros(hq)
        acc = torch.zeros(hq, d)
        for p in range(nparts):
            m_ij = torch.maximum(part_lse[p], m_i)
            alpha = torch.exp(m_i - m_ij)
            w = torch.exp(part_lse[p] - m_ij)
            acc = alpha[:, None] * acc + w[:, None] * part_o[p]
            d_i = alpha * d_i + w
            m_i = m_ij
        out[b] = (acc / d_i[:, None]).to(out.dtype)  # zero-length row: 0/0 -> NaN like the reference
    return out


# --------------------------------------------------------------------------- #
# a6: flash_attention2_no_pad  (kernels/flashattention2_nopad.py:45-231)
# --------------------------------------------------------------------------- #
def flash_attention2_no_pad(q, k, v, sm_scale, b_start_loc, b_seq_len, max_seq_len):
    """Varlen causal prefill: 64x64 tiles, ``exp2`` softmax (``sm_scale`` already
    carries log2 e), masked score ``-1.0e8``, ``P`` rounded to the V dtype before
    the PV product, fp32 accumulators.  Rows past ``b_seq_len`` are left untouched
    (the reference leaves ``torch.empty_like`` garbage there; we write zeros)."""
    blk = 64
    hq, d = q.shape[1], q.shape[2]
    hkv = k.shape[1]
    groups = hq // hkv
    out = torch.zeros_like(q)
    for b in range(b_seq_len.shape[0]):
        start, length = int(b_start_loc[b]), int(b_seq_len[b])
        for h in range(hq):
            kvh = h // groups
            qs = q[start : start + length, h].float()
            ks = k[start : start + length, kvh].float()
            vs = v[start : start + length, kvh].float()
            for m0 in range(0, length, blk):
                m1 = min(length, m0 + blk)
                rows = torch.arange(m0, m1)
                m_i = torch.full((m1 - m0,), float("-inf"))
                d_i = torch.zeros(m1 - m0)
                acc = torch.zeros(m1 - m0, d)
                for n0 in range(0, m1, blk):
                    n1 = min(m1, n0 + blk)
                    cols = torch.arange(n0, n1)
                    s = qs[m0:m1] @ ks[n0:n1].T
                    causal = rows[:, None] >= cols[None, :]
                    s = torch.where(causal, s * sm_scale, torch.full_like(s, -1.0e8))
                    m_ij = torch.maximum(m_i, s.max(dim=1).values)
                    p = torch.exp2(s - m_ij[:, None])
                    alpha = torch.exp2(m_i - m_ij)
                    d_i = d_i * alpha + p.sum(dim=1)
                    acc = acc * alpha[:, None] + p.to(v.dtype).float() @ vs[n0:n1]
                    m_i = m_ij
                out[start + m0 : start + m1, h] = (acc / d_i[:, None]).to(out.dtype)
    return out


# --------------------------------------------------------------------------- #
# a7: swiglu_forward  (kernels/swiglu.py:24-65)
# --------------------------------------------------------------------------- #
def swiglu_forward(a, b):
    af = a.float()
    return ((af * torch.sigmoid(af)) * b.float()).to(a.dtype)


# element-wise activation helpers (kernels/activations.py:19-57; fp32 arithmetic, one rounding)
def activation(x, kind: str):
    xf = x.float()
    if kind == "relu":
        y = torch.clamp_min(xf, 0)
    elif kind == "leaky_relu":  # the slope is cast to the storage dtype first (activations.py:35-37)
        slope = torch.tensor(1e-2).to(x.dtype)
        return torch.where(x >= 0, x, slope * x)
    elif kind == "tanh":
        y = 2 / (1 + torch.exp(-2 * xf)) - 1
    elif kind == "gelu":
        y = xf * 0.5 * (1.0 + torch.erf(xf / math.sqrt(2.0)))
    else:
        raise ValueError(kind)
    return y.to(x.dtype)


# --------------------------------------------------------------------------- #
# a8: w4a16_matmul  (kernels/quantization/w4a16.py:28-207)
# --------------------------------------------------------------------------- #
def unpack_int4(qweight):
    """``[N, K/8] int32`` -> ``[N, K] int32`` nibbles, LSB first along K
    (w4a16.py:9-13,99-105).  Bit-exact contract."""
    shifts = torch.arange(8, dtype=torch.int32) * 4
    return ((qweight.unsqueeze(-1) >> shifts) & 0xF).reshape(qweight.shape[0], -1)


def dequant_int4(qweight, scales, zeros, group_size):
    nib = unpack_int4(qweight).float()
    z = zeros.float().repeat_interleave(group_size, dim=1)
    s = scales.float().repeat_interleave(group_size, dim=1)
    return (nib - z) * s


def w4a16_matmul(x, qweight, scales, zeros, *, group_size: int = 128, bias=None):
    if x.dtype != torch.float16:
        raise ValueError(f"w4a16 activations must be fp16, got {x.dtype}")
    if qweight.dtype != torch.int32:
        raise ValueError(f"qweight must be int32 (packed int4), got {qweight.dtype}")
    n, kp = qweight.shape
    k = kp * 8
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    if k % group_size != 0:
        raise ValueError(f"K ({k}) must be a multiple of group_size ({group_size})")
    w = dequant_int4(qweight, scales, zeros, group_size)
    acc = x.reshape(-1, k).float() @ w.T
    if bias is not None:
        acc = acc + bias.float()
    return acc.to(x.dtype).reshape(*x.shape[:-1], n)


# --------------------------------------------------------------------------- #
# a9: w8a16_matmul  (kernels/quantization/w8a16.py:48-216)
# --------------------------------------------------------------------------- #
def fp8e4m3_bits_to_fp16(q_u8):
    """The reference's bit surgery (w8a16.py:48-62): drop sign/exp/mantissa of the
    e4m3 byte into an fp16 pattern; the result is 1/256 of the true value.
    Bit-exact contract (every finite e4m3 code is exact in fp16)."""
    b = q_u8.to(torch.int32)
    bits = ((b & 0x80) << 8) | ((b & 0x7F) << 7)
    bits = torch.where(bits >= 32768, bits - 65536, bits).to(torch.int16)
    return bits.view(torch.float16)


def _widen8(qweight):
    if qweight.dtype == torch.uint8:
        return fp8e4m3_bits_to_fp16(qweight).float(), FP8_BIT_TRICK_SCALE
    return qweight.float(), 1.0


def _blockscaled_matmul(a_f32, w_f32, scales, group_n, group_k, block_k=128):
    """``sum_tiles (A_tile @ W_tile.T) * scale[n//gn, k0//gk]`` with 128-wide k tiles
    (w8a16.py:108-121; fused_moe.py:179-201)."""
    n, k = w_f32.shape
    acc = torch.zeros(a_f32.shape[0], n, dtype=torch.float32)
    rows = torch.arange(n) // group_n
    for k0 in range(0, k, block_k):
        k1 = min(k, k0 + block_k)
        s = scales[rows, k0 // group_k].float()
        acc += (a_f32[:, k0:k1] @ w_f32[:, k0:k1].T) * s[None, :]
    return acc


def w8a16_matmul(x, qweight, scales, *, group_n: int, group_k: int, bias=None):
    is_fp8 = qweight.dtype == torch.uint8
    if not is_fp8 and qweight.dtype != torch.int8:
        raise ValueError(f"qweight must be uint8 (fp8) or int8, got {qweight.dtype}")
    if x.dtype != torch.float16:
        raise ValueError(f"w8a16 activations must be fp16, got {x.dtype}")
    n, k = qweight.shape
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    if group_k % 128 != 0 and group_k < k:
        raise ValueError(f"group_k ({group_k}) must be a multiple of 128 unless it covers K")
    w, dq = _widen8(qweight)
    acc = _blockscaled_matmul(x.reshape(-1, k).float(), w, scales, group_n, min(group_k, k))
    acc = acc * dq
    if bias is not None:
        acc = acc + bias.float()
    return acc.to(x.dtype).reshape(*x.shape[:-1], n)


# --------------------------------------------------------------------------- #
# a10: smoothquant_matmul  (kernels/quantization/w8a8.py:34-217)
# --------------------------------------------------------------------------- #
def quantize_activations_int8(a):
    """Per-row ``scale = absmax/127`` (1.0 when 0) and ``q = trunc(x / scale)``
    (w8a8.py:49-66; the float->int8 cast truncates toward zero).  Bit-exact."""
    af = a.float()
    scale = af.abs().amax(dim=1) / 127.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    q = torch.trunc(af / scale[:, None]).to(torch.int8)
    return q, scale


def smoothquant_matmul(x, qweight, weight_scales, *, bias=None):
    if x.dtype != torch.float16:
        raise ValueError(f"smoothquant activations must be fp16, got {x.dtype}")
    if qweight.dtype != torch.int8:
        raise ValueError(f"qweight must be int8, got {qweight.dtype}")
    n, k = qweight.shape
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    qa, a_scale = quantize_activations_int8(x.reshape(-1, k))
    # int32 accumulate is exact; do it in int64 on CPU then narrow (values fit).
    acc = (qa.to(torch.int64) @ qweight.to(torch.int64).T).to(torch.int32)
    ws = weight_scales.reshape(-1).float()
    res = acc.float() * a_scale[:, None] * ws[None, :]
    if bias is not None:
        res = res + bias.float()
    return res.to(x.dtype).reshape(*x.shape[:-1], n)


def smoothquant_int32_acc(x, qweight):
    """The exact int32 accumulators (for the bit-exact part of a10)."""
    qa, a_scale = quantize_activations_int8(x.reshape(-1, x.shape[-1]))
    return (qa.to(torch.int64) @ qweight.to(torch.int64).T).to(torch.int32), qa, a_scale


# --------------------------------------------------------------------------- #
# a11: moe_align_block_size + fused_moe  (kernels/fused_moe.py:45-99, :105-438)
# --------------------------------------------------------------------------- #
def moe_align_block_size(topk_ids, block_size: int, num_experts: int):
    """Stable sort of slot ids by expert, every expert's run padded to
    ``block_size`` with the sentinel ``num_slots``; static output shapes."""
    flat = topk_ids.reshape(-1).to(torch.int64).tolist()
    num_slots = len(flat)
    max_padded = num_slots + num_experts * (block_size - 1)
    max_blocks = (max_padded + block_size - 1) // block_size
    sorted_ids = [num_slots] * max_padded
    per_expert = [[] for _ in range(num_experts)]
    for slot, e in enumerate(flat):
        per_expert[e].append(slot)
    pos = 0
    block_ends = []
    nblk = 0
    for e in range(num_experts):
        run = per_expert[e]
        for j, slot in enumerate(run):
            sorted_ids[pos + j] = slot
        padded = (len(run) + block_size - 1) // block_size * block_size
        pos += padded
        nblk += padded // block_size
        block_ends.append(nblk)
    expert_ids = []
    for blk in range(max_blocks):
        e = 0
        while e < num_experts and block_ends[e] <= blk:  # searchsorted(right=True)
            e += 1
        expert_ids.append(min(e, num_experts - 1))
    return (
        torch.tensor(sorted_ids, dtype=torch.int32),
        torch.tensor(expert_ids, dtype=torch.int32),
        torch.tensor([pos], dtype=torch.int32),
    )


def moe_block_m(num_tokens: int) -> int:
    """fused_moe.py:214-222."""
    return 16 if num_tokens <= 16 else (32 if num_tokens <= 64 else 64)


def _expert_gemm(a_f32, w, scale, group_n, group_k):
    if scale is None:
        return a_f32 @ w.float().T
    wf, dq = _widen8(w)
    k = w.shape[1]
    return _blockscaled_matmul(a_f32, wf, scale, group_n or 1, min(group_k, k) if group_k else 1) * dq


def fused_moe(hidden_states, w1, w2, topk_weights, topk_ids, *, w1_scale=None, w2_scale=None,
              group_n: int = 0, group_k: int = 0):
    """GEMM1 -> fp16 ``[T*k, 2I]``; silu*up -> fp16; GEMM2 (x router weight in fp32)
    -> fp16 ``[T*k, H]``; fp32 sum over top_k -> fp16 ``[T, H]``."""
    t, hidden = hidden_states.shape
    e, two_i, _ = w1.shape
    inter = two_i // 2
    top_k = topk_ids.shape[1]
    dtype = hidden_states.dtype
    quant = w1_scale is not None
    if quant:
        for w in (w1, w2):
            if w.dtype not in (torch.uint8, torch.int8):
                raise ValueError(f"quantised expert weights must be uint8 or int8, got {w.dtype}")
    if (w1_scale is None) != (w2_scale is None):
        raise ValueError("w1 and w2 must use the same quantisation format")
    if quant and group_k % 128 != 0 and group_k < min(hidden, inter):
        raise ValueError(f"group_k ({group_k}) must be a multiple of 128 unless it covers K")
    flat_ids = topk_ids.reshape(-1).to(torch.int64)
    flat_w = topk_weights.reshape(-1).to(dtype).float()
    x = hidden_states.float()
    expanded = torch.zeros(t * top_k, hidden, dtype=dtype)
    for slot in range(t * top_k):
        ex = int(flat_ids[slot])
        xa = x[slot // top_k : slot // top_k + 1]
        gu = _expert_gemm(xa, w1[ex], None if not quant else w1_scale[ex], group_n, group_k).to(dtype)
        g, u = gu[:, :inter].float(), gu[:, inter:].float()
        act = (g * torch.sigmoid(g) * u).to(dtype)
        y = _expert_gemm(act.float(), w2[ex], None if not quant else w2_scale[ex], group_n, group_k)
        expanded[slot] = (y * flat_w[slot]).to(dtype)[0]
    out = expanded.view(t, top_k, hidden).float().sum(dim=1)
    return out.to(dtype)


# --------------------------------------------------------------------------- #
# a16: greedy sampling  (engine/sampler.py:227-228,264)
# --------------------------------------------------------------------------- #
def greedy_argmax(logits):
    return torch.argmax(logits, dim=-1)


# --------------------------------------------------------------------------- #
# sampler row: engine/sampler.py:77-137
# --------------------------------------------------------------------------- #
def apply_repetition_penalty(logits, token_ids, mask, penalty):
    """sampler.py:77-115: logits of tokens that occur at a ``mask``-ed position of ``token_ids`` are
    divided by ``penalty`` when >= 0 and multiplied when negative; repeats are penalised once; padded
    positions never count.  Arithmetic in fp32 with one rounding to the result dtype, which is the
    logits dtype for a Python-scalar penalty and the promoted dtype (float32 for the reference's
    float32 ``[batch, 1]`` factors) for a tensor penalty -- torch's type promotion."""
    batch, vocab = logits.shape
    if torch.is_tensor(penalty):
        out_dtype = torch.promote_types(logits.dtype, penalty.dtype)
        pen = penalty.reshape(batch, -1)[:, :1].float()
    else:
        out_dtype = logits.dtype
        pen = torch.full((batch, 1), float(penalty), dtype=torch.float32)
    out = logits.to(out_dtype).clone()
    for b in range(batch):
        for tok in set(token_ids[b][mask[b]].tolist()):
            x = logits[b, tok].float()
            out[b, tok] = (x * pen[b, 0] if x < 0 else x / pen[b, 0]).to(out_dtype)
    return out


def top_p_distribution(logits, temperature, top_p):
    """sampler.py:118-137 up to the random draw: softmax(logits / T), sort descending, drop a token
    once the mass BEFORE it exceeds top_p, renormalise.  Returned in token order ``[batch, vocab]``."""
    batch, vocab = logits.shape
    t = torch.as_tensor(temperature, dtype=torch.float32).reshape(-1, 1)
    tp = torch.as_tensor(top_p, dtype=torch.float32).reshape(-1, 1)
    probs = torch.softmax(logits.float() / t, dim=-1)
    sp, si = torch.sort(probs, dim=-1, descending=True, stable=True)  # ties: ascending token index
    cum = torch.cumsum(sp, dim=-1)
    sp = torch.where(cum - sp > tp, torch.zeros_like(sp), sp)
    sp = sp / sp.sum(dim=-1, keepdim=True)
    return torch.zeros_like(probs).scatter_(1, si, sp)


def sample_from_distribution(dist, uniform):
    """Inverse CDF in TOKEN order: the first index whose cumulative mass exceeds ``u`` (the reference
    draws with torch.multinomial on the sorted distribution -- same distribution, other RNG stream)."""
    cdf = torch.cumsum(dist.double(), dim=-1)
    u = torch.as_tensor(uniform, dtype=torch.float64).reshape(-1, 1) * cdf[:, -1:]
    return (cdf > u).float().argmax(dim=-1)


# --------------------------------------------------------------------------- #
# Weight quantisers that define the on-device formats
# (models/quantization/params/int4.py:12-49, int8.py:12-53, fp8.py:16-30)
# --------------------------------------------------------------------------- #
def quantize_int4_groupwise(weight, group_size: int = 128):
    n, k = weight.shape
    if k % group_size != 0:
        raise ValueError(f"in_features {k} must be a multiple of group_size {group_size}")
    w = weight.float().reshape(n, k // group_size, group_size)
    lo, hi = w.amin(dim=-1), w.amax(dim=-1)
    scale = (hi - lo).clamp(min=1e-5) / 14.0
    zero = (-lo / scale).round().clamp(0, 15)
    q = (w / scale[..., None] + zero[..., None]).round().clamp(0, 15).to(torch.int64)
    q = q.reshape(n, k // 8, 8)
    packed = torch.zeros(n, k // 8, dtype=torch.int64)
    for j in range(8):
        packed |= q[:, :, j] << (4 * j)
    packed = torch.where(packed >= 2**31, packed - 2**32, packed)  # wrap to int32
    return packed.to(torch.int32), scale.float(), zero.float()


def quantize_int8_per_channel(weight):
    scale = weight.abs().amax(dim=-1, keepdim=True).float() / 127.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    return (weight.float() / scale).round().clamp_(-127, 127).to(torch.int8), scale


def quantize_int8_groupwise(weight, group_size: int = 128):
    k = weight.shape[-1]
    if k % group_size != 0:
        raise ValueError(f"in_features {k} must be a multiple of group_size {group_size}")
    w = weight.float().unflatten(-1, (k // group_size, group_size))
    scale = w.abs().amax(dim=-1) / 127.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    q = (w / scale.unsqueeze(-1)).round().clamp_(-127, 127).to(torch.int8)
    return q.flatten(-2), scale


def quantize_fp8_per_channel(weight):
    scale = weight.abs().amax(dim=-1, keepdim=True).float() / 448.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    q = (weight.float() / scale).clamp_(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale
