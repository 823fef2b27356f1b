"""fused_moe / moe_align_block_size -- mirror of lite_llama/kernels/fused_moe.py:45-99,352-438
over the HIP C-ABI.  Same vLLM data protocol: ``sorted_token_ids`` (slot ids sorted by expert,
each expert's run padded to BLOCK_M with the sentinel ``num_slots``), ``expert_ids`` (expert
per row block) and ``num_tokens_post_padded`` (device scalar; no host sync anywhere)."""

from __future__ import annotations

import torch

from .. import _lib as L


def moe_align_block_size(
    topk_ids: torch.Tensor, block_size: int, num_experts: int
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns ``(sorted_token_ids i32[max_padded], expert_ids i32[max_blocks],
    num_tokens_post_padded i32[1])`` -- all shapes static (graph-capturable)."""
    L.require_cuda(topk_ids)
    device = topk_ids.device
    flat = topk_ids.reshape(-1)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    num_slots = flat.numel()
    max_padded = num_slots + num_experts * (block_size - 1)
    max_blocks = (max_padded + block_size - 1) // block_size
    sorted_ids = torch.empty((max_padded,), dtype=torch.int32, device=device)
    expert_ids = torch.empty((max_blocks,), dtype=torch.int32, device=device)
    num_post = torch.empty((1,), dtype=torch.int32, device=device)
    # prefill-sized inputs: per-chunk counts -> prefix -> stable placement over a scratch table (same outputs; the one-workgroup
    # kernel walks all slots once per expert: 21.9 ms at 32 768 tokens x top-8, round 6)
    ws_ints = L.lib().ll_moe_align_workspace_ints(num_slots, int(num_experts))
    ws = torch.empty((ws_ints,), dtype=torch.int32, device=device) if ws_ints > 0 else None
    L.check(
        L.lib().ll_moe_align_block_size_ws(
            flat.data_ptr(), L.index_width(flat), num_slots, int(num_experts), int(block_size),
            sorted_ids.data_ptr(), expert_ids.data_ptr(), num_post.data_ptr(), L.ptr(ws), ws_ints, L.stream_ptr(),
        ),
        "moe_align_block_size",
    )
    return sorted_ids, expert_ids, num_post


def _block_m(num_tokens: int) -> int:
    # reference _launch_config (fused_moe.py:214-222): both GEMMs share one alignment
    if num_tokens <= 16:
        return 16
    if num_tokens <= 64:
        return 32
    return 64


def fused_moe_block_m(num_tokens: int) -> int:
    """The row-block size :func:`fused_moe` aligns ``num_tokens`` tokens to."""
    return _block_m(num_tokens)


def _wfmt(weight: torch.Tensor, scale) -> int:
    if scale is None:
        return L.LL_W_F16
    if weight.dtype == torch.uint8:
        return L.LL_W_FP8E4M3
    if weight.dtype == torch.int8:
        return L.LL_W_INT8
    raise ValueError(f"quantised expert weights must be uint8 or int8, got {weight.dtype}")


def _moe_gemm(a, w, c, w_scale, topk_w, sorted_ids, expert_ids, num_post, top_k, mul_w, wfmt,
              group_n, group_k, block_m):
    assert a.stride(-1) == 1 and w.stride(-1) == 1, "last dims must be contiguous"
    n, k = w.shape[1], w.shape[2]
    if w_scale is not None:
        ss = w_scale.stride()
        gk = min(group_k, k) if group_k else 1
        gn = group_n or 1
    else:
        ss, gk, gn = (0, 0, 0), 1, 1
    L.check(
        L.lib().ll_moe_gemm(
            c.data_ptr(), a.data_ptr(), w.data_ptr(), L.ptr(w_scale), topk_w.data_ptr(),
            sorted_ids.data_ptr(), expert_ids.data_ptr(), num_post.data_ptr(), c.shape[0],
            sorted_ids.numel(), block_m, n, k, top_k, int(mul_w), wfmt, gn, gk, a.stride(0),
            w.stride(0), w.stride(1), ss[0], ss[1], ss[2], L.dtype_code(c.dtype), L.stream_ptr(),
        ),
        "fused_moe grouped GEMM",
    )


def fused_moe(
    hidden_states: torch.Tensor,
    w1: torch.Tensor,
    w2: torch.Tensor,
    topk_weights: torch.Tensor,
    topk_ids: torch.Tensor,
    *,
    w1_scale: torch.Tensor | None = None,
    w2_scale: torch.Tensor | None = None,
    group_n: int = 0,
    group_k: int = 0,
    w1_group: tuple | None = None,
    w2_group: tuple | None = None,
    slots_ok: bool = False,
    w1_interleaved: bool = False,
    aligned: tuple | None = None,
) -> torch.Tensor:
    """``sum_k w_k * (silu(x W1g^T) * (x W1u^T)) W2^T`` over each token's top-k experts.

    ``w1_group`` / ``w2_group`` (extension, ``(group_n, group_k)`` per matrix): a tensor-parallel shard whose cut does
    not end on the checkpoint's scale blocks carries its scales re-expressed on a finer grid -- gate|up along N, down
    along K (model.py::SparseMoeBlock); multiples of 8 along K.

    ``slots_ok`` (extension): return the per-slot rows as a ``SlotSums`` for ``skip_rmsnorm_partials`` instead of running
    ``moe_sum`` (the same values, one launch less).

    ``aligned`` (extension): ``(sorted_token_ids, expert_ids, num_tokens_post_padded, block_m)`` of ``topk_ids`` when the router
    already produced them (:func:`moe_router` with ``align_block=fused_moe_block_m(num_tokens)``); ignored if ``block_m`` differs.

    ``w1_interleaved`` (extension): the rows of ``w1`` (and of its scales) were interleaved at load time -- row ``2j`` =
    ``gate_j``, row ``2j + 1`` = ``up_j`` instead of the stacked halves -- so that ``silu(gate) * up`` runs in the first grouped
    GEMM's epilogue (same values as GEMM + ``silu_and_mul``: both outputs are rounded to the activation dtype first).

    Pipeline and intermediate dtypes as the reference: align -> GEMM1 (``[T*k, 2I]`` in x's
    dtype) -> silu*up -> GEMM2 with the router weight folded in fp32 -> fp32 sum over top_k."""
    num_tokens, hidden = hidden_states.shape
    num_experts, two_inter, _ = w1.shape
    intermediate = two_inter // 2
    top_k = topk_ids.shape[1]
    device = hidden_states.device
    dtype = hidden_states.dtype

    wfmt = _wfmt(w1, w1_scale)
    if wfmt != _wfmt(w2, w2_scale):
        raise ValueError("w1 and w2 must use the same quantisation format")
    if wfmt and group_k % 128 != 0 and group_k < min(hidden, intermediate) and w1_group is None and w2_group is None:
        raise ValueError(f"group_k ({group_k}) must be a multiple of 128 unless it covers K")
    g1 = w1_group if w1_group is not None else (group_n, group_k)
    g2 = w2_group if w2_group is not None else (group_n, group_k)
    if wfmt and any(gk < kdim and gk % 8 for gk, kdim in ((g1[1], hidden), (g2[1], intermediate))):
        raise ValueError("scale blocks along K must be multiples of 8")
    L.require_cuda(hidden_states, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale)
    if hidden_states.stride(-1) != 1:
        hidden_states = hidden_states.contiguous()

    if topk_ids.dtype not in (torch.int32, torch.int64):  # (the align kernel reads either width: no cast launch)
        topk_ids = topk_ids.to(torch.int32)
    flat_weights = topk_weights.reshape(-1).to(dtype).contiguous()

    block_m = _block_m(num_tokens)
    if (num_tokens * top_k >= 256 * num_experts and hidden % 128 == 0 and intermediate % 128 == 0
            and (not wfmt or all(gk >= kdim or gk % 64 == 0 for gk, kdim in ((g1[1], hidden), (g2[1], intermediate))))
            and w1.stride(1) % 16 == 0 and w2.stride(1) % 16 == 0 and w1.stride(0) % 16 == 0 and w2.stride(0) % 16 == 0):
        # (extension, round 6) prefill-sized inputs -- 256+ rows per expert on average: 128-row blocks on the full-line grouped GEMM
        # (a weight tile is dequantised once for four MFMA row tiles and re-read from L2 half as often); same values per slot
        block_m = 128
    if aligned is not None and aligned[3] == block_m:  # (extension) moe_align_block_size already ran inside the router's launch
        sorted_ids, expert_ids, num_post = aligned[:3]
    else:
        sorted_ids, expert_ids, num_post = moe_align_block_size(topk_ids, block_m, num_experts)

    act = torch.empty((num_tokens * top_k, intermediate), device=device, dtype=dtype)
    if w1_interleaved:
        _moe_gemm(hidden_states, w1, act, w1_scale, flat_weights, sorted_ids, expert_ids, num_post,
                  top_k, 2, wfmt, g1[0], g1[1], block_m)
    else:
        gate_up = torch.empty((num_tokens * top_k, two_inter), device=device, dtype=dtype)
        _moe_gemm(hidden_states, w1, gate_up, w1_scale, flat_weights, sorted_ids, expert_ids, num_post,
                  top_k, False, wfmt, g1[0], g1[1], block_m)
        L.check(
            L.lib().ll_silu_and_mul(act.data_ptr(), gate_up.data_ptr(), num_tokens * top_k, intermediate,
                                    L.dtype_code(dtype), L.stream_ptr()),
            "silu_and_mul",
        )

    # GEMM2 gathers per-slot rows of ``act`` (top_k = 1 makes slot // top_k the identity,
    # fused_moe.py:420-430) and folds the router weight in fp32.
    expanded = torch.empty((num_tokens * top_k, hidden), device=device, dtype=dtype)
    _moe_gemm(act, w2, expanded, w2_scale, flat_weights, sorted_ids, expert_ids, num_post,
              1, True, wfmt, g2[0], g2[1], block_m)

    if slots_ok and top_k <= 8 and hidden % 8 == 0 and hidden <= 8192:
        # (extension) the caller's add-and-normalise adds the k slots up itself: no moe_sum launch
        from .norm_act import SlotSums
        return SlotSums(expanded.view(num_tokens, top_k, hidden), (num_tokens, hidden))
    out = torch.empty((num_tokens, hidden), device=device, dtype=dtype)
    L.check(
        L.lib().ll_moe_sum(out.data_ptr(), expanded.data_ptr(), num_tokens, top_k, hidden,
                           L.dtype_code(dtype), L.stream_ptr()),
        "moe_sum",
    )
    return out


@torch.no_grad()
def moe_route_topk(router_logits: torch.Tensor, top_k: int, norm_topk_prob: bool = True):
    """Extension (decode-step launch count): the router's tail -- fp32 softmax over all experts, top-k, optional
    renormalisation, cast to the activation dtype (models/qwen3_moe.py:95-100) -- as ONE launch instead of four tensor
    ops.  ``router_logits [tokens, experts]`` fp16 / bf16 -> ``(weights [tokens, top_k] in that dtype, ids int64)``."""
    L.require_cuda(router_logits)
    if router_logits.dim() != 2 or router_logits.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("router logits must be a 16-bit [tokens, experts] tensor")
    if router_logits.stride(1) != 1:
        router_logits = router_logits.contiguous()
    t, e = router_logits.shape
    w = torch.empty((t, top_k), dtype=router_logits.dtype, device=router_logits.device)
    ids = torch.empty((t, top_k), dtype=torch.int64, device=router_logits.device)
    L.check(L.lib().ll_moe_route_topk(w.data_ptr(), ids.data_ptr(), router_logits.data_ptr(), t, e, router_logits.stride(0),
                                      int(top_k), 1 if norm_topk_prob else 0, L.dtype_code(router_logits.dtype),
                                      L.stream_ptr()), "moe_route_topk")
    return w, ids


@torch.no_grad()
def moe_router(x: torch.Tensor, gate_weight: torch.Tensor, top_k: int, norm_topk_prob: bool = True, align_block: int = 0):
    """Extension (round 6): the WHOLE router of a decode batch without a library GEMM -- ``F.linear(x, gate_weight)`` (the
    reference's unquantised fp16 gate, models/qwen3_moe.py:102-111) as split-K fp32 planes from a grid of one-wave MFMA
    workgroups, then :func:`moe_route_topk`'s arithmetic over the planes' sum rounded once to the activation dtype.
    ``x [tokens, hidden]``, ``gate_weight [experts, hidden]`` of the same 16-bit dtype -> ``(weights [tokens, top_k], ids int64)``;
    ``None`` when the shape is not served (more than 64 tokens, experts % 32, hidden % 128: the caller keeps GEMM + tail).
    ``align_block > 0``: a third result ``(sorted_token_ids, expert_ids, num_tokens_post_padded, align_block)`` =
    ``moe_align_block_size(ids, align_block, experts)`` computed inside the tail's launch (``fused_moe(..., aligned=...)``)."""
    if (not x.is_cuda or x.dim() != 2 or gate_weight.dim() != 2 or x.dtype not in (torch.float16, torch.bfloat16)
            or gate_weight.dtype != x.dtype or x.shape[1] != gate_weight.shape[1]):
        return None
    t, h = x.shape
    e = gate_weight.shape[0]
    if not L.lib().ll_moe_router_supported(t, e, h) or top_k > e or top_k > 64:
        return None
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    if gate_weight.stride(1) != 1 or gate_weight.stride(0) % 8 or gate_weight.data_ptr() % 16:
        return None
    planes = torch.empty(L.lib().ll_moe_router_workspace_floats(t, e, h), dtype=torch.float32, device=x.device)
    w = torch.empty((t, top_k), dtype=x.dtype, device=x.device)
    ids = torch.empty((t, top_k), dtype=torch.int64, device=x.device)
    aligned, ptrs = None, (None, None, None, None)
    if align_block > 0:
        max_padded = t * top_k + e * (align_block - 1)
        sorted_ids = torch.empty((max_padded,), dtype=torch.int32, device=x.device)
        expert_ids = torch.empty(((max_padded + align_block - 1) // align_block,), dtype=torch.int32, device=x.device)
        num_post = torch.empty((1,), dtype=torch.int32, device=x.device)
        _, counters = L.gemm_workspace(x.device, t, e, h)  # zeroed int32 words every kernel leaves at zero
        aligned = (sorted_ids, expert_ids, num_post, int(align_block))
        ptrs = (sorted_ids.data_ptr(), expert_ids.data_ptr(), num_post.data_ptr(), counters.data_ptr())
    L.check(L.lib().ll_moe_router(w.data_ptr(), ids.data_ptr(), x.data_ptr(), gate_weight.data_ptr(), planes.data_ptr(), t, e, h,
                                  x.stride(0), gate_weight.stride(0), int(top_k), 1 if norm_topk_prob else 0, L.dtype_code(x.dtype),
                                  int(align_block), *ptrs, L.stream_ptr()), "moe_router")
    return (w, ids, aligned) if align_block > 0 else (w, ids)
