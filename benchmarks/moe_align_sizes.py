"""moe_align_block_size at prefill sizes: the multi-workgroup form (three launches over a workspace) against the one-workgroup
kernel (the raw ABI entry).  us per call, hipGraph-free event timing over 20 calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lite_llama_amd._lib as L
from lite_llama_amd.kernels.fused_moe import moe_align_block_size

dev = "cuda"
for tokens, topk, experts, block in [(129, 8, 128, 64), (256, 8, 128, 64), (512, 8, 128, 64), (2048, 8, 128, 64), (32768, 8, 128, 64)]:
    ids = torch.randint(0, experts, (tokens, topk), dtype=torch.int64, device=dev)
    slots = tokens * topk
    outs = moe_align_block_size(ids, block, experts)
    one = [torch.empty_like(t) for t in outs]

    def new():
        moe_align_block_size(ids, block, experts)

    def old():
        L.check(L.lib().ll_moe_align_block_size(ids.data_ptr(), L.index_width(ids.view(-1)), slots, experts, block, one[0].data_ptr(),
                                                one[1].data_ptr(), one[2].data_ptr(), L.stream_ptr()), "align")

    res = []
    for fn, reps in ((new, 20), (old, 20 if slots <= 20000 else 3)):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1e3 / reps)
    same = all(torch.equal(x, y) for x, y in zip(outs, one))
    print(f"tokens {tokens:6d} x top-{topk} of {experts}: multi-workgroup {res[0]:9.1f} us   one workgroup {res[1]:9.1f} us   equal {same}")
