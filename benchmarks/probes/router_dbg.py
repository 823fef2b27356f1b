import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lite_llama_amd.kernels.fused_moe import moe_router, moe_align_block_size
T, E, H, k = [int(a) for a in sys.argv[1:5]]
blk = int(sys.argv[5])
torch.manual_seed(0)
x = (torch.randn(T, H) * 0.5).half().cuda()
gw = (torch.randn(E, H) * 0.05).half().cuda()
w, ids = moe_router(x, gw, k, True)
torch.cuda.synchronize(); print("router ok", ids[0].tolist(), flush=True)
s_ref, e_ref, n_ref = moe_align_block_size(ids, blk, E)
torch.cuda.synchronize(); print("standalone align ok", int(n_ref), flush=True)
w3, ids3, (s, e, n, b) = moe_router(x, gw, k, True, align_block=blk)
torch.cuda.synchronize(); print("fused ok", int(n), torch.equal(s, s_ref), torch.equal(ids3, ids), flush=True)
