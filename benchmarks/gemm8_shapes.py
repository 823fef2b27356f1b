"""Graph-replayed timing of the 8-bit GEMM paths (W8A16 int8, W8A8) at the Qwen2.5-7B gate/up shape."""
import torch, sys
sys.path.insert(0, "/root/repo")
import lite_llama_amd.kernels as K
dev="cuda"
for name, fn in [("w8a16-int8", lambda x,w,s: K.w8a16_matmul(x, w, s, group_n=1, group_k=3584)),
                 ("w8a8", lambda x,w,s: K.smoothquant_matmul(x, w, s.view(-1)))]:
    n,k=18944,3584
    ws=[torch.randint(-127,128,(n,k),dtype=torch.int8,device=dev) for _ in range(6)]
    s=torch.rand(n,1,device=dev)*0.01
    x=torch.randn(64,k,device=dev,dtype=torch.float16)
    fn(x,ws[0],s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(12): fn(x,ws[i%6],s)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/60
    print(f"{name}: {us:.1f} us/launch  {n*k/us/1e6:.2f} TB/s")
