import torch
M=32768
for name,n,k in [("qkv",4608,3584),("o",3584,3584),("gateup",37888,3584),("down",3584,18944)]:
    x=(torch.randn(M,k,device="cuda")*0.5).half(); w=(torch.randn(n,k,device="cuda")*0.02).half()
    for _ in range(2): y=torch.nn.functional.linear(x,w)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): y=torch.nn.functional.linear(x,w)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/4
    print(name, round(ms,3), "ms", round(2.0*M*n*k/ms/1e9,1), "TFLOP/s (hipBLASLt fp16)")
    del x,w,y
