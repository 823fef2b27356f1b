import sys, torch
sys.path.insert(0, ".")
from lite_llama_amd.quantization.params import quantize_int4_groupwise
from oracle import oracle as O
torch.manual_seed(3)
w = (torch.randn(4608, 3584) * 0.02).half()
qc, sc, zc = quantize_int4_groupwise(w, 128)
qg, sg, zg = quantize_int4_groupwise(w.cuda(), 128)
qo, so, zo = O.quantize_int4_groupwise(w, 128)
print("cpu product vs oracle:", torch.equal(qc, qo), torch.equal(sc, so), torch.equal(zc, zo))
print("gpu vs cpu: scales equal", torch.equal(sg.cpu(), sc), "zeros equal", torch.equal(zg.cpu(), zc), "words differing", (qg.cpu() != qc).sum().item(), "of", qc.numel())
kk = torch.arange(3584)
nc = (qc[:, kk // 8] >> (4 * (kk % 8))) & 0xF
ng = (qg.cpu()[:, kk // 8] >> (4 * (kk % 8))) & 0xF
d = (nc != ng)
print("nibbles differing", d.sum().item(), "max |dq|", (nc - ng).abs().max().item(), "scale rel diff max", ((sg.cpu() - sc).abs() / sc).max().item())
from lite_llama_amd.quantization.params import quantize_int8_per_channel, quantize_int8_groupwise, quantize_fp8_per_channel
for fn in (quantize_int8_per_channel, quantize_int8_groupwise, quantize_fp8_per_channel):
    a = fn(w); b = fn(w.cuda())
    print(fn.__name__, "codes equal", torch.equal(a[0], b[0].cpu()), "scales equal", torch.equal(a[1], b[1].cpu()))
