"""Register / spill / scratch numbers of the kernels of one HIP source (code-object metadata of a -save-temps build).

    python tools/kernel_regs.py lite_llama_amd/csrc/gemm_w4_v3.hip [substring] [-DFLAG ...]
"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lite_llama_amd import build as B

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
flags = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    subprocess.run([B._hipcc(), *B.FLAGS, *flags, "-save-temps=obj", "-c", src, "-o", os.path.join(d, "k.o")], check=True,
                   stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f][0]
    text = open(os.path.join(d, asm)).read()
keys = ["sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"]
for blk in text.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or pat not in name.group(1):
        continue
    vals = {k: re.search(rf"\.{k}:\s+(\d+)", blk) for k in keys}
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    print(dem[:70].ljust(70), " ".join(f"{k.replace('_count','').replace('_fixed_size','')}={v.group(1) if v else '?'}" for k, v in vals.items()))
