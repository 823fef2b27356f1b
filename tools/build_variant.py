"""A/B builds: compile ONE kernel source with extra flags and link it with the objects of the regular build.

    python tools/build_variant.py NAME SOURCE.hip [-DFLAG ...]   ->  lite_llama_amd/lib/ab/NAME.so

Use with LL_LIB_OVERRIDE=<that path> (benchmarks only -- never in tests / bench.py defaults).

    python tools/build_variant.py --regs SOURCE.hip [substring] [-DFLAG ...]

prints the register / spill / scratch numbers of the source's kernels (code-object metadata of a -save-temps build) instead."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lite_llama_amd import build as B

if sys.argv[1] == "--regs":
    import re, tempfile
    src = sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else ""
    flags = [a for a in sys.argv[3:] if a.startswith("-")]
    with tempfile.TemporaryDirectory() as d:
        # device-only assembly (not -save-temps: the host pass of a -save-temps build rejects the kernarg-preload clones of gemm_short.hip)
        asm = os.path.join(d, "k.s")
        subprocess.run([B._hipcc(), *[f for f in B.flags_for(src) if f != "-fPIC"], *flags, "--cuda-device-only", "-S",
                        os.path.join(B.CSRC, os.path.basename(src)), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    keys = ["sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size"]
    for blk in text.split("  - .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or pat not in nm.group(1):
            continue
        vals = {k: re.search(rf"\.{k}:\s+(\d+)", blk) for k in keys}
        dem = subprocess.run(["c++filt", nm.group(1)], capture_output=True, text=True).stdout.strip()
        print(dem[:70].ljust(70), " ".join(f"{k.replace('_count', '').replace('_fixed_size', '')}={v.group(1) if v else '?'}" for k, v in vals.items()))
    sys.exit(0)
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
objdir = os.path.join(B.LIBDIR, "obj")
abdir = os.path.join(B.LIBDIR, "ab")
os.makedirs(abdir, exist_ok=True)
base = os.path.basename(src)[:-4]
obj = os.path.join(abdir, f"{name}_{base}.o")
subprocess.run([B._hipcc(), *B.flags_for(src), *flags, "-c", os.path.join(B.CSRC, os.path.basename(src)), "-o", obj], check=True)
objs = [obj] + [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != base + ".o"]
out = os.path.join(abdir, name + ".so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", out], check=True)
print(out)
