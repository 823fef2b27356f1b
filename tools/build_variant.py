"""A/B builds: compile ONE kernel source with extra flags and link it with the objects of the regular build.

    python tools/build_variant.py NAME SOURCE.hip [-DFLAG ...]   ->  lite_llama_amd/lib/ab/NAME.so

Use with LL_LIB_OVERRIDE=<that path> (benchmarks only -- never in tests / bench.py defaults)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lite_llama_amd import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
objdir = os.path.join(B.LIBDIR, "obj")
abdir = os.path.join(B.LIBDIR, "ab")
os.makedirs(abdir, exist_ok=True)
base = os.path.basename(src)[:-4]
obj = os.path.join(abdir, f"{name}_{base}.o")
subprocess.run([B._hipcc(), *B.FLAGS, *flags, "-c", os.path.join(B.CSRC, os.path.basename(src)), "-o", obj], check=True)
objs = [obj] + [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != base + ".o"]
out = os.path.join(abdir, name + ".so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", out], check=True)
print(out)
