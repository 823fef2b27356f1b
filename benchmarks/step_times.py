"""Per-replay duration of the captured decode step (HIP events recorded between the graph launches): how long do the first
replays of a fresh graph take, and how steady is the step afterwards?  Headline model by default."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lite_llama_amd.executor import DecodeEngine
from lite_llama_amd.model import GEOMETRY, CausalLM
from lite_llama_amd.quantization import QuantConfig

steps = int(os.environ.get("STEPS", 60))
geo = GEOMETRY[os.environ.get("MODEL", "qwen2.5-7b")]
quant = QuantConfig.for_runtime_scheme(os.environ.get("QUANT", "int4"))
B = int(os.environ.get("BATCH", 64))
with torch.device("cuda"):
    model = CausalLM(geo, quant)
model.init_synthetic(seed=0, quant=quant, device="cuda")
if hasattr(model, "compact_weights"):
    model.compact_weights()
eng = DecodeEngine(model, max_batch=B, max_seq_len=512 + steps + 8, device="cuda")
first = eng.synthetic_context(B, 512, seed=1)
if os.environ.get("IDLE"):
    torch.cuda.synchronize(); time.sleep(float(os.environ["IDLE"]))
evs, host = [], []

def on_step(i):
    host.append(time.perf_counter())   # host time between two calls = the previous replay's hipGraphLaunch (+ this hook)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)

t0 = time.perf_counter()
eng.decode(first, steps, use_graph=True, on_step=on_step)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
d = [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(len(evs) - 1)]
h = [round((host[i + 1] - host[i]) * 1e3, 3) for i in range(len(host) - 1)]
print(json.dumps({"host_launch_first10_ms": h[:10], "host_launch_median_ms": sorted(h)[len(h) // 2], "host_launch_max_ms": max(h),
                  "first10_ms": d[:10], "median_ms": sorted(d)[len(d) // 2], "max_after10_ms": max(d[10:]), "wall_s_incl_capture": round(wall, 3)}))
