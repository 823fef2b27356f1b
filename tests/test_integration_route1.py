"""INTEGRATION.md route 1, exercised: with ``lite_llama_amd.integration.install()`` the REFERENCE package's own
modules bind the HIP wrappers (name wiring only -- nothing is launched; needs /root/reference, i.e. the build
container).  Both orders: install before the first ``import lite_llama`` (the reference's Triton kernel modules are
never executed) and install after it (already-bound names are re-pointed)."""

import importlib
import inspect
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF + "/lite_llama"), reason="reference checkout not present (GPU box)")

BOUND = {
    "lite_llama.models.base": ("flash_attention2_no_pad", "flash_decoding", "rope_emb_forward", "skip_rmsnorm",
                               "swiglu_forward", "update_kv_buffer"),
    "lite_llama.executor.model_runner": ("update_kv_index",),
    "lite_llama.models.quantization.methods.unquantized": ("fused_moe",),
    "lite_llama.models.quantization.methods.w4a16": ("w4a16_matmul",),
    "lite_llama.models.quantization.methods.w8a8": ("smoothquant_matmul",),
    "lite_llama.models.quantization.methods.w8a16": ("fused_moe", "w8a16_matmul"),
}


def _forget_reference(monkeypatch):
    for name in [m for m in sys.modules if m == "lite_llama" or m.startswith("lite_llama.")]:
        monkeypatch.delitem(sys.modules, name)


def _check_bound():
    import lite_llama_amd.kernels as amd
    for modname, names in BOUND.items():
        mod = importlib.import_module(modname)
        for name in names:
            assert getattr(mod, name) is getattr(amd, name), (modname, name)


def test_install_before_import(monkeypatch):
    monkeypatch.syspath_prepend(REF)
    _forget_reference(monkeypatch)
    import lite_llama_amd.integration as hip
    rep = hip.install()
    assert rep["shimmed"] == ["lite_llama.kernels", "lite_llama.kernels.quantization"]
    import lite_llama  # noqa: F401  (engine, executor, models: everything the product path imports)
    _check_bound()
    # the reference's Triton kernel modules were never executed
    assert "lite_llama.kernels.flashdecoding" not in sys.modules and "lite_llama.kernels.quantization.w4a16" not in sys.modules
    from lite_llama import ContinuousBatchingEngine, LLM  # noqa: F401
    # same call signatures as the wrappers they replace (read from the reference source, not imported)
    import ast
    import lite_llama_amd.kernels as amd
    for path, fn in (("kernels/skip_rmsnorm.py", "skip_rmsnorm"), ("kernels/flashdecoding.py", "flash_decoding"),
                     ("kernels/quantization/w4a16.py", "w4a16_matmul"), ("kernels/fused_moe.py", "fused_moe")):
        tree = ast.parse(open(f"{REF}/lite_llama/{path}").read())
        node = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == fn)
        ref_args = [a.arg for a in node.args.args] + [a.arg for a in node.args.kwonlyargs]
        mine = [p for p in inspect.signature(getattr(amd, fn)).parameters if not p.startswith("_")]
        assert mine[: len(ref_args)] == ref_args, (fn, ref_args, mine)


def test_install_after_import(monkeypatch):
    pytest.importorskip("triton")
    monkeypatch.setenv("TRITON_INTERPRET", "1")
    monkeypatch.syspath_prepend(REF)
    _forget_reference(monkeypatch)
    import lite_llama  # noqa: F401  binds the Triton wrappers
    import lite_llama_amd.integration as hip
    rep = hip.install()
    assert ("lite_llama.models.base", "skip_rmsnorm") in rep["rebound"]
    _check_bound()
