"""INTEGRATION.md route 1 as code: make the reference package (harleyszhang/lite_llama) bind the MI355X kernels.

The reference binds its kernels at import time (``from ..kernels import skip_rmsnorm, ...`` in
``lite_llama/models/base.py:31-38``, ``models/quantization/methods/*.py``, ``executor/model_runner.py:22``), and
``import lite_llama`` itself already imports those modules (``lite_llama/__init__.py`` pulls in the engine).  So the
swap has to be in place BEFORE the first ``import lite_llama``: :func:`install` registers this package's kernel layer
in ``sys.modules`` under the names the reference imports (``lite_llama.kernels`` and
``lite_llama.kernels.quantization``) -- the reference's Triton modules are then never executed.  If ``lite_llama`` was
imported earlier, the already-bound names of the importing modules are re-pointed instead.

    import lite_llama_amd.integration as hip
    hip.install()
    import lite_llama          # TextGenerator / ContinuousBatchingEngine now launch the HIP kernels
"""

from __future__ import annotations

import importlib
import sys
import types

# modules of the reference that bind kernel names at import time, and the names each binds
_BINDINGS = {
    "lite_llama.models.base": ("flash_attention2_no_pad", "flash_decoding", "rope_emb_forward", "skip_rmsnorm",
                               "swiglu_forward", "update_kv_buffer"),
    "lite_llama.executor.model_runner": ("update_kv_index",),
    "lite_llama.models.quantization.methods.unquantized": ("fused_moe",),
    "lite_llama.models.quantization.methods.w4a16": ("w4a16_matmul",),
    "lite_llama.models.quantization.methods.w8a8": ("smoothquant_matmul",),
    "lite_llama.models.quantization.methods.w8a16": ("fused_moe", "w8a16_matmul"),
}


def install() -> dict:
    """Returns ``{"shimmed": [...module names registered...], "rebound": [...(module, name) re-pointed...]}``."""
    import lite_llama_amd.kernels as amd
    import lite_llama_amd.kernels.quantization as amd_q

    report = {"shimmed": [], "rebound": []}
    if "lite_llama" not in sys.modules:
        pkg = types.ModuleType("lite_llama.kernels")
        pkg.__path__ = []  # a package: ``from ....kernels.quantization import ...`` resolves through sys.modules
        pkg.__doc__ = "lite_llama_amd kernel layer registered under the reference's package name"
        for name in amd.__all__:
            setattr(pkg, name, getattr(amd, name))
        pkg.__all__ = list(amd.__all__)
        quant = types.ModuleType("lite_llama.kernels.quantization")
        for name in ("w4a16_matmul", "w8a16_matmul", "smoothquant_matmul"):
            setattr(quant, name, getattr(amd_q, name))
        quant.__all__ = ["w8a16_matmul", "w4a16_matmul", "smoothquant_matmul"]
        pkg.quantization = quant
        sys.modules["lite_llama.kernels"] = pkg
        sys.modules["lite_llama.kernels.quantization"] = quant
        report["shimmed"] = ["lite_llama.kernels", "lite_llama.kernels.quantization"]
        return report
    # the reference is already imported: re-point what its modules bound
    for modname, names in _BINDINGS.items():
        mod = sys.modules.get(modname)
        if mod is None:
            try:
                mod = importlib.import_module(modname)
            except Exception:  # optional module of another reference version
                continue
        for name in names:
            if hasattr(mod, name):
                setattr(mod, name, getattr(amd, name))
                report["rebound"].append((modname, name))
    for modname in ("lite_llama.kernels", "lite_llama.kernels.quantization"):
        mod = sys.modules.get(modname)
        if mod is not None:
            for name in amd.__all__:
                if hasattr(mod, name):
                    setattr(mod, name, getattr(amd, name))
    return report
