from .config import FP8, INT4, INT8, SMOOTHQUANT, QuantConfig
from .methods import (
    LinearQuantMethod,
    MoeQuantMethod,
    SmoothQuantLinearMethod,
    UnquantizedLinearMethod,
    UnquantizedMoeMethod,
    W4A16LinearMethod,
    W8A16LinearMethod,
    W8A16MoeMethod,
    get_linear_method,
    get_moe_method,
)
from .params import (
    quantize_fp8_per_channel,
    quantize_int4_groupwise,
    quantize_int8_groupwise,
    quantize_int8_per_channel,
)
from .checkpoint_layouts import awq_to_w4a16, gptq_to_w4a16, load_int4_checkpoint_linear  # noqa: E402,F401
