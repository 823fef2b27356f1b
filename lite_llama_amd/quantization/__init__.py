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
