"""relu / leaky_relu / tanh / gelu -- the four activation names the reference's kernel package
exports (lite_llama/kernels/activations.py:19-57).  In the reference they are ``@triton.jit``
device helpers with no host launch and no caller in the model; here they are plain
elementwise callables on tensors so the exported name set is complete."""

import math

import torch


def relu(x):
    return torch.clamp_min(x, 0)


def leaky_relu(x):
    return torch.where(x >= 0, x, x * 0.01)


def tanh(x):
    return torch.tanh(x)


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
