"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import shim that lets the *unmodified* reference (/root/reference, read-only)
be imported in this GPU-less container so that golden vectors can be generated
from it.  It
  * registers a 3-symbol stand-in for ``timm.models.layers`` (DropPath,
    to_2tuple, trunc_normal_) -- timm is not installed and cannot be fetched;
  * makes ``.cuda()`` a no-op (the reference hard-codes it, e.g.
    net/utils/frequency_decompose.py:17-26, net/utils/moco.py:161);
  * sets ``sys.argv`` before ``option.py`` parses it at import time.

Nothing here is copied from the reference; the reference itself is only ever
imported from where it lies and never travels to the GPU box.
"""
import sys
import types
import collections.abc
from itertools import repeat

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


class _DropPath(nn.Module):
    """Per-sample stochastic depth (timm semantics): keep-mask ~ Bernoulli(1-p),
    scaled by 1/(1-p); identity when p == 0 or in eval mode."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return x * mask


def _to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return tuple(repeat(x, 2))


def install(argv=None):
    """Install the shim and return the imported reference ``options`` object."""
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        layers.DropPath = _DropPath
        layers.to_2tuple = _to_2tuple
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = models
        models.layers = layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.layers"] = layers
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.argv = ["ref"] + list(argv or [])
    import option  # noqa: the reference's import-time argparse singleton

    return option.options
