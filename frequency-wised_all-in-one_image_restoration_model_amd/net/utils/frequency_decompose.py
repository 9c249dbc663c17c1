"""`FrequencyDecompose(type, size, h, w, inverse)` -- drop-in for the reference module of the same name
(net/utils/frequency_decompose.py:5-125), computed by the HIP partial-DFT kernels (csrc/fw_heads.hip)
instead of torch.fft.  Same constructor, same output shapes:
    'frequency_decompose'   bands [0,s) ... [1-s,1]          -> [nb,   B, C, h, w]
    'frequency_decompose_1' DC, (0,s] ... (1-s,1]            -> [nb+1, B, C, h, w]
    'frequency_decompose_dc' mean / residual                 -> [2,    B, C, h, w]
    inverse=False -> [..., 2] (re, im) of the masked, un-shifted spectrum;  inverse='visual' -> magnitudes.
Square power-of-two maps up to 128x128 (what the model uses); anything else raises NotImplementedError.
"""
import math

import torch
from torch import nn

from fwair import lfs
from fwair.lib import call


class FrequencyDecompose(nn.Module):
    def __init__(self, type, size, h, w, inverse=True):
        super().__init__()
        self.type, self.size, self.h, self.w, self.inverse = type, size, h, w, inverse
        assert size > 0 and size <= 1, 'invalid frequency band width(size=%s)' % (size)
        self._masks = {}
        if self.type in ['frequency_decompose', 'frequency_decompose_1']:
            if h != w or h & (h - 1) or not 8 <= h <= 128:
                raise NotImplementedError('HIP band decomposition handles square power-of-two maps, 8 <= N <= 128')
            self.num_bands = math.floor(1. / self.size + 0.1)

    def _mask(self, device):
        key = str(device)
        if key not in self._masks:
            m = torch.stack(lfs.band_masks_shifted(self.type, self.size, self.h, self.w)).float()
            self._masks[key] = torch.fft.ifftshift(m, dim=(-2, -1)).contiguous().to(device)     # host-built constant
        return self._masks[key]

    def forward(self, x):
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError('FrequencyDecompose backward (only needed by --num_frequency_bands_l1) is not on the hot path yet')
        B, C, N = x.shape[0], x.shape[1], x.shape[2]
        x = x.detach().contiguous().float()
        n = B * C
        if self.type not in ['frequency_decompose', 'frequency_decompose_1']:
            out = torch.empty((2, B, C, N, x.shape[3]), dtype=torch.float32, device=x.device)
            call('fw_dc_split', x, out, n, N * x.shape[3])
            return out
        assert N == self.h and x.shape[3] == self.w
        mask = self._mask(x.device)
        nb = mask.shape[0]
        fr = torch.empty((n, N, N), dtype=torch.float32, device=x.device)
        fi = torch.empty_like(fr)
        call('fw_dft2_fwd', x, fr, fi, n, N)
        if self.inverse is True:
            out = torch.empty((nb, B, C, N, N), dtype=torch.float32, device=x.device)
            call('fw_dft2_bands', fr, fi, mask, out, n, N, nb, 0)
        elif self.inverse is False:
            out = torch.empty((nb, B, C, N, N, 2), dtype=torch.float32, device=x.device)
            call('fw_dft2_bands', fr, fi, mask, out, n, N, nb, 1)
        elif self.inverse == 'visual':
            out = torch.empty((nb, B, C, N, N), dtype=torch.float32, device=x.device)
            call('fw_dft2_bands', fr, fi, mask, out, n, N, nb, 2)
            # the reference's fftshift has no dim argument: it also rolls the batch and channel axes (:32)
            out = torch.roll(out, shifts=(B // 2, C // 2), dims=(1, 2))
        else:
            raise AssertionError
        return out
