"""`FrequencyDecompose(type, size, h, w, inverse)` -- drop-in for the reference module of the same name
(net/utils/frequency_decompose.py:5-125), computed by the HIP partial-DFT kernels (csrc/fw_heads.hip)
instead of torch.fft.  Same constructor, same output shapes:
    'frequency_decompose'   bands [0,s) ... [1-s,1]          -> [nb,   B, C, h, w]
    'frequency_decompose_1' DC, (0,s] ... (1-s,1]            -> [nb+1, B, C, h, w]
    'frequency_decompose_dc' mean / residual                 -> [2,    B, C, h, w]
    inverse=False -> [..., 2] (re, im) of the masked, un-shifted spectrum;  inverse='visual' -> magnitudes.
Square power-of-two maps up to 256x256 (the model uses 128, or 256 with --patch_size 256); anything else raises NotImplementedError.
"""
import math

import torch
from torch import nn

from fwair import lfs
from fwair.lib import call


_panels = {}


def _dft_panels(N, device):
    """f32 [3][N][N]: cos, sin, -sin of 2 pi u i / N (the MFMA decomposition reads them as ready-made fragments from L2)."""
    key = (N, str(device))
    if key not in _panels:
        ang = 2 * math.pi * torch.outer(torch.arange(N, dtype=torch.float64), torch.arange(N, dtype=torch.float64)) / N
        _panels[key] = torch.stack([torch.cos(ang), torch.sin(ang), -torch.sin(ang)]).float().contiguous().to(device)
    return _panels[key]


def _bands(x, mask, mode, partition=False, dc_bits=0):
    """x: f32 [n, N, N] -> mode 0: Re IDFT2(mask_b * DFT2 x) [nb, n, N, N];  mode 1: (re, im) of mask_b * DFT2 x [nb, n, N, N, 2];
    mode 2: |.| in fftshift-ed coordinates."""
    n, N = x.shape[0], x.shape[1]
    nb = mask.shape[0]
    if mode == 0 and partition and nb >= 2 and N in (64, 128):
        # the band masks partition the spectrum: the whole decomposition in one launch on the f32 MFMA (csrc/fw_heads.hip)
        out = torch.empty((nb, n, N, N), dtype=torch.float32, device=x.device)
        call('fw_dft2_decompose', x, mask, _dft_panels(N, x.device), out, n, N, nb, int(dc_bits))
        return out
    fr = torch.empty((n, N, N), dtype=torch.float32, device=x.device)
    fi = torch.empty_like(fr)
    call('fw_dft2_fwd', x, fr, fi, n, N)
    out = torch.empty((nb, n, N, N, 2) if mode == 1 else (nb, n, N, N), dtype=torch.float32, device=x.device)
    if mode == 0 and partition and nb >= 2:
        # the band masks partition the spectrum (sum_b M_b = 1, frequency_decompose.py:47-60), so the band images sum to x:
        # nb-1 masked inverse transforms, the last (widest) band by subtraction
        call('fw_dft2_bands', fr, fi, mask, out, n, N, nb - 1, 0)
        call('fw_band_residual', x, out, n, N, nb)
    else:
        call('fw_dft2_bands', fr, fi, mask, out, n, N, nb, mode)
    return out


class _BandsFn(torch.autograd.Function):
    """Differentiable band decomposition (the frequency L1 loss of train.py:69-70,90-91 back-propagates through it).
    The DFT is linear, so the backward pass is the adjoint transform, evaluated by the same kernels:
      mode 0 (band images)   out_b = Re(A_b x), A_b = F^-1 M_b F is Hermitian for a real mask  =>  dx = sum_b Re(A_b dout_b)
      mode 1 (spectra)       out_b = M_b F x (re, im)                                          =>  dx = N^2 Re F^-1( sum_b M_b (dre_b + i dim_b) )"""

    @staticmethod
    def forward(ctx, x, mask, mode):
        ctx.mask, ctx.mode = mask, mode
        return _bands(x, mask, mode)

    @staticmethod
    def backward(ctx, dout):
        mask, mode = ctx.mask, ctx.mode
        dout = dout.contiguous().float()
        nb, n, N = dout.shape[0], dout.shape[1], dout.shape[2]
        if mode == 0:
            dx = None
            for b in range(nb):
                t = _bands(dout[b], mask[b:b + 1], 0)[0]
                dx = t if dx is None else dx + t
            return dx, None, None
        if mode == 1:
            m = mask.unsqueeze(1)                                            # [nb, 1, N, N]
            gr = (dout[..., 0] * m).sum(0).contiguous()
            gi = (dout[..., 1] * m).sum(0).contiguous()
            ones = torch.ones((1, N, N), dtype=torch.float32, device=dout.device)
            out = torch.empty((1, n, N, N), dtype=torch.float32, device=dout.device)
            call('fw_dft2_bands', gr, gi, ones, out, n, N, 1, 0)             # Re IDFT2 (already divided by N^2)
            return out[0] * float(N * N), None, None
        raise NotImplementedError("FrequencyDecompose(inverse='visual') is a magnitude plot, not a differentiable output")


class FrequencyDecompose(nn.Module):
    def __init__(self, type, size, h, w, inverse=True):
        super().__init__()
        self.type, self.size, self.h, self.w, self.inverse = type, size, h, w, inverse
        assert size > 0 and size <= 1, 'invalid frequency band width(size=%s)' % (size)
        self._masks = {}
        self._partition = False
        self._dc_bits = 0
        if self.type in ['frequency_decompose', 'frequency_decompose_1']:
            if h != w or h & (h - 1) or not 8 <= h <= 256:
                raise NotImplementedError('HIP band decomposition handles square power-of-two maps, 8 <= N <= 256')
            self.num_bands = math.floor(1. / self.size + 0.1)

    def _mask(self, device):
        key = str(device)
        if key not in self._masks:
            m = torch.stack(lfs.band_masks_shifted(self.type, self.size, self.h, self.w)).float()
            self._partition = bool((m.sum(0) == 1).all())                                        # every frequency in exactly one band
            un = torch.fft.ifftshift(m, dim=(-2, -1)).contiguous()
            self._dc_bits = sum(1 << b for b in range(un.shape[0]) if float(un[b].sum()) == 1.0 and float(un[b, 0, 0]) == 1.0)
            self._masks[key] = un.to(device)                                                     # host-built constant
        return self._masks[key]

    def forward(self, x):
        B, C, N = x.shape[0], x.shape[1], x.shape[2]
        need_grad = x.requires_grad and torch.is_grad_enabled()
        n = B * C
        if self.type not in ['frequency_decompose', 'frequency_decompose_1']:
            if need_grad:                                                # mean / residual split: two lines of tensor algebra, autograd-native
                mean = x.float().mean(dim=(-2, -1), keepdim=True).expand_as(x)
                return torch.stack([mean, x.float() - mean], 0)
            xc = x.detach().contiguous().float()
            out = torch.empty((2, B, C, N, x.shape[3]), dtype=torch.float32, device=x.device)
            call('fw_dc_split', xc, out, n, N * x.shape[3])
            return out
        assert N == self.h and x.shape[3] == self.w
        mask = self._mask(x.device)
        nb = mask.shape[0]
        mode = 0 if self.inverse is True else 1 if self.inverse is False else 2
        assert self.inverse in (True, False, 'visual')
        xf = x.contiguous().float().reshape(n, N, N)
        out = _BandsFn.apply(xf, mask, mode) if need_grad else _bands(xf.detach(), mask, mode, self._partition, self._dc_bits)
        out = out.reshape((nb, B, C, N, N, 2) if mode == 1 else (nb, B, C, N, N))
        if mode == 2:
            # the reference's fftshift has no dim argument: it also rolls the batch and channel axes (:32)
            out = torch.roll(out, shifts=(B // 2, C // 2), dims=(1, 2))
        return out
