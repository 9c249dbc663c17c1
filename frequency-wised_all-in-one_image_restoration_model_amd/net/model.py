"""`net.model` -- the reference's model facade (net/model.py:13-71) over the MI355X-native modules.

Same plug-in seam: an encoder / decoder is selected by name, `globals()[opt.encoder_type + 'Encoder']`
(net/model.py:17,31); same forward contract:
    train:  AirNet(x_query, x_key) -> (restored, logits, labels)      eval: -> restored
    Encoder(x_query, x_key) -> (fea, logits, labels, inter) | (fea, inter)
    Decoder(x_query, inter) -> restored
Compute dtype: `opt.compute_dtype` in {'fp32' (default, reference numerics), 'bf16'}.
"""
import torch
from torch import nn

from fwair import functional as _Fn
from fwair.modules import MoCo, UformerDecoder, UformerEncoder


def _unsupported(name, why):
    class _Unsupported(nn.Module):
        def __init__(self, opt):
            raise NotImplementedError(f'{name}: {why}')
    _Unsupported.__name__ = name
    return _Unsupported


# Registered names of the seam.  The ResNet / ViT variants do not run in the reference itself
# (DCN asserts, MoCo head-count mismatch); they are listed so the failure is explicit.
ResNetDecoder = _unsupported('ResNetDecoder', 'DGRN needs mmcv DCNv2, absent from the reference tree; scheduled after the Uformer path')
ResNetEncoder = _unsupported('ResNetEncoder', 'not runnable in the reference (MoCo L-mismatch); scheduled after the Uformer path')
ViTEncoder = _unsupported('ViTEncoder', 'not runnable in train mode in the reference; scheduled after the Uformer path')


def _apply_dtype(opt):
    name = getattr(opt, 'compute_dtype', 'fp32') or 'fp32'
    _Fn.config.compute_dtype = {'fp32': torch.float32, 'float32': torch.float32, 'bf16': torch.bfloat16,
                                'bfloat16': torch.bfloat16}[str(name)]


class Decoder(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.R = globals()[opt.decoder_type + 'Decoder'](opt)

    def forward(self, x_query, inter):
        return self.R(x_query, inter)


class Encoder(nn.Module):
    def __init__(self, opt):
        super().__init__()
        encoder = globals()[opt.encoder_type + 'Encoder']
        self.E = MoCo(opt=opt, base_encoder=encoder, dim=opt.encoder_dim, K=opt.batch_size * 3)     # net/model.py:35

    def forward(self, x_query, x_key, _step_begun=False):
        if self.training:
            if not _step_begun:                              # called on its own (phase 1 of train.py:82-86): a training step of its own
                _Fn.droppath_begin(x_query.device, 'encoder')
            fea, logits, labels, inter = self.E(x_query, x_key)
            return fea, logits, labels, inter
        fea, inter = self.E(x_query, x_query)
        return fea, inter


class AirNet(nn.Module):
    def __init__(self, opt):
        super().__init__()
        _apply_dtype(opt)
        self.opt = opt
        self.R = Decoder(opt)
        self.E = Encoder(opt)

    def forward(self, x_query, x_key):
        _apply_dtype(self.opt)
        if self.training:
            _Fn.droppath_begin(x_query.device, 'airnet')
            fea, logits, labels, inter = self.E(x_query, x_key, True)
            restored = self.R(x_query, inter)
            return restored, logits, labels
        fea, inter = self.E(x_query, x_query)
        return self.R(x_query, inter)
