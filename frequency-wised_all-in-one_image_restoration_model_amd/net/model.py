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
from fwair.convnets import DGRN, ResNetEncoder                  # noqa: F401  (registered names of the seam)
from fwair.modules import MoCo, UformerDecoder, UformerEncoder  # noqa: F401
from fwair.vit import ViTEncoder                                # noqa: F401

# Registered names of the seam (net/model.py:3,17,31 of the reference: `DGRN as ResNetDecoder`).  In the reference the ResNet /
# ViT encoders fail in TRAIN mode inside MoCo (range(opt.L) heads indexed on their 1-element output, moco.py:127-128) and the DGRN
# decoder asserts in its deformable convolution (deform_conv.py:64); here MoCo runs len(q) heads and DCNv2 is implemented
# (parity unpinned, see fwair/convnets.py).
ResNetDecoder = DGRN


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
