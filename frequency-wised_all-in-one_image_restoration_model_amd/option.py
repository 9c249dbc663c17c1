"""`option.options` -- the import-time argparse singleton every reference script imports
(`from option import options as opt`, reference option.py:3-115).  Flag names, types, defaults and the
derived fields (batch_size, ckpt_path, encoder_dim, lr) follow the reference so that its train.py / test.py /
plot_*.py drive this package unchanged.  Quirks kept on purpose: `type=bool` flags are truthy
for any non-empty string; `contrast_loss_weight` stays None unless passed (the reference computes a default into
a local variable it never uses, option.py:59-64) -- `default_contrast_loss_weight` exposes that intended value.
Additions (reference-preserving defaults): --compute_dtype, --grad_allreduce_dtype.
"""
import argparse

_FLAGS = [
    # name, kwargs                                                         (reference option.py line)
    ('--cuda', dict(type=int, default=0)),                                                              # :6
    ('--epochs', dict(type=int, default=1000, help='maximum number of epochs to train the total model.')),
    ('--epochs_encoder', dict(type=int, default=100, help='number of epochs to train encoder.')),
    ('--lr', dict(type=float, default=None, help='learning rate of encoder.')),
    ('--contrast_loss_weight', dict(type=float, default=None, help='contrast loss weight in objective function.')),
    ('--frequency_l1_loss_weight', dict(type=float, default=0.1, help='frequency l1 loss weight in objective function.')),
    ('--de_type', dict(nargs='+', type=str, default=['denoising_0', 'deraining', 'dehazing', 'deblurring'],
                       help='which type of degradations are training for.')),
    ('--test_de_type', dict(nargs='+', type=str,
                            default=['denoising_bsd68_15', 'denoising_bsd68_25', 'denoising_bsd68_50', 'deraining', 'dehazing',
                                     'deblurring'], help='which type of degradations are testing for.')),
    ('--patch_size', dict(type=int, default=128, help='patch size of input.')),
    ('--num_workers', dict(type=int, default=16, help='number of workers.')),
    ('--save_imgs', dict(type=bool, default=False, help='whether or not to save output images.')),
    ('--crop_test_imgs_size', dict(type=int, default=128, help='crop test images to smaller than given resolution.')),
    ('--output_path', dict(type=str, default='output/tmp/', help='output and checkpoint save path')),
    ('--encoder_type', dict(type=str, default='Uformer', help='should be in [ResNet, ViT, Uformer]')),
    ('--decoder_type', dict(type=str, default='Uformer', help='should be in [ResNet, Uformer]')),
    ('--encoder_dim', dict(type=int, default=None, help='the output dimensionality of encoder.')),
    ('--frequency_decompose_type', dict(type=str, default='none', help='should be in [%_bands, DC, none].')),
    ('--debug_mode', dict(type=bool, default=False, help='whether or not to enable debug mode.')),
    ('--encoder_embed_dim', dict(type=int, default=28, help='the embedding dimensionality of Uformer Encoder.')),
    ('--embed_dim', dict(type=int, default=56, help='the embedding dimensionality of Uformer Decoder.')),
    ('--degradation_embedding_method', dict(nargs='+', type=str, default=['residual'],
                                            help='degradation embedding method (all_%_bands, all_DC run; see SURVEY 0.1).')),
    ('--learnable_modulator', dict(type=bool, default=False, help='add learnable modulator in Uformer decoder.')),
    ('--num_frequency_bands_encoder', dict(type=int, default=-1)),
    ('--num_frequency_bands', dict(type=int, default=-1)),
    ('--num_frequency_bands_l1', dict(type=int, default=-1)),
    ('--frequency_feature_enhancement_method', dict(nargs='+', type=str, default=[])),
    ('--L', dict(type=int, default=3, help='number of frequency bands used in attention map frequency modulation.')),
    ('--encoder_msa_type', dict(type=str, default='freq', help='should be in [origin, freq].')),
    ('--out_channels', dict(type=int, default=3)),
    ('--batch_wise_decompose', dict(type=bool, default=False)),
    ('--frequency_decompose_type_2', dict(type=bool, default=False)),
    # ---- additions of this package ------------------------------------------------------------------------
    ('--compute_dtype', dict(type=str, default='fp32', choices=['fp32', 'bf16'],
                             help='storage type of activations / GEMM operands in the HIP kernels (accumulation is f32).')),
    ('--grad_allreduce_dtype', dict(type=str, default='fp32', choices=['fp32', 'bf16'],
                                    help='wire type of the data-parallel gradient all-reduce.')),
]

_TASKS = {
    '2tasks': (['denoising_0', 'deraining'], ['denoising_bsd68_15', 'denoising_bsd68_25', 'denoising_bsd68_50', 'deraining']),
    '3tasks': (['denoising_0', 'deraining', 'dehazing'],
               ['denoising_bsd68_15', 'denoising_bsd68_25', 'denoising_bsd68_50', 'deraining', 'dehazing']),
    '4tasks': (['denoising_0', 'deraining', 'dehazing', 'deblurring'],
               ['denoising_bsd68_15', 'denoising_bsd68_25', 'denoising_bsd68_50', 'deraining', 'dehazing', 'deblurring']),
}
_ENCODER_DEFAULTS = {'ResNet': (256, 1e-3), 'ViT': (3, 3e-4), 'Uformer': (256, 2e-4), 'Oformer': (256, 2e-4)}   # option.py:80-101


def build_parser():
    p = argparse.ArgumentParser()
    for name, kw in _FLAGS:
        p.add_argument(name, **kw)
    return p


def finalize(o):
    """Derived fields, option.py:57-115."""
    assert o.L in (2, 3)                                                     # option.py:59-64
    o.default_contrast_loss_weight = 0.6 if o.L == 3 else 0.2
    if o.de_type and o.de_type[0] in _TASKS:
        o.de_type, o.test_de_type = (list(x) for x in _TASKS[o.de_type[0]])
    o.batch_size = len(o.de_type)                                            # option.py:76
    o.ckpt_path = o.output_path + 'ckpt/'
    assert o.encoder_type in _ENCODER_DEFAULTS, 'invalid encoder type.'
    dim, lr = _ENCODER_DEFAULTS[o.encoder_type]
    if o.encoder_dim is None:
        o.encoder_dim = dim
    if o.lr is None:
        o.lr = lr
    t = o.frequency_decompose_type.split('_')
    ok = o.frequency_decompose_type in ('DC', 'none') or (len(t) == 2 and t[0].isdigit() and t[1] == 'bands')
    assert ok, 'invalid frequency decomposition type.'
    return o


parser = build_parser()
options = finalize(parser.parse_args())
