"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the counter-based Bernoulli masks of the HIP kernels
(csrc/fw_common.h: fw_hash32 / fw_site_key / fw_keep; users: fw_dropout, fw_gattn_fwd / fw_gattn_bwd).

The product draws nn.Dropout masks (net/encoder_ViT.py:31,33,67,73,158: p = 0.1) as a pure function of
(seed, call site, flat element index) so that the backward pass can re-derive them.  The oracle -- and the golden
generator, which patches torch.nn.Dropout.forward of the imported reference with these masks -- evaluates the same
integers, so train-mode parity with Dropout ON is a deterministic statement.  Only `tests/`, `smoke()` and
`bench.py`'s cpu_baseline may import this module.
"""
import zlib

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _hash32(x):
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def site_key(seed, site):
    s = _hash32(np.array([(int(site) * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF], dtype=np.uint64))
    return _hash32(np.array([int(seed) & 0xFFFFFFFF], dtype=np.uint64) ^ s)[0]


def thresh(p):
    t = float(np.float32(p)) * 4294967296.0          # the C side multiplies the f32 argument as a double
    return 0 if t <= 0 else min(int(t), 4294967295)


def keep_mask(seed, site, shape, p):
    """bool array `shape`: element with flat index i is kept iff hash(hash(lo(i) ^ key) + hi(i)) >= p * 2^32."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    key = np.uint64(site_key(seed, site))
    r = _hash32((_hash32((idx & M32) ^ key) + (idx >> np.uint64(32))) & M32)
    return (r >= np.uint64(thresh(p))).reshape(shape)


def site_base(prefix):
    """Call-site base of one encoder copy (fwair/vit.py: ViTEncoder.set_prefix)."""
    return ((zlib.crc32(prefix.encode()) & 0xFFFF) << 8) if prefix else 0


def vit_site(base, layer, which):
    """which: 'attn' (attention map), 'out' (to_out), 'hidden' (FeedForward after GELU), 'ff' (FeedForward output), 'emb'."""
    if which == 'emb':
        return base + 0xFF
    return base + 4 * layer + {'attn': 0, 'out': 1, 'hidden': 2, 'ff': 3}[which]
