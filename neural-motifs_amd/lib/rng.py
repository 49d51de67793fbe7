"""
Randomness used INSIDE the forward pass (dropout masks, recurrent-dropout masks).

Production mode draws masks on the device with torch's generator.  For parity runs a seeded host generator
(`HostRNG`, numpy MT19937) can be installed with `use_host_rng(seed)`: every site then draws its mask on the host
in call order and uploads it, so the CPU oracle (oracle/model.py: HostRNG) consumes the identical stream
(SURVEY.md §8d "LSTM dropout masks injected (not sampled) for parity runs").
"""
import numpy as np
import torch


class HostRNG(object):
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def keep_mask(self, shape, keep_prob, device):
        m = (self.rs.random_sample(tuple(shape)) < keep_prob).astype(np.float32)
        from lib.pytorch_misc import h2d
        return h2d(m, device)


_host_rng = None


def use_host_rng(seed):
    """Install (seed is an int) or remove (seed is None) the deterministic host mask source."""
    global _host_rng
    _host_rng = None if seed is None else HostRNG(seed)
    return _host_rng


def keep_mask(shape, keep_prob, device):
    """float32 {0,1} tensor on `device` with P(1) = keep_prob"""
    if _host_rng is not None:
        return _host_rng.keep_mask(shape, keep_prob, device)
    return torch.empty(tuple(shape), dtype=torch.float32, device=device).bernoulli_(keep_prob)      # one launch


def dropout(x, p, training):
    """inverted dropout: x * mask / (1-p)   (torch.nn.Dropout semantics, mask source switchable)"""
    if not training or p == 0.0:
        return x
    if _host_rng is None:
        return torch.nn.functional.dropout(x, p, True)       # fused mask + scale: one launch forward, one backward
    return x * keep_mask(x.shape, 1.0 - p, x.device) / (1.0 - p)


# torch.nn.AlphaDropout (the SELU RoI head of the ResNet detector branch, reference lib/object_detector.py:89-96):
# dropped units take the SELU saturation value alpha' = -scale * alpha, then the affine (a, b) restores zero mean / unit variance
_ALPHA_PRIME = -1.7580993408473766


def alpha_dropout_coeffs(p):
    a = ((1.0 - p) * (1.0 + p * _ALPHA_PRIME ** 2)) ** -0.5
    return a, -a * _ALPHA_PRIME * p


def alpha_dropout(x, p, training):
    """torch.nn.functional.alpha_dropout with a switchable mask source: y = a (x m + alpha' (1 - m)) + b"""
    if not training or p == 0.0:
        return x
    if _host_rng is None:
        return torch.nn.functional.alpha_dropout(x, p, True)
    m = keep_mask(x.shape, 1.0 - p, x.device)
    a, b = alpha_dropout_coeffs(p)
    return a * (x * m + _ALPHA_PRIME * (1.0 - m)) + b
