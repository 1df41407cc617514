"""Diagonal Gaussian posterior of the KL autoencoder — API of
ldm/modules/distributions/distributions.py:24-37,61-62 in zhanwenchen/pbe.

``parameters`` are the moments as the HIP encoder produced them: NHWC fp16 ``[B, h, w, 2*z]``
(mean | logvar).  ``sample()`` draws its N(0,1) noise from the CPU generator exactly like the
reference (``torch.randn(shape)`` then moved to the device, distributions.py:36) unless ``noise``
is injected, and evaluates mean + exp(0.5 clamp(logvar, -30, 20)) * noise in one kernel."""
import torch

from pbe_amd import ops


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters                  # [B, h, w, 2z] fp16 NHWC
        self.deterministic = deterministic
        self.z = parameters.shape[-1] // 2

    @property
    def mean(self):
        return ops.nhwc_to_nchw(self.parameters, self.z)

    def sample(self, noise=None, scale=1.0):
        B, h, w, _ = self.parameters.shape
        if self.deterministic:
            return scale * self.mean
        if noise is None:
            noise = torch.randn((B, self.z, h, w))
        noise = noise.to(device=self.parameters.device, dtype=torch.float32)
        return ops.posterior_sample(self.parameters, noise, scale)

    def mode(self):
        return self.mean
