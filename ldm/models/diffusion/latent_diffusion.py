"""Import-path alias: the zhanwenchen/pbe fork defines LatentDiffusion in
ldm/models/diffusion/latent_diffusion.py:85 while configs/v1.yaml:3 (and upstream
Paint-by-Example) target ldm.models.diffusion.ddpm.LatentDiffusion — both resolve to one class."""
from ldm.models.diffusion.ddpm import DDPM, DiffusionWrapper, LatentDiffusion, disabled_train  # noqa: F401
