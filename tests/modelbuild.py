"""Test helpers: build the narrow / full LatentDiffusion of this repo with name-seeded weights."""
import torch

import cases
from ldm.util import instantiate_from_config, load_yaml_config
from pbe_amd.weights import fill_latent_diffusion_


def _unet_cfg(c):
    return {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
            "params": dict(image_size=32, in_channels=c["in_channels"], out_channels=c["out_channels"], model_channels=c["model_channels"],
                           attention_resolutions=list(c["attention_resolutions"]), num_res_blocks=c["num_res_blocks"],
                           channel_mult=list(c["channel_mult"]), num_heads=c["num_heads"], use_spatial_transformer=True, transformer_depth=1,
                           context_dim=c["context_dim"], use_checkpoint=True, legacy=False)}


def _vae_cfg(c):
    return {"target": "ldm.models.autoencoder.AutoencoderKL",
            "params": dict(embed_dim=c["embed_dim"], ddconfig=dict(double_z=True, z_channels=c["z_channels"], resolution=256, in_channels=c["in_channels"],
                                                                  out_ch=c["out_ch"], ch=c["ch"], ch_mult=list(c["ch_mult"]), num_res_blocks=c["num_res_blocks"],
                                                                  attn_resolutions=[], dropout=0.0), lossconfig={"target": "torch.nn.Identity"})}


def _clip_cfg(c, m):
    return {"target": "ldm.modules.encoders.modules.FrozenCLIPImageEmbedder",
            "params": dict(clip_config=dict(hidden_size=c["hidden"], intermediate_size=c["mlp"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                                            image_size=c["image"], patch_size=c["patch"], layer_norm_eps=c["eps"]), mapper_layers=m["layers"])}


def narrow_config():
    return {"target": "ldm.models.diffusion.ddpm.LatentDiffusion",
           "params": dict(linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000, first_stage_key="inpaint",
                          cond_stage_key="image", image_size=16, channels=4, cond_stage_trainable=True, conditioning_key="crossattn",
                          scale_factor=0.18215, use_ema=False, unet_config=_unet_cfg(cases.UNET_NARROW), first_stage_config=_vae_cfg(cases.VAE_NARROW),
                          cond_stage_config=_clip_cfg(cases.CLIP_NARROW, cases.MAPPER_NARROW), cond_embed_dim=cases.MAPPER_NARROW["width"])}


def narrow_model(device):
    model = instantiate_from_config(narrow_config())
    fill_latent_diffusion_(model)
    return model.to(device).eval()


def full_model(device, parts=("unet", "vae", "clip")):
    """configs/v1.yaml model with name-seeded weights (only the requested parts are filled)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = instantiate_from_config(load_yaml_config(os.path.join(root, "configs", "v1.yaml"))["model"])
    fill_latent_diffusion_(model)
    return model.to(device).eval()
