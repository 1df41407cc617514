"""Synthesise a checkpoint FILE in the layout of the published Paint-by-Example ``model.ckpt`` (a pytorch-lightning 1.4
checkpoint, scripts/inference.py:58-75): ``state_dict`` under the reference's key names (Hugging Face 4.19 CLIP spelling
``cond_stage_model.transformer.vision_model.*``), EMA shadow weights under ``model_ema.*`` (LitEma strips the dots,
ldm/modules/ema.py:20-24), and ``callbacks`` / ``hyper_parameters`` objects whose classes live in packages that are
NOT installed where the file is read (pytorch_lightning, omegaconf) — what makes ``torch.load(weights_only=True)``
refuse the real file."""
import collections
import sys
import types

import torch


def _foreign_class(module: str, name: str):
    """A picklable class that appears to live in `module` (registered only while the file is written)."""
    parts = module.split(".")
    for i in range(1, len(parts) + 1):
        m = ".".join(parts[:i])
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    cls = type(name, (), {"__module__": module, "__init__": lambda self, **kw: self.__dict__.update(kw)})
    setattr(sys.modules[module], name, cls)
    return cls


def write_lightning_checkpoint(path: str, state_dict, *, ema_from_prefix: str = "model.", hf5_clip_names: bool = False, foreign: bool = True):
    """state_dict: {reference key: tensor}.  Adds model_ema.* copies of the `ema_from_prefix` weights (perturbed, so a loader
    that wrongly used them would change the outputs), optimizer/callback/hyper-parameter entries, and saves with torch.save."""
    sd = collections.OrderedDict()
    for k, v in state_dict.items():
        if hf5_clip_names:                          # transformers >= 5 spelling: no `vision_model.` level
            k = k.replace("cond_stage_model.transformer.vision_model.", "cond_stage_model.transformer.")
        sd[k] = v.clone()
    for k, v in state_dict.items():
        if k.startswith(ema_from_prefix) and v.dtype.is_floating_point:
            sd["model_ema." + k[len(ema_from_prefix):].replace(".", "")] = v * 1.5 + 0.25
    sd["model_ema.decay"] = torch.tensor(0.9999)
    sd["model_ema.num_updates"] = torch.tensor(12345, dtype=torch.int32)
    ckpt = {"epoch": 39, "global_step": 283999, "pytorch-lightning_version": "1.4.2", "state_dict": sd,
            "optimizer_states": [], "lr_schedulers": []}
    made = []
    if foreign:
        MC = _foreign_class("pytorch_lightning.callbacks.model_checkpoint", "ModelCheckpoint")
        DC = _foreign_class("omegaconf.dictconfig", "DictConfig")
        made = ["pytorch_lightning.callbacks.model_checkpoint", "pytorch_lightning.callbacks", "pytorch_lightning", "omegaconf.dictconfig", "omegaconf"]
        ckpt["callbacks"] = {MC: {"monitor": "val/loss_simple_ema", "best_model_score": torch.tensor(0.123), "best_model_path": "/x/epoch=39.ckpt"}}
        ckpt["hyper_parameters"] = DC(content={"model": {"base_learning_rate": 1e-5}}, flags={"readonly": False})
    try:
        torch.save(ckpt, path)
    finally:
        for m in made:                               # the reader must not be able to resolve these
            sys.modules.pop(m, None)
    return sorted(sd)
