"""Config plumbing with the reference's entry points (ldm/util.py:78-93 in zhanwenchen/pbe):
``instantiate_from_config`` / ``get_obj_from_str`` / ``count_params``.  Configs may be plain
dicts (PyYAML) or OmegaConf objects — only mapping access is used."""
import importlib


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def _plain(x):
    """OmegaConf / dict / list -> builtin containers (the modules keep no config objects)."""
    if hasattr(x, "items"):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)) or type(x).__name__ == "ListConfig":
        return [_plain(v) for v in x]
    return x


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", None) or {}
    return get_obj_from_str(config["target"])(**_plain(params))


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


def load_yaml_config(path):
    """configs/v1.yaml loader for callers without OmegaConf (PyYAML safe loader)."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)
