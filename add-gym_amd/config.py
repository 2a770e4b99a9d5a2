"""Small Hydra-compatible config composition (Hydra/OmegaConf are not installed offline): a root YAML with a
`defaults` list of `/group@key: name` entries, `_self_`, and `a.b.c=value` command-line overrides."""
import copy
import os

import yaml

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(PKG_DIR, "configs")


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _yaml(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def load_config(config_name="train", overrides=(), config_dir=CONFIG_DIR):
    root = _yaml(os.path.join(config_dir, config_name + ".yaml"))
    defaults = root.pop("defaults", [])
    groups = {}
    for ov in overrides:  # group selection overrides: `engine=kinematic`
        k, _, v = ov.partition("=")
        if "." not in k and os.path.isdir(os.path.join(config_dir, k)):
            groups[k] = v
    cfg = {}
    for entry in defaults:
        if entry == "_self_":
            _merge(cfg, root)
            continue
        (spec, name), = entry.items()
        group, _, key = spec.lstrip("/").partition("@")
        key = key or group
        name = groups.get(key, name)
        cfg[key] = _merge(cfg.get(key, {}), _yaml(os.path.join(config_dir, group, name + ".yaml")))
    if "_self_" not in defaults:
        _merge(cfg, root)
    for ov in overrides:
        k, _, v = ov.partition("=")
        if k in groups:
            continue
        node = cfg
        parts = k.lstrip("+").split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(v)
    return resolve_paths(cfg)


def resolve_paths(cfg):
    p = cfg.get("robot", {}).get("urdf_path")
    if p and not os.path.isabs(p) and not os.path.exists(p):
        cfg["robot"]["urdf_path"] = os.path.join(PKG_DIR, p)
    return cfg
