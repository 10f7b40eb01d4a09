"""Hydra-free composer for a `configurations/` tree laid out as the reference's (configurations/config.yaml:1-12):

    config.yaml            defaults: [{experiment: matrix_flow}, {dataset: sintel}, {algorithm: pwc_learner}] + own keys (wandb: ...)
    experiment/base.yaml   experiment/matrix_flow.yaml  (defaults: [base])   algorithm/flow_diffuser.yaml   dataset/sintel.yaml

`compose(dir, overrides)` applies the subset of Hydra's override grammar main.py is launched with: `group=option` picks a file of a
group, `a.b.c=value` sets a (YAML-typed) leaf, `+a.b=value` adds one, `~a.b` deletes one.  The result is a `Config`: a dict with
attribute access that also answers the probes the reference makes on a DictConfig (`'clipping' in dir(cfg.experiment.training)`,
exp_base.py:191; `{**cfg}`, main.py:20; `.get`)."""
import os
import re

import yaml


class Config(dict):
    """nested dict with attribute access (the part of omegaconf.DictConfig the FlowDiffuser path touches)"""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, list):
            return [Config._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def __dir__(self):
        return list(self.keys()) + list(super().__dir__())

    def to_container(self):
        return {k: (v.to_container() if isinstance(v, Config) else v) for k, v in self.items()}


_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")


def _typed(v):
    """OmegaConf reads `lr: 1e-5` (flow_diffuser.yaml:9) as a float; YAML 1.1 (PyYAML) as a string"""
    if isinstance(v, str) and _FLOAT.match(v):
        return float(v)
    if isinstance(v, dict):
        return {k: _typed(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_typed(x) for x in v]
    return v


def _load(path):
    with open(path) as f:
        return _typed(yaml.safe_load(f) or {})


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_group(config_dir, group, option):
    """a group file with its own (same-group) defaults list, e.g. experiment/matrix_flow.yaml: `defaults: [base]`"""
    path = os.path.join(config_dir, group, f"{option}.yaml")
    if not os.path.exists(path):
        have = sorted(f[:-5] for f in os.listdir(os.path.join(config_dir, group)) if f.endswith(".yaml"))
        raise FileNotFoundError(f"no option '{option}' in config group '{group}' (have: {', '.join(have)})")
    d = _load(path)
    out = {}
    for parent in d.pop("defaults", []) or []:
        if parent == "_self_":
            continue
        _merge(out, _load_group(config_dir, group, parent if isinstance(parent, str) else list(parent.values())[0]))
    return _merge(out, d)


def compose(config_dir, overrides=(), config_name="config"):
    root = _load(os.path.join(config_dir, f"{config_name}.yaml"))
    choices = {}
    for entry in root.pop("defaults", []) or []:
        if isinstance(entry, dict):
            choices.update({str(k): v for k, v in entry.items()})
    leaf = []
    for ov in overrides:
        key, eq, val = ov.partition("=")
        if eq and "." not in key and key.lstrip("+") in choices | {g: None for g in os.listdir(config_dir) if os.path.isdir(os.path.join(config_dir, g))}:
            choices[key.lstrip("+")] = val                                  # group=option
        else:
            leaf.append(ov)
    cfg = {}
    for group, option in choices.items():
        if option is not None:
            cfg[group] = _load_group(config_dir, group, str(option))
    _merge(cfg, root)
    for ov in leaf:
        if ov.startswith("~"):
            d, keys = cfg, ov[1:].split(".")
            for k in keys[:-1]:
                d = d.get(k, {})
            d.pop(keys[-1], None)
            continue
        key, eq, val = ov.partition("=")
        if not eq:
            raise ValueError(f"override '{ov}': expected key=value, group=option or ~key")
        d, keys = cfg, key.lstrip("+").split(".")
        for k in keys[:-1]:
            d = d.setdefault(k, {})
        d[keys[-1]] = _typed(yaml.safe_load(val))
    return Config(cfg)
