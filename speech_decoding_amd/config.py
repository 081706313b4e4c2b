"""Hydra-optional loader for configs/config.yaml (same keys as the reference's config.yaml:1-53).

`load_config(path, overrides=["batch_size=128", "preprocs.clamp_lim=10"])` returns a `Config`, a dict
with attribute access on every level — the two access styles the hot path uses (`args.D1`,
`args.preprocs["last4layers"]`).  When Hydra/OmegaConf are installed the reference's `@hydra.main`
flow works unchanged; this loader covers environments without them."""
from __future__ import annotations

import os
from typing import Iterable, Optional

import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Config({k: Config.wrap(v) for k, v in obj.items()})
        if isinstance(obj, list):
            return [Config.wrap(v) for v in obj]
        return obj


def default_config_path() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "config.yaml")


def load_config(path: Optional[str] = None, overrides: Iterable[str] = ()) -> Config:
    with open(path or default_config_path()) as f:
        cfg = Config.wrap(yaml.safe_load(f))
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not key=value")
        key, val = ov.split("=", 1)
        node = cfg
        parts = key.lstrip("+").split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, Config())
        node[parts[-1]] = Config.wrap(yaml.safe_load(val))
    return cfg
