"""Sensor positions for SpatialAttention.

The reference looks the 2-D layout up through MNE (speech_decoding/utils/layout.py:9-35), which needs
MNE plus the raw dataset and is outside this build's scope.  Here positions come,
in this order, from `args.sensor_positions` (a (C, 2) array / tensor / .npy path), from MNE when it is
importable and the dataset is on disk, or from a seeded synthetic table.  The normalisation of
layout.py:38-41 (min-max per axis, then *0.8 + 0.1) is always applied."""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

DEFAULT_CHANNELS = {"Gwilliams2022": 208, "Brennan2018": 60}


def normalise(raw_xy) -> torch.Tensor:
    xy = np.asarray(raw_xy, dtype=np.float64)
    span = xy.max(axis=0) - xy.min(axis=0)
    out = (xy - xy.min(axis=0)) / span
    return torch.from_numpy((out * 0.8 + 0.1).astype(np.float32))


def synthetic_positions(num_channels: int, seed: int = 0) -> torch.Tensor:
    """Seeded stand-in sensor table (uniform in the unit square, then the reference's normalisation) for
    benchmarks and runs without MNE / the dataset on disk."""
    return normalise(np.random.RandomState(seed).rand(num_channels, 2))


def _opt(args, key, default=None):
    if isinstance(args, dict):
        return args.get(key, default)
    return getattr(args, key, default)


def _from_mne(args):
    import mne  # noqa: F401  (optional dependency; absent in the build image)
    if args.dataset == "Brennan2018":
        montage = mne.channels.make_standard_montage("easycap-M10")
        info = mne.create_info(ch_names=montage.ch_names, sfreq=512.0, ch_types="eeg")
        info.set_montage(montage)
        pos = mne.channels.find_layout(info, ch_type="eeg").pos[:, :2]
        return np.delete(pos, 28, axis=0)          # channel 29 is broken in this dataset
    import mne_bids
    path = mne_bids.BIDSPath(subject="01", session="0", task="0", datatype="meg",
                             root=os.path.join(str(args.root_dir), "data", "Gwilliams2022"))
    raw = mne_bids.read_raw_bids(path)
    return mne.channels.find_layout(raw.info, ch_type="meg").pos[:, :2]


def ch_locations_2d(args) -> torch.Tensor:
    given = _opt(args, "sensor_positions")
    if given is not None:
        if isinstance(given, (str, os.PathLike)):
            given = np.load(given)
        if torch.is_tensor(given):
            given = given.detach().cpu().numpy()
        return normalise(given)
    dataset = args.dataset
    if dataset not in DEFAULT_CHANNELS:
        raise ValueError(f"unknown dataset {dataset!r}")
    try:
        return normalise(_from_mne(args))
    except Exception:
        n = int(_opt(args, "num_channels", DEFAULT_CHANNELS[dataset]))
        warnings.warn(f"MNE sensor layout unavailable; using a seeded synthetic {n}-sensor layout "
                      "(pass args.sensor_positions for real geometry)")
        return synthetic_positions(n, 0)
