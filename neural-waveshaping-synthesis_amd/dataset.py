"""Control-feature dataset reader for batched offline rendering.

Directory layout written by the reference's preprocessing (data/utils/create_dataset.py, read by
data/general.py:9-57):  <root>/data_mean.npy, <root>/data_std.npy  (C,1) statistics, and per split
<root>/<split>/control/control_<name>.npy  (C,T) NORMALISED controls, optionally
<root>/<split>/audio/audio_<name>.npy target audio.  As in the reference's GeneralDataset.__getitem__,
F0 in Hz is recovered by de-normalising row 0; the model gets F0 in Hz and the normalised controls.
"""
from __future__ import annotations

import os
from collections import defaultdict

import numpy as np


class ControlDataset:
    def __init__(self, root: str, split: str = "test"):
        self.root, self.split = root, split
        self.control_dir = os.path.join(root, split, "control")
        self.audio_dir = os.path.join(root, split, "audio")
        if not os.path.isdir(self.control_dir):
            raise FileNotFoundError(f"no control features under {self.control_dir}")
        self.names = sorted(f[len("control_"):-4] for f in os.listdir(self.control_dir)
                            if f.startswith("control_") and f.endswith(".npy"))
        self.mean = np.load(os.path.join(root, "data_mean.npy")).astype(np.float64)
        self.std = np.load(os.path.join(root, "data_std.npy")).astype(np.float64)

    def __len__(self):
        return len(self.names)

    def item(self, name: str):
        control = np.load(os.path.join(self.control_dir, f"control_{name}.npy"))
        denorm = control.astype(np.float64) * self.std + self.mean      # general.py:49
        out = {"name": name, "control": control.astype(np.float32), "f0": denorm[0:1].astype(np.float32)}
        ap = os.path.join(self.audio_dir, f"audio_{name}.npy")
        if os.path.exists(ap):
            out["audio"] = np.load(ap).astype(np.float32)
        return out

    def shard(self, rank: int, world: int):
        """Round-robin split of the (sorted) item list: independent objects, no collective needed."""
        return self.names[rank::world]

    def batches(self, names, batch_size: int):
        """Group items of equal frame count (forward needs a rectangular batch), keep file order inside a group."""
        by_len = defaultdict(list)
        for n in names:
            by_len[np.load(os.path.join(self.control_dir, f"control_{n}.npy"), mmap_mode="r").shape[-1]].append(n)
        for T in sorted(by_len):
            group = by_len[T]
            for i in range(0, len(group), batch_size):
                items = [self.item(n) for n in group[i:i + batch_size]]
                yield {"names": [it["name"] for it in items],
                       "f0": np.stack([it["f0"] for it in items]),
                       "control": np.stack([it["control"] for it in items]),
                       "audio": [it.get("audio") for it in items]}
