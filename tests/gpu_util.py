"""Helpers shared by the -m gpu parity tests."""
import json
import os

import numpy as np
import torch

from conftest import ROOT

REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.json")


def record(name, **metrics):
    """Append measured errors to gpurun_out/parity_report.json so a passing run still shows its margins."""
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        try:
            data = json.load(open(REPORT))
        except Exception:
            data = {}
    data[name] = {k: (float(v) if not isinstance(v, str) and (np.isscalar(v) or isinstance(v, (np.floating, float))) else v)
                  for k, v in metrics.items()}
    json.dump(data, open(REPORT, "w"), indent=1, sort_keys=True)


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype).cuda().contiguous()


def build_model(fast):
    import nws_amd as nws

    nws.ensure_default_config()
    m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", "weights_vn.npz")).cuda().eval()
    if fast:
        m.newt = nws.FastNEWT(m.newt)
    return m


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
