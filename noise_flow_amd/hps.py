"""``hps.txt`` reader/writer compatible with the reference.

``hps_loader`` follows ``NoiseFlowWrapper.hps_loader``
(reference ``borealisflows/NoiseFlowWrapper.py:96-138``): CSV ``key,value`` rows,
rows with fewer than two fields (layer names, parameter count) skipped, values
coerced int → float → bool → str, and ``param_inits`` REBUILT from constants, not
parsed.  ``hps_logger`` follows ``borealisflows/utils.py:110-119``.
"""
from __future__ import annotations

import csv
from types import SimpleNamespace

import numpy as np


class Hps(SimpleNamespace):
    pass


def hps_loader(path: str) -> Hps:
    hps = Hps()
    with open(path, "r") as f:
        for pair in csv.reader(f):
            if len(pair) < 2:
                continue
            val = pair[1]
            try:
                val = int(val)
            except ValueError:
                try:
                    val = float(val)
                except ValueError:
                    if val == "True":
                        val = True
                    elif val == "False":
                        val = False
            setattr(hps, pair[0], val)
    # NoiseFlowWrapper.py:121-137 — npcam is left undefined by the reference when
    # the arch has neither sdn5 nor sdn6 (quirk Q8); default to 3 here.
    arch = str(getattr(hps, "arch", ""))
    npcam = 1 if ("sdn6" in arch and "sdn5" not in arch) else 3
    c_i = 1.0
    gain_params_i = np.full([5], -5.0 / c_i)
    cam_params_i = np.ones([npcam, 5])
    hps.param_inits = (c_i, -5.0 / c_i, 0.0, gain_params_i, cam_params_i)
    return hps


def hps_loader_raw(path: str) -> Hps:
    """``borealisflows/utils.py:122-135``: every value kept as a string."""
    hps = Hps()
    with open(path, "r") as f:
        for pair in csv.reader(f):
            if len(pair) < 2:
                continue
            setattr(hps, pair[0], pair[1])
    return hps


def hps_logger(path: str, hps, layer_names, num_params) -> None:
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        for n in layer_names:
            w.writerow([n])
        w.writerow([num_params])
        for k, v in vars(hps).items():
            w.writerow([k, v])
