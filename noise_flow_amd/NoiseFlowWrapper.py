"""``NoiseFlowWrapper`` — drop-in for ``borealisflows/NoiseFlowWrapper.py``.

Same constructor and ``sample_noise_nf(batch_x, b1, b2, iso, cam)`` signature
(reference ``NoiseFlowWrapper.py:20-44, 81-87``); called the same way from
``sample_noise_flow.py:40,71`` and ``train_dncnn_noiseflow.py:262,142,160``.

Two reference quirks are explicit options here:

* Q1 ``binding``: the reference wrapper builds ONLY the sampling graph, which by
  TF-1.12 ``make_template`` semantics attaches the trained coupling-CNN weights
  to the layers in reversed order (``'sample_first'``).  The default here is
  ``'loss_first'`` — the binding the model was trained with, and the one that reproduces the camera: on the shipped
  checkpoint its samples carry 1.1 - 1.3 x the S6 NLF's standard deviation at temperature 1 (the "too-high noise variance"
  ``sample_noise_flow.py:36-39`` tempers with 0.6), the reversed binding 3 - 47 x (pinned by
  ``tests/test_gpu_scripts.py::test_wrapper_default_is_the_mode_that_reproduces_the_camera_noise``).
* Q2 ``is_training=True`` in the reference's feed (``NoiseFlowWrapper.py:86``)
  switches BN to batch statistics.  The default here, ``bn_mode='running'``,
  uses the stored running statistics (the evaluation mode of
  ``train_noise_flow.py:167-168``); ``bn_mode='batch'`` reproduces the
  reference wrapper literally (moments of the call's own patches).
"""
from __future__ import annotations

import logging
import os

import numpy as np

from .hps import hps_loader
from .noise_flow_model import NoiseFlow


class NoiseFlowWrapper:
    def __init__(self, path, sampling_temperature=0.6, binding="loss_first", device=None, seed=None,
                 bn_mode="running", compat=None, patch_shape=None):
        if compat not in (None, "reference"):
            raise ValueError("compat must be None or 'reference'")
        if compat == "reference":
            # what borealisflows/NoiseFlowWrapper.py literally builds: only the sampling graph, so the CNN templates
            # bind in sampling order (:64, quirk Q1), and is_training=True at every call (:86, quirk Q2)
            binding, bn_mode = "sample_first", "batch"
        self.compat = compat
        if bn_mode not in ("running", "batch"):
            raise ValueError("bn_mode must be 'running' or 'batch'")
        self.bn_mode = bn_mode
        self.logger = logging.getLogger(__name__)
        self.nf_path = path
        self.nf_model = None
        self.is_cond = True
        self.temp = sampling_temperature
        # the reference fixes 32x32 (NoiseFlowWrapper.py:47); any (H, W) up to 4096 per side here — beyond 64x64 the patches are
        # evaluated as overlapping tiles, same numbers as a kernel that held them whole
        self.patch_shape = (32, 32) if patch_shape is None else (int(patch_shape[0]), int(patch_shape[1]))
        self.binding = binding
        self.device = device
        self.hps = self.hps_loader(os.path.join(self.nf_path, "hps.txt"))
        if seed is not None:
            self.hps.seed = seed
        self.ckpt_dir = os.path.join(self.nf_path, "ckpt")
        self.model_checkpoint_path = os.path.join(self.ckpt_dir, "model.ckpt.best")
        self.load_noise_flow_model()

    def load_noise_flow_model(self):
        self.x_shape = [None, self.patch_shape[0], self.patch_shape[1], 4]   # NoiseFlowWrapper.py:47 fixes 32x32 (quirk Q8)
        if not hasattr(self.hps, "x_shape") or isinstance(self.hps.x_shape, str):
            setattr(self.hps, "x_shape", self.x_shape)
        self.logger.info("Building Noise Flow")
        self.nf_model = NoiseFlow(self.x_shape[1:], self.bn_mode == "batch", self.hps, binding=self.binding, device=self.device)
        self.logger.info("Restoring best model")
        self.nf_model.restore(self.model_checkpoint_path)

    def sample_noise_nf(self, batch_x, b1, b2, iso, cam):
        """NoiseFlowWrapper.py:81-87 → float32 [B,32,32,4] noise (unclipped)."""
        return self.sample_sidd_tf(batch_x, b1, b2, iso, cam)

    def sample_sidd_tf(self, batch_x, b1=0.0, b2=0.0, iso=100.0, cam=2.0):
        if self.is_cond:
            x = self.nf_model.sample(batch_x, self.temp, batch_x, [b1], [b2], [iso], [cam])
        else:
            x = self.nf_model.sample(batch_x, self.temp)
        return x.astype(np.float32) if isinstance(x, np.ndarray) else x

    @staticmethod
    def hps_loader(path):
        return hps_loader(path)
