"""noise_flow_amd — MI355X-native (gfx950) Noise Flow bijector stack.

Only the hot path of BorealisAI/noise_flow lives here: the bijector chain in the
likelihood direction (NLL + log|det J|) and in the sampling direction, as fused
HIP kernels behind a C ABI (``include/noiseflow_hip.h``), under the reference's
``NoiseFlow`` / ``NoiseFlowWrapper`` operator surface.
"""
from .noise_flow_model import NoiseFlow, default_hps  # noqa: F401
from .NoiseFlowWrapper import NoiseFlowWrapper  # noqa: F401
from .squeeze import squeeze2d, unsqueeze2d  # noqa: F401   (borealisflows/utils.py:30-86; host-side index maps)

__all__ = ["NoiseFlow", "NoiseFlowWrapper", "default_hps", "squeeze2d", "unsqueeze2d"]
