"""Import-name shim: ``import MultiScaleDeformableAttention as MSDA`` (reference
mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:22) resolves to the gfx950
implementation when the repo root (or an installed copy of this file) is on ``sys.path``.  Exports the two
functions of the reference's pybind module (ops/src/vision.cpp:18-21)."""
from dvis_plus_amd.functions import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
