"""MI355X-native implementation of the MVNeRF volumetric-rendering hot path of
TWeber132/thesis-clip-nerf (src/lib/mvnerf): hand-written HIP kernels behind a C ABI
(include/mvnerf_hip.h, thesis_clip_nerf_amd/csrc), with a Python surface that mirrors the
reference's `MVVNeRFRenderer` / `nerf_utils` names.  See DESIGN.md."""
from . import _lib, model, nerf_utils, ops, synthetic  # noqa: F401
from .model import MVVNeRFRenderer, render, render_view  # noqa: F401

__all__ = ['_lib', 'ops', 'synthetic', 'model', 'nerf_utils', 'MVVNeRFRenderer', 'render', 'render_view']
