"""interflow_amd -- MI355X (gfx950) implementation of InterFlow's conditional-flow forward/inverse path.

`InterpFlow` is a drop-in for the reference's `models.flow_model.InterpFlow`; the arithmetic of the
hot path runs in `libinterflow_hip.so` (C ABI: include/interflow_hip.h).
"""
from .flow_model import InterpFlow, RRDBGlowNet, RRDBNet, Gaussian  # noqa: F401
from .flow_modules import FlowUpsamplerNet, FlowStep, SqueezeLayer, Split2d, opt_get  # noqa: F401
from . import hip, pack, synth  # noqa: F401

__all__ = ["InterpFlow", "RRDBGlowNet", "RRDBNet", "Gaussian", "FlowUpsamplerNet", "FlowStep", "SqueezeLayer",
           "Split2d", "opt_get", "hip", "pack", "synth"]
