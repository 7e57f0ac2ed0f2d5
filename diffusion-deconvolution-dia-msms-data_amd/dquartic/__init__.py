"""dquartic (D^4) on MI355X: drop-in host-side mirror of the reference package's hot path
(dquartic.model.model / dquartic.model.unet1d / dquartic.model.model_interface / dquartic.cli) on top of
libdq_hip.so (hand-written gfx950 kernels).  See DESIGN.md and include/dq_hip.h."""
__version__ = "0.1.0"
