"""Reference-side glue (`neurosis.trainer`): only the Lightning adapter for the hot path is built here; the CLI, callbacks and
loggers of the reference are its control plane and stay the reference's (out of scope)."""
from .lightning import DiffusionEngineMI355X, lightning_available

__all__ = ["DiffusionEngineMI355X", "lightning_available"]
