"""Samplers with the names of CelestePy/util/infer (only what the render path's callers use)."""
from .slicesample import slicesample, slicesample_lockstep  # noqa: F401
