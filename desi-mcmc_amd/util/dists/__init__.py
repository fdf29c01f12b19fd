"""Mixture-of-Gaussians helpers with the names of CelestePy/util/dists (mog.py)."""
