"""agcn_amd -- MI355X-native (gfx950) implementation of the 2s-AGCN hot path.

Loaded under the name ``agcn_amd`` (see ``/agcn_amd.py`` at the repo root).
"""
__version__ = "0.1.0"
