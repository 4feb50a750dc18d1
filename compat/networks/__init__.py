"""Shim package: `networks` as the reference's scripts import it, backed by virnet_amd.networks (see compat/README.md)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)          # make `virnet_amd` importable from a reference checkout
