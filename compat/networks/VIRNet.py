"""`from networks.VIRNet import VIRAttResUNet, VIRAttResUNetSR` (scripts/testing_demo.py:23,37,51) -> the MI355X modules."""
from virnet_amd.networks.VIRNet import VIRAttResUNet, VIRAttResUNetSR, log_max, log_min  # noqa: F401
