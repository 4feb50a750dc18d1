"""`networks.KNet` of the reference -> the MI355X parameter holders."""
from virnet_amd.networks.KNet import CALayer, KernelNet, RB_Layer  # noqa: F401
