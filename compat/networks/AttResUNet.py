"""`networks.AttResUNet` of the reference (RNet) -> the MI355X parameter holders."""
from virnet_amd.networks.AttResUNet import AttLayer, AttResBlock, AttResUNet, DownBlock, UpBlock  # noqa: F401
