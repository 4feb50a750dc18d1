"""`networks.DnCNN` of the reference (SNet) -> the MI355X parameter holder."""
from virnet_amd.networks.DnCNN import DnCNN  # noqa: F401
