from .VIRNet import VIRAttResUNet, VIRAttResUNetSR  # noqa: F401
