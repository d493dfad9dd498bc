"""MI355X-native engine for the ResNet forward/backward hot path of lucaslingle/pytorch_ddp_resnet."""
from .architectures.resnet import ResNet  # noqa: F401
