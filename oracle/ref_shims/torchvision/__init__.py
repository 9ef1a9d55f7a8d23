"""stub of torchvision (resnet18 + ToTensor only); see ../README.md"""
from . import models, transforms  # noqa: F401
