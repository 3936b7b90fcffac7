from . import resnet  # noqa
