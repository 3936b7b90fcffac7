from .models.resnet import create_model  # noqa
from . import models  # noqa
