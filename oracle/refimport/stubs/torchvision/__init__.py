from . import transforms  # noqa
