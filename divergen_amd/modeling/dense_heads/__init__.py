from .centernet import CenterNet  # noqa
from .centernet_head import CenterNetHead  # noqa
