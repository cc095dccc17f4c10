from .ddp import ArenaReducer  # noqa
from .launch import launch, default_argument_parser  # noqa
