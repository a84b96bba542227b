from .node import ConfigNode
from .defaults import default_config, conf
from .cmd_parser import parse_args, merge_config
