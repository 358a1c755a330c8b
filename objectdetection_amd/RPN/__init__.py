from .anchor_generator import AnchorGenerator  # noqa: F401
from .rpn_head import RPNHead  # noqa: F401
from .rpn import RPN  # noqa: F401
