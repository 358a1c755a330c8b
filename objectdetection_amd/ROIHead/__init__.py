from .roi_pooling import ROIPooling  # noqa: F401
from .mlp_head import TwoMLPHead  # noqa: F401
from .predictor import FasterRCNNPredictor  # noqa: F401
from .roi_head import ROIHead  # noqa: F401
