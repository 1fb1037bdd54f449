from .iou_loss import IOULoss  # noqa: F401
from .sigmoid_focal_loss import SigmoidFocalLoss, sigmoid_focal_loss  # noqa: F401
