from .sigmoid_focal_loss import SigmoidFocalLoss, sigmoid_focal_loss  # noqa: F401
