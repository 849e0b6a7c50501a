from .dice import MDiceLoss, MDiceLoss_Val  # noqa: F401
