"""Drop-in `loss` package (`from loss import MDiceLoss`, train_mmwhs_noPad.py:19; MicFormer/loss/__init__.py:1)."""
from loss.dice import MDiceLoss, MDiceLoss_Val  # noqa: F401  (the reference's own absolute form)
