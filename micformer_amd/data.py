"""Device-side tail of the reference's input pipeline.

The reference feeds `Head` from a MONAI dict-transform chain (train_mmwhs_noPad.py:116-130) applied per sample on the CPU to the
loader's float16 image (2, D, H, W) and boolean one-hot label planes (MMWHS.py:386-405), then casts to float32 and moves both
to the GPU (train.py:177-181).  Here the loader side stops at the raw float16 volume and the uint8 CLASS MAP (8x fewer label
bytes over PCIe and in HBM; MDiceLoss expands it inside the loss kernel) and the rest runs on the device in two launches
(micf_intensity_stats + micf_input_prepare, csrc/misc.hip):

    RandFlipd(spatial_axis 0 / 1 / 2, prob 0.5) on image and label  ->  index arithmetic
    NormalizeIntensityd(nonzero=True, channel_wise=True)              ->  per-channel mean / std over the non-zero voxels
    RandScaleIntensityd(factors 0.1, prob 1), RandShiftIntensityd(offsets 0.1, prob 1)  ->  one affine map per sample
    .float()                                                          ->  the output is float32

MONAI is an un-pinned dependency that is not under /root/reference: the transforms are restated from their published behaviour
(oracle/micformer_ref.py::input_pipeline_tail is the referee; parity with MONAI itself is unpinned).
"""
import torch

from . import ops


def draw_augmentation(batch, generator=None, device="cpu", flip_prob=0.5, scale=0.1, shift=0.1):
    """[B, 5] float32 {flip D, flip H, flip W, scale factor f ~ U(-scale, scale), shift offset o ~ U(-shift, shift)}: the random
    draws of train.py:118-123 for a batch (MONAI draws them per sample from its own RandomState; the distribution is what is kept)."""
    u = torch.rand(batch, 5, generator=generator)
    p = torch.empty(batch, 5)
    p[:, :3] = (u[:, :3] < flip_prob).float()
    p[:, 3] = (2 * u[:, 3] - 1) * scale
    p[:, 4] = (2 * u[:, 4] - 1) * shift
    return p.to(device)


def prepare_batch(image, label_map=None, params=None):
    """image [B, 2, D, H, W] float16 / float32 raw intensities on the GPU, label_map [B, D, H, W] uint8 (or None), params from
    draw_augmentation (None = the validation transform: normalise only, train.py:126-130).  -> (x float32 for Head, label map)."""
    return ops.input_prepare(image.contiguous(), None if label_map is None else label_map.contiguous(), params)
