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


class RawBatch:
    """The raw image batch with its augmentation draws, for a Head whose patch embedding applies the input tail itself (flips as
    index arithmetic, normalise / scale / shift as one affine map per (sample, channel) inside the patch gather,
    micf_patch_rows_prepared): `model(RawBatch(image, params))` equals `model(prepare_batch(image, None, params)[0])` without the
    float32 volume ever being written.  Quacks like the (B, 2, D, H, W) tensor where Head / TrainEngine look at it."""

    def __init__(self, image, params=None, sums=None):
        if image.dim() != 5 or image.shape[1] != 2 or image.dtype not in (torch.float16, torch.float32) or not image.is_cuda:
            raise ValueError("RawBatch expects a (B, 2, D, H, W) float16 / float32 CUDA volume")
        self.image = image.contiguous()
        self.params = None if params is None else params.to(image.device, torch.float32).contiguous()
        self.sums = sums                                  # per-channel non-zero statistics (computed on first use)

    shape = property(lambda self: self.image.shape)
    dtype = property(lambda self: self.image.dtype)
    device = property(lambda self: self.image.device)
    is_cuda = True

    def dim(self):
        return 5

    def clone(self):
        return RawBatch(self.image.clone(), None if self.params is None else self.params.clone())

    def copy_(self, other, non_blocking=False):
        """In-place refresh of a captured step's static input (TrainEngine): the statistics are recomputed by the replay."""
        if not isinstance(other, RawBatch) or (self.params is None) != (other.params is None):
            # (silently dropping the draws would leave the image un-flipped under labels prepare_raw_batch already flipped)
            raise ValueError("RawBatch.copy_: source must be a RawBatch with the same augmentation state (params None / not None)")
        self.image.copy_(other.image, non_blocking=non_blocking)
        if self.params is not None:
            self.params.copy_(other.params, non_blocking=non_blocking)
        return self

    def patch_rows(self, k):
        """-> ([rows of modality 0, rows of modality 1], (B, D', H', W')): two launches (statistics, gather)."""
        B, _, D, H, W = self.image.shape
        sums = ops.intensity_stats(self.image)            # (inside a captured step: part of the graph, fresh per replay)
        return ops.patch_rows_prepared(self.image, sums, self.params, k), (B, -(-D // k), -(-H // k), -(-W // k))


def prepare_raw_batch(image, label_map=None, params=None):
    """The fused counterpart of prepare_batch: -> (RawBatch for Head, flipped uint8 label map)."""
    if params is not None:
        params = params.to(image.device, torch.float32).contiguous()      # (the C-ABI takes float32 draws only)
    lab = None if label_map is None else ops.flip_labels(label_map.contiguous(), params)
    return RawBatch(image, params), lab
