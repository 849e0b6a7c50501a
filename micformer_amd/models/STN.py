"""SpatialTransformer of the reference's MicFormer/models/STN.py on the HIP sampler kernel.

forward(src (B,C,D,H,W), flow (B,3,D,H,W)): new = voxel index + flow, normalised as 2*(new/(S-1) - .5) and sampled by
grid_sample(trilinear, zeros, align_corners=False) -- i.e. at continuous index new*S/(S-1) - .5 (STN.py:9-32).  Inside
CrossTransformerBlock3D the sampler is fused with the offset head (micf_offset_sample_*); this module is the
standalone entry kept for API compatibility (3-D, mode='bilinear' only).
"""
import torch
import torch.nn as nn

from .. import ops


class _STNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, flow):
        B, C, D, H, W = src.shape
        s = src.permute(0, 2, 3, 4, 1).contiguous().reshape(-1, C)
        f = flow.permute(0, 2, 3, 4, 1).contiguous().reshape(-1, 3)
        out = ops.stn_fwd(s, f, (B, D, H, W))
        ctx.save_for_backward(s, f)
        ctx.shape = (B, C, D, H, W)
        return out.reshape(B, D, H, W, C).permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dy):
        s, f = ctx.saved_tensors
        B, C, D, H, W = ctx.shape
        dyl = dy.permute(0, 2, 3, 4, 1).contiguous().reshape(-1, C)
        ds, df = ops.stn_bwd(dyl, s, f, (B, D, H, W))
        return ds.reshape(B, D, H, W, C).permute(0, 4, 1, 2, 3), df.reshape(B, D, H, W, 3).permute(0, 4, 1, 2, 3)


class SpatialTransformer(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, src, flow, mode='bilinear'):
        if src.dim() != 5 or mode != 'bilinear':
            raise NotImplementedError("HIP SpatialTransformer: 3-D volumes, mode='bilinear' (trilinear)")
        return _STNFn.apply(src.float(), flow.float())


class Re_SpatialTransformer(nn.Module):
    """STN.py:35-43 (unused by MicFormer): warp with the negated, self-warped flow."""

    def __init__(self):
        super().__init__()
        self.stn = SpatialTransformer()

    def forward(self, src, flow, mode='bilinear'):
        flow = -1 * self.stn(flow, flow, mode='bilinear')
        return self.stn(src, flow, mode)
