"""state_dict key/shape table of the reference ``Head`` -- TEST INFRASTRUCTURE ONLY.

Restated from MS.py constructors (MS.py:161-177, 295-324, 448-470, 535-540, 565-569, 600-687, 903-990, 1042-1046)
and pinned against the real reference by tests/golden/state_dict_*.json.
"""
from collections import OrderedDict

import torch

from .fill import fill_tensor
from .micformer_ref import Cfg


def _block(sd, pre, C, cross, hidden=16):
    sd[pre + "norm1.weight"] = (C,)
    sd[pre + "norm1.bias"] = (C,)
    a = pre + ("cross_attn." if cross else "self_attn.")
    sd[a + "q.weight"] = (C, C)
    sd[a + "q.bias"] = (C,)
    sd[a + "kv.weight"] = (2 * C, C)
    sd[a + "kv.bias"] = (2 * C,)
    sd[a + "proj.weight"] = (C, C)
    sd[a + "proj.bias"] = (C,)
    if cross:
        sd[pre + "conv_offset.0.weight"] = (hidden, 2 * C, 3, 3, 3)
        sd[pre + "conv_offset.0.bias"] = (hidden,)
        sd[pre + "conv_offset.1.norm.weight"] = (hidden,)
        sd[pre + "conv_offset.1.norm.bias"] = (hidden,)
        sd[pre + "conv_offset.3.weight"] = (3, hidden, 1, 1, 1)
    sd[pre + "norm2.weight"] = (C,)
    sd[pre + "norm2.bias"] = (C,)
    sd[pre + "mlp.fc1.weight"] = (4 * C, C)
    sd[pre + "mlp.fc1.bias"] = (4 * C,)
    sd[pre + "mlp.fc2.weight"] = (C, 4 * C)
    sd[pre + "mlp.fc2.bias"] = (C,)


def _layer(sd, pre, C, depth):
    # registration order in BasicLayer.__init__: blocks1, blocks2, self_blocks1, self_blocks2, downsample
    for name, cross in (("blocks1", True), ("blocks2", True), ("self_blocks1", False), ("self_blocks2", False)):
        for i in range(depth):
            _block(sd, f"{pre}{name}.{i}.", C, cross)


def state_dict_shapes(cfg: Cfg) -> "OrderedDict[str, tuple]":
    E, nl = cfg.embed_dim, len(cfg.depths)
    p = cfg.patch_size
    sd = OrderedDict()
    sd["swin.patch_embed.proj.weight"] = (E, 1, p, p, p)
    sd["swin.patch_embed.proj.bias"] = (E,)
    for s in range(nl):
        C = E * 2 ** s
        _layer(sd, f"swin.layers.{s}.", C, cfg.depths[s])
        if s < nl - 1:
            sd[f"swin.layers.{s}.downsample.down_conv.weight"] = (2 * C, C, 2, 2, 2)
            sd[f"swin.layers.{s}.downsample.down_conv.bias"] = (2 * C,)
            sd[f"swin.layers.{s}.downsample.norm.weight"] = (2 * C,)
            sd[f"swin.layers.{s}.downsample.norm.bias"] = (2 * C,)
    for inx in range(nl):
        s = nl - 1 - inx
        C = E * 2 ** s
        _layer(sd, f"swin.up_layers.{inx}.", C, cfg.depths[s])
        if s > 0:
            sd[f"swin.up_layers.{inx}.downsample.up_conv.weight"] = (C, C // 2, 2, 2, 2)
            sd[f"swin.up_layers.{inx}.downsample.up_conv.bias"] = (C // 2,)
            sd[f"swin.up_layers.{inx}.downsample.norm.weight"] = (C // 2,)
            sd[f"swin.up_layers.{inx}.downsample.norm.bias"] = (C // 2,)
    for inx in range(nl):
        C = E * 2 ** (nl - 1 - inx)
        sd[f"swin.concat_back_dim.{inx}.weight"] = (C, 2 * C)
        sd[f"swin.concat_back_dim.{inx}.bias"] = (C,)
    sd["swin.norm.weight"] = (E * 2 ** (nl - 1),)
    sd["swin.norm.bias"] = (E * 2 ** (nl - 1),)
    sd["swin.norm2.weight"] = (2 * E,)
    sd["swin.norm2.bias"] = (2 * E,)
    sd["swin.reverse_patch_embedding.weight"] = (2 * E, E // 2, p, p, p)
    sd["swin.reverse_patch_embedding.bias"] = (E // 2,)
    sd["out_conv.weight"] = (cfg.num_classes, E // 2, 3, 3, 3)
    sd["out_conv.bias"] = (cfg.num_classes,)
    return sd


def filled_params(cfg: Cfg, dtype=torch.float32):
    """Closed-form parameters for `cfg`, identical to fill_state_dict() on the reference module."""
    return {k: fill_tensor(k, torch.empty(shp)).to(dtype) for k, shp in state_dict_shapes(cfg).items()}
