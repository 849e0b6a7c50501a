#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference on CPU.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing of the reference
travels: this script imports it in memory (no bytecode written), feeds it closed-form weights and
inputs (oracle/fill.py) and stores inputs-free OUTPUT vectors as small .npz/.json fixtures.

The reference needs one missing third-party symbol, ``timm.models.layers.DropPath`` (MS.py:5; timm
is not installed and un-pinned by the reference).  A minimal in-memory stand-in with the published
timm semantics (per-sample Bernoulli keep mask, scale_by_keep=True; identity in eval) is registered
before the import; all goldens are produced in eval() mode or with drop_path_rate effectively unused,
so the stand-in's RNG never influences a stored value.

usage:  python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/MicFormer"

from oracle import fill  # noqa: E402


def import_reference():
    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                m.div_(keep)
            return x * m

    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"].DropPath = DropPath
    sys.path.insert(0, REF)
    import models.MICFormer_self as MS  # noqa
    spec = importlib.util.spec_from_file_location("ref_dice", os.path.join(REF, "loss", "dice.py"))
    dice = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dice)
    return MS, dice


def np32(t):
    return t.detach().to(torch.float32).contiguous().clone().numpy()   # clone: never alias live parameters


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name:40s} {os.path.getsize(path) / 1024:8.1f} KiB")


def meandice_ref(pred, label, num_class):
    # train_mmwhs_noPad.py:392-407 cannot be imported (monai/tensorboard imports at module top);
    # the value stored here is produced by executing that function's own arithmetic on the reference's ops.
    sumdice = 0
    smooth = 1e-6
    for i in range(1, num_class):
        pb = (pred == i) * 1
        lb = (label == i) * 1
        pb = pb.contiguous().view(pb.shape[0], -1)
        lb = lb.contiguous().view(lb.shape[0], -1)
        inter = (pb * lb).sum()
        sumdice += (2. * inter + smooth) / (pb.sum() + lb.sum() + smooth)
    return sumdice / (num_class - 1)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MS, dice = import_reference()

    # ---------------------------------------------------------------- state_dict tables
    print("state_dict tables")
    for tag, E, depths in (("base", 48, None), ("tiny", 24, [1, 1, 1, 1]), ("large", 96, None)):
        h = MS.Head(embed_dim=E, num_classes=8)
        if depths is not None:
            h.swin = MS.MicFormer(window_size=(2, 2, 2), in_chans=1, embed_dim=E, depths=depths)
        table = [[k, list(v.shape)] for k, v in h.state_dict().items()]
        with open(os.path.join(HERE, f"state_dict_{tag}.json"), "w") as f:
            json.dump({"embed_dim": E, "depths": depths or [2, 2, 6, 2], "n_params": sum(p.numel() for p in h.parameters()),
                       "n_buffers": len(list(h.buffers())), "keys": table}, f)
        del h

    # ---------------------------------------------------------------- F1 per-module goldens (toy shapes, with backward)
    print("F1 modules")
    for tag, (B, D, H, W, C, heads) in {"c24": (2, 4, 6, 4, 24, 3), "c48": (1, 6, 4, 8, 48, 3), "pad": (1, 5, 3, 4, 24, 3)}.items():
        x = fill.lattice((B, D, H, W, C), f"F1.{tag}.x", 0.8, 0.211).requires_grad_(True)
        xa = fill.lattice((B, D, H, W, C), f"F1.{tag}.xa", 0.7, 0.173).requires_grad_(True)
        gy = fill.lattice((B, D, H, W, C), f"F1.{tag}.gy", 1.0, 0.291)
        out = {}
        # self block
        blk = MS.TransformerBlock3D(dim=C, num_heads=heads, window_size=(2, 2, 2), qkv_bias=True).eval()
        fill.fill_state_dict(blk)
        y = blk(x)
        g = torch.autograd.grad((y * gy).sum(), [x] + list(blk.parameters()))
        out.update(self_y=np32(y), self_gx=np32(g[0]))
        for (n, _), gv in zip(blk.named_parameters(), g[1:]):
            out["self_g." + n] = np32(gv)
        # cross block
        cb = MS.CrossTransformerBlock3D(dim=C, num_heads=heads, window_size=(2, 2, 2), qkv_bias=True).eval()
        fill.fill_state_dict(cb)
        y = cb(x, xa)
        g = torch.autograd.grad((y * gy).sum(), [x, xa] + list(cb.parameters()))
        out.update(cross_y=np32(y), cross_gx=np32(g[0]), cross_gxa=np32(g[1]))
        for (n, _), gv in zip(cb.named_parameters(), g[2:]):
            out["cross_g." + n] = np32(gv)
        out["cross_part1"] = np32(cb.forward_part1(x, xa))
        if tag != "pad":
            # patch merging / expand (own modules)
            pm = MS.PatchMerging(C).eval()
            fill.fill_state_dict(pm)
            y = pm(x)
            gm = fill.lattice(tuple(y.shape), f"F1.{tag}.gm", 1.0, 0.31)
            g = torch.autograd.grad((y * gm).sum(), [x] + list(pm.parameters()))
            out.update(merge_y=np32(y), merge_gx=np32(g[0]))
            for (n, _), gv in zip(pm.named_parameters(), g[1:]):
                out["merge_g." + n] = np32(gv)
            pe = MS.PatchExpand(C).eval()
            fill.fill_state_dict(pe)
            y = pe(x)
            ge = fill.lattice(tuple(y.shape), f"F1.{tag}.ge", 1.0, 0.33)
            g = torch.autograd.grad((y * ge).sum(), [x] + list(pe.parameters()))
            out.update(expand_y=np32(y), expand_gx=np32(g[0]))
            for (n, _), gv in zip(pe.named_parameters(), g[1:]):
                out["expand_g." + n] = np32(gv)
        save(f"f1_modules_{tag}.npz", **out)

    # odd dims through PatchMerging (pad branch MS.py:551-555) and PatchEmbed3D pad (MS.py:864-869)
    x = fill.lattice((1, 5, 3, 6, 24), "F1.odd.x", 0.8, 0.211)
    pm = MS.PatchMerging(24).eval()
    fill.fill_state_dict(pm)
    vol = fill.lattice((2, 1, 9, 8, 10), "F1.odd.vol", 0.9, 0.113)
    pe3 = MS.PatchEmbed3D(patch_size=(4, 4, 4), in_chans=1, embed_dim=24).eval()
    fill.fill_state_dict(pe3)
    yv = pe3(vol)
    gv = fill.lattice(tuple(yv.shape), "F1.odd.gv", 1.0, 0.3)
    gw = torch.autograd.grad((yv * gv).sum(), list(pe3.parameters()))
    save("f1_odd.npz", merge_y=np32(pm(x)), embed_y=np32(yv), embed_gw=np32(gw[0]), embed_gb=np32(gw[1]))

    # ---------------------------------------------------------------- F2 STN / ref-point edge cases
    print("F2 STN")
    out = {}
    stn = MS.SpatialTransformer()
    cb = MS.CrossTransformerBlock3D(dim=8, num_heads=1, window_size=(2, 2, 2))
    for tag, (D, H, W) in {"s1": (1, 1, 1), "s2": (2, 2, 2), "s3": (3, 3, 3), "s5": (5, 5, 5), "nc": (4, 6, 8),
                           "nc2": (3, 5, 2), "flat": (1, 4, 4)}.items():
        src = fill.lattice((2, 8, D, H, W), f"F2.{tag}.src", 1.0, 0.7)            # (B,C,D,H,W)
        off = fill.lattice((2, D, H, W, 3), f"F2.{tag}.off", 1.3, 0.9)
        ref = cb._get_ref_points(D, H, W, 2, torch.float32, "cpu")
        pos = (off + ref).requires_grad_(True)
        srcg = src.clone().requires_grad_(True)
        with np.errstate(all="ignore"):
            y = stn(srcg, pos.permute(0, 4, 1, 2, 3))
        gy = fill.lattice(tuple(y.shape), f"F2.{tag}.gy", 1.0, 0.41)
        gs, gp = torch.autograd.grad((y * gy).sum(), [srcg, pos])
        out[tag + "_ref"] = np32(ref[0])
        out[tag + "_y"] = np32(y.permute(0, 2, 3, 4, 1))                           # channels-last
        out[tag + "_gsrc"] = np32(gs.permute(0, 2, 3, 4, 1))
        out[tag + "_gpos"] = np32(gp)
    save("f2_stn.npz", **out)

    # ---------------------------------------------------------------- F3 tiny config (BASELINE config 1)
    print("F3 tiny 32^3")
    tiny = MS.Head(embed_dim=24, num_classes=8)
    tiny.swin = MS.MicFormer(window_size=(2, 2, 2), in_chans=1, embed_dim=24, depths=[1, 1, 1, 1])
    tiny.eval()
    fill.fill_state_dict(tiny)
    x = fill.make_volume(1, 32, 32, 32)
    lab = fill.make_label_map(1, 32, 32, 32)
    tgt = fill.one_hot(lab)
    crit = dice.MDiceLoss()
    with torch.no_grad():
        logits = tiny(x)
        loss = crit(logits, tgt)
        mask = torch.argmax(torch.softmax(logits, 1), 1)
        md = meandice_ref(mask, lab, 8)
        vloss = dice.MDiceLoss_Val()(logits, tgt)
    save("f3_tiny32.npz", logits=logits.numpy().astype(np.float16), logits_stride=np32(logits[:, :, ::4, ::4, ::4]),
         mask=mask.numpy().astype(np.uint8), loss=np32(loss), val_loss=np32(vloss), meandice=np.float64(md.item()),
         top2_margin_min=np32((logits.topk(2, 1).values[:, 0] - logits.topk(2, 1).values[:, 1]).min()))
    # tiny train-mode-free backward at 32^3 hits S=2 and S=1 stages: store grad finiteness + a few grads
    tiny.zero_grad()
    loss = crit(tiny(x), tgt)
    loss.backward()
    gn = {}
    for n, p in tiny.named_parameters():
        if p.grad is None:
            gn[n] = "none"
        elif not torch.isfinite(p.grad).all():
            gn[n] = "nonfinite"
        else:
            gn[n] = float(p.grad.double().norm())
    with open(os.path.join(HERE, "f3_tiny32_gradnorms.json"), "w") as f:
        json.dump(gn, f)

    # ---------------------------------------------------------------- F4 base @64^3 (+ non-cubic large-like @ (40,40,32))
    print("F4 base 64^3")
    base = MS.Head(embed_dim=48, num_classes=8).eval()
    fill.fill_state_dict(base)
    x = fill.make_volume(1, 64, 64, 64)
    lab = fill.make_label_map(1, 64, 64, 64)
    tgt = fill.one_hot(lab)
    base.zero_grad()
    logits = base(x)
    loss = crit(logits, tgt)
    loss.backward()
    mask = torch.argmax(logits, 1)
    gn = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in base.named_parameters()}
    with open(os.path.join(HERE, "f4_base64_gradnorms.json"), "w") as f:
        json.dump(gn, f)
    top2 = logits.topk(2, 1).values
    save("f4_base64.npz", logits_stride=np32(logits[:, :, ::4, ::4, ::4]), mask=mask.numpy().astype(np.uint8),
         loss=np32(loss), meandice=np.float64(meandice_ref(mask, lab, 8).item()),
         margin=(top2[:, 0] - top2[:, 1]).detach().numpy().astype(np.float16),
         g_out_conv_w=np32(base.out_conv.weight.grad), g_patch_embed_w=np32(base.swin.patch_embed.proj.weight.grad),
         g_l0_b1_q=np32(base.swin.layers[0].blocks1[0].cross_attn.q.weight.grad),
         g_l2_off3=np32(base.swin.layers[2].blocks2[3].conv_offset[3].weight.grad),
         g_up3_fc1_b=np32(base.swin.up_layers[3].self_blocks2[1].mlp.fc1.bias.grad))

    # F5 one Adam step on the same base model (train_mmwhs_noPad.py:114,148,200-207), lr 1e-4, cosine T_max=150
    print("F5 adam")
    opt = torch.optim.Adam(base.parameters(), lr=1e-4, weight_decay=0)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=150)
    names = ["out_conv.weight", "swin.patch_embed.proj.weight", "swin.layers.0.blocks1.0.cross_attn.q.weight",
             "swin.layers.2.blocks2.3.conv_offset.3.weight", "swin.up_layers.3.self_blocks2.1.mlp.fc1.bias",
             "swin.norm2.weight", "swin.concat_back_dim.0.weight"]
    opt.step()
    sched.step()
    sd = base.state_dict()
    sub = lambda t: np32(t.reshape(-1)[::17] if t.numel() > 20000 else t)   # big tensors: every 17th element
    out = {"w1." + n: sub(sd[n]) for n in names}
    out["lr_after_1"] = np.float64(sched.get_last_lr()[0])
    # second step with the same gradients re-computed
    opt.zero_grad()
    loss2 = crit(base(x), tgt)
    loss2.backward()
    opt.step()
    sched.step()
    sd = base.state_dict()
    out.update({"w2." + n: sub(sd[n]) for n in names})
    out["loss2"] = np32(loss2)
    out["lr_after_2"] = np.float64(sched.get_last_lr()[0])
    save("f5_adam.npz", **out)
    del base

    print("F4b non-cubic (40,40,32) E=24 depths [1,1,1,1]  (pad-to-window at the 5x5x4 stage)")
    nc = MS.Head(embed_dim=24, num_classes=8)
    nc.swin = MS.MicFormer(window_size=(2, 2, 2), in_chans=1, embed_dim=24, depths=[1, 1, 1, 1])
    nc.eval()
    fill.fill_state_dict(nc)
    x = fill.make_volume(1, 40, 40, 32)
    lab = fill.make_label_map(1, 40, 40, 32)
    tgt = fill.one_hot(lab)
    nc.zero_grad()
    logits = nc(x)
    loss = crit(logits, tgt)
    loss.backward()
    gn = {n: (float(p.grad.double().norm()) if p.grad is not None else "none") for n, p in nc.named_parameters()}
    with open(os.path.join(HERE, "f4b_noncubic_gradnorms.json"), "w") as f:
        json.dump(gn, f)
    save("f4b_noncubic.npz", logits_stride=np32(logits[:, :, ::2, ::2, ::2]), mask=torch.argmax(logits, 1).numpy().astype(np.uint8),
         loss=np32(loss))
    # odd token grid (36^3 -> 9^3 -> 5^3 -> 3^3 -> 2^3): trilinear-resize branch MS.py:1018-1025
    x = fill.make_volume(1, 36, 36, 36)
    with torch.no_grad():
        logits = nc(x)
    save("f4c_odd36.npz", logits_stride=np32(logits[:, :, ::2, ::2, ::2]), mask=torch.argmax(logits, 1).numpy().astype(np.uint8))

    # ---------------------------------------------------------------- F6 loss closed form on random-ish logits, incl. saturation
    print("F6 loss")
    z = fill.lattice((2, 8, 6, 5, 7), "F6.z", 6.0, 0.77)
    z[0, 0, 0, 0, 0] = 120.0     # sigmoid saturates -> log clamp at -100 (nn.BCELoss)
    z[1, 3, 2, 1, 4] = -120.0
    lab = fill.make_label_map(2, 6, 5, 7)
    t = fill.one_hot(lab)
    zg = z.clone().requires_grad_(True)
    l = crit(zg, t)
    gz, = torch.autograd.grad(l, zg)
    save("f6_loss.npz", z=np32(z), label=lab.numpy().astype(np.uint8), loss=np32(l), gz=np32(gz),
         val_loss=np32(dice.MDiceLoss_Val()(z, t)))
    print("done")


if __name__ == "__main__":
    main()
