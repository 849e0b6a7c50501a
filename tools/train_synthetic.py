#!/usr/bin/env python3
"""The reference's training script in miniature (train_mmwhs_noPad.py:92-241, utils.py:226-262) on synthetic data, to show the
drop-in surface end to end: Head + MDiceLoss + TrainEngine (Adam + per-iteration cosine LR), validation by sliding-window
inference + argmax meandice, checkpoint save / resume in the reference's torch.save layout.

  python tools/train_synthetic.py [--embed-dim 24] [--vol 64] [--steps 30] [--out /tmp/model_best.pth.tar]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micformer_amd import ops
from micformer_amd.engine import TrainEngine
from micformer_amd.inference import sliding_window_inference
from micformer_amd.loss.dice import MDiceLoss, MDiceLoss_Val
from micformer_amd.models.MICFormer_self import Head


def make_case(vol, seed, device):
    """A CT+MR pair whose label regions are visible in both 'modalities' (noisy blocky class map)."""
    g = torch.Generator(device=device).manual_seed(seed)
    coarse = torch.randint(0, 8, (1, vol // 16, vol // 16, vol // 16), generator=g, device=device)
    lab = coarse.repeat_interleave(16, 1).repeat_interleave(16, 2).repeat_interleave(16, 3)
    ct = lab.float() / 7 + 0.3 * torch.randn(lab.shape, generator=g, device=device)
    mr = 1 - lab.float() / 7 + 0.3 * torch.randn(lab.shape, generator=g, device=device)
    return torch.stack([ct, mr], 1), lab.to(torch.uint8)


def validate(model, cases, roi):
    model.eval()
    dices, losses = [], []
    with torch.no_grad():
        for x, lab in cases:
            logits = sliding_window_inference(x, roi, 4, model, overlap=0.5)
            _, md = ops.argmax_meandice(logits, lab)
            dices.append(float(md))
            losses.append(float(MDiceLoss_Val()(logits, lab)))
    model.train()
    return sum(dices) / len(dices), sum(losses) / len(losses)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--embed-dim", type=int, default=24)
    ap.add_argument("--vol", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--out", default="/tmp/model_best.pth.tar")
    ap.add_argument("--dtype", default="fp32", choices=("fp32", "bf16"), help="matrix-core arithmetic (ops.set_compute_dtype)")
    args = ap.parse_args()
    ops.set_compute_dtype(args.dtype)
    dev = torch.device("cuda")
    torch.manual_seed(1234)
    depths = (1, 1, 1, 1) if args.embed_dim < 48 else (2, 2, 6, 2)
    model = Head(embed_dim=args.embed_dim, num_classes=8, depths=depths).to(dev).train()
    eng = TrainEngine(model, base_lr=1e-3, t_max=args.steps, criterion=MDiceLoss(), use_graph=True)
    train = [make_case(args.vol, 100 + i, dev) for i in range(4)]
    val = [make_case(args.vol, 900 + i, dev) for i in range(2)]
    roi = (args.vol // 2,) * 3 if args.vol >= 64 else (args.vol,) * 3
    d0, l0 = validate(model, val, roi)
    t0 = time.perf_counter()
    for it in range(args.steps):
        xa, la = train[(2 * it) % 4]
        xb, lb = train[(2 * it + 1) % 4]
        loss = eng.step(torch.cat([xa, xb]), torch.cat([la, lb]))          # uint8 class maps as targets
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d1, l1 = validate(model, val, roi)
    torch.save(eng.checkpoint(epoch=1), args.out)
    model2 = Head(embed_dim=args.embed_dim, num_classes=8, depths=depths).to(dev).train()
    eng2 = TrainEngine(model2, base_lr=1e-3, t_max=args.steps, use_graph=False)
    epoch = eng2.load_checkpoint(torch.load(args.out, map_location=dev, weights_only=False))
    d2, l2 = validate(model2, val, roi)
    print(json.dumps({"train_loss_last": float(loss), "val_meandice": [round(d0, 4), round(d1, 4)], "val_loss": [round(l0, 4), round(l1, 4)],
                      "resumed_epoch": epoch, "resumed_val_meandice": round(d2, 4), "s_per_step": round(dt / args.steps, 4)}))
    # (split reductions and the small grids' fp32 atomics make two forwards of the same weights differ by ~1e-6 on the logits,
    # which can flip a handful of argmax voxels: the resumed model's meandice is compared to 1e-3)
    assert abs(d2 - d1) < 1e-3 and l1 < l0, "checkpoint round trip / learning sanity"


if __name__ == "__main__":
    main()
