"""Generate golden vectors by IMPORTING THE REFERENCE in the build container.

Run here (never on the GPU box -- /root/reference does not exist there):

    python tests/golden/make_golden.py

Writes small .npz fixtures next to this file.  Each holds the inputs and the outputs the
reference's own functions produced for them:

  code/FasterRCNN/RPN/anchor_generator.py  AnchorGenerator            -> anchors.npz
  code/FasterRCNN/utils.py                 box_iou / tagging_box /
                                           box_deviation              -> targets.npz
  code/FasterRCNN/utils.py                 box_regression / box_clip /
                                           remove_small_box           -> decode.npz
  code/FasterRCNN/utils.py                 box_regression, backward
                                           (torch autograd through it) -> decode_grad.npz
  code/FasterRCNN/utils.py                 smooth_l1_loss + torch BCE/CE on
                                           sampled rows, with gradients -> losses.npz
  code/FasterRCNN/generalized_transform.py GeneralizedTransform.forward,
  code/FasterRCNN/utils.py                 batched_images             -> transform.npz
  code/FasterRCNN/utils.py                 get_sampling_factor        -> sampling.npz
  RPN/rpn.py:90-137 and ROIHead/roi_head.py:99-124,141-152 (the training
  halves: labels, encodings, get_sampling_factor under torch.manual_seed,
  negatives-first concat, gathers, losses) replayed line by line with the
  reference's utils functions                                          -> train_sampling.npz
  RPN/rpn.py:55-83 call sequence replayed with the reference's utils
  functions and torch.topk                                             -> rpn_stages.npz
  RPN/rpn.py, ROIHead/roi_head.py, ROIHead/roi_pooling.py THEMSELVES, imported
  and run (RPN.forward eval + training, ROIHead.forward eval, ROIPooling.forward)
  with `torchvision.ops` pre-seeded by a module whose nms / MultiScaleRoIAlign are
  the CPU oracle's: the files' own orchestration, not a transcription  -> ref_modules.npz

`utils.py` imports cv2 at module level (utils.py:5) only for draw_anchor_box; an empty
module object is registered under that name so the import succeeds.  Nothing on the hot
path touches it.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference/code/FasterRCNN"
HERE = os.path.dirname(os.path.abspath(__file__))

warnings.filterwarnings("ignore")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "RPN"))

import utils as R  # noqa: E402  (the reference's utils.py)
from anchor_generator import AnchorGenerator  # noqa: E402


def checksum(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum())


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(name, os.path.getsize(path), "bytes")


def gen_anchors():
    out = {}
    cases = {
        "smoke": ((2, 256, 256), [(112, 112), (64, 64)], None, None),
        "cfg1": ((1, 600, 1000), [(37, 62)], None, None),
        "fpn": ((2, 800, 1344), [(13, 21), (25, 42), (50, 84)], None, None),
        "small_full": ((3, 64, 96), [(4, 6), (3, 5)], None, None),
        "custom": ((1, 100, 150), [(7, 9)], ((32,), (64,), (100,)), ((0.5, 1.0, 2.0), (0.25, 1.0, 3.0), (0.7, 1.3, 1.9))),
    }
    for name, (info, fsz, sizes, ratios) in cases.items():
        ag = AnchorGenerator() if sizes is None else AnchorGenerator(sizes, ratios)
        feats = [torch.zeros(info[0], 1, h, w) for h, w in fsz]
        res = ag(info, feats)
        out[f"{name}.info"] = np.asarray(info)
        out[f"{name}.fsz"] = np.asarray(fsz)
        out[f"{name}.base"] = ag.base_anchor.numpy()
        if sizes is not None:
            out[f"{name}.sizes"] = np.asarray(sizes, np.float32)
            out[f"{name}.ratios"] = np.asarray(ratios, np.float32)
        for l, a in enumerate(res):
            assert a.shape[0] == info[0] and a.shape[2:] == (9, 4)
            a0 = a[0].contiguous()
            out[f"{name}.l{l}.checksum"] = np.asarray(checksum(a0))
            out[f"{name}.l{l}.shape"] = np.asarray(a.shape)
            if a0.shape[0] <= 64:
                out[f"{name}.l{l}.full"] = a0.numpy()
            else:
                out[f"{name}.l{l}.head"] = a0[:32].numpy()
                out[f"{name}.l{l}.tail"] = a0[-32:].numpy()
    save("anchors.npz", **out)


def rand_gt(g, m, W=1000.0, H=600.0):
    w = torch.rand(m, generator=g) * 240 + 16
    h = torch.rand(m, generator=g) * 240 + 16
    x1 = torch.rand(m, generator=g) * (W - w)
    y1 = torch.rand(m, generator=g) * (H - h)
    return torch.stack([x1, y1, x1 + w, y1 + h], 1)


def gen_targets():
    g = torch.Generator().manual_seed(20211117)
    ag = AnchorGenerator()
    anchors = ag((1, 600, 1000), [torch.zeros(1, 1, 37, 62)])[0][0].reshape(-1, 4)
    out = {}
    # (name, N, M, neg, pos, with_labels, weights)
    cases = [
        ("m1", 2048, 1, 0.3, 0.7, False, None),
        ("m2", 2048, 2, 0.3, 0.7, True, [10.0, 10.0, 5.0, 5.0]),
        ("m100", 1536, 100, 0.3, 0.7, False, None),
        ("m100w", 1024, 100, 0.5, 0.5, True, [10.0, 10.0, 5.0, 5.0]),
        ("m7swap", 512, 7, 0.6, 0.4, True, None),  # pos < neg: assignment order matters
    ]
    for name, N, M, neg, pos, with_labels, weights in cases:
        sel = torch.randperm(anchors.shape[0], generator=g)[:N].sort().values
        box1 = anchors[sel].clone()
        box2 = rand_gt(g, M)
        if M >= 2:
            box2[1] = box2[0]  # duplicate GT: first-max tie rule
        # adversarial rows
        box1[0] = box2[0]  # IoU == 1
        box1[1] = torch.tensor([2000.0, 2000.0, 2100.0, 2100.0])  # IoU == 0 with every GT
        box1[2] = torch.tensor([10.0, 10.0, 10.0, 10.0])  # zero area
        # exact threshold hit: GT0 = [x,y,x+w,y+h]; box with same x-range and half the height -> IoU 0.5
        x1, y1, x2, y2 = box2[0].tolist()
        box1[3] = torch.tensor([x1, y1, x2, y1 + (y2 - y1) / 2])
        labels = (torch.randint(1, 21, (M,), generator=g).float()) if with_labels else None
        iou = R.box_iou(box1, box2)
        lab = R.tagging_box(box1, box2, neg, pos, labels)
        dev = R.box_deviation(box1, box2, weights)
        mx, arg = iou.max(dim=1)
        out[f"{name}.box1"] = box1.numpy()
        out[f"{name}.box2"] = box2.numpy()
        out[f"{name}.thr"] = np.asarray([neg, pos], np.float64)
        if labels is not None:
            out[f"{name}.labels"] = labels.numpy()
        if weights is not None:
            out[f"{name}.weights"] = np.asarray(weights, np.float32)
        if N * M <= 1536 * 100:
            out[f"{name}.iou"] = iou.numpy()
        out[f"{name}.max"] = mx.numpy()
        out[f"{name}.argmax"] = arg.numpy()
        out[f"{name}.label"] = lab.numpy()
        out[f"{name}.dev"] = dev.numpy()
    # degenerate pairs: 0/0 -> NaN (box_iou has no epsilon); NaN rows keep label 0
    box1 = torch.tensor([[5.0, 5.0, 5.0, 5.0], [0.0, 0.0, 10.0, 10.0], [1.0, 1.0, 1.0, 9.0]])
    box2 = torch.tensor([[0.0, 0.0, 4.0, 4.0], [5.0, 5.0, 5.0, 5.0], [2.0, 2.0, 8.0, 8.0]])
    iou = R.box_iou(box1, box2)
    mx, arg = iou.max(dim=1)
    out["nan.box1"] = box1.numpy()
    out["nan.box2"] = box2.numpy()
    out["nan.iou"] = iou.numpy()
    out["nan.max"] = mx.numpy()
    out["nan.argmax"] = arg.numpy()
    out["nan.label"] = R.tagging_box(box1, box2, 0.3, 0.7).numpy()
    out["nan.thr"] = np.asarray([0.3, 0.7])
    save("targets.npz", **out)


def gen_decode():
    g = torch.Generator().manual_seed(777)
    ag = AnchorGenerator()
    anchors = ag((1, 600, 1000), [torch.zeros(1, 1, 37, 62)])[0][0].reshape(-1, 4)
    n = 4096
    sel = torch.randperm(anchors.shape[0], generator=g)[:n]
    box = anchors[sel].clone()
    out = {}
    for K in (1, 3):
        regs = torch.randn(n, 4 * K, generator=g)
        regs[:, 0::4] *= 0.1
        regs[:, 1::4] *= 0.1
        regs[:, 2::4] *= 0.2
        regs[:, 3::4] *= 0.2
        regs[0, 2::4] = 9.0  # above the log(1000/16) clamp
        regs[1, 3::4] = 4.135166556742356
        regs[2, :] = 0.0
        regs[3, 2::4] = -30.0  # collapses the width
        dec = R.box_regression(box, regs)
        out[f"k{K}.box"] = box.numpy()
        out[f"k{K}.regs"] = regs.numpy()
        out[f"k{K}.decoded"] = dec.numpy()
    # clip + small-box filter on wider-ranging boxes
    raw = (torch.rand(n, 4, generator=g) - 0.25) * torch.tensor([1400.0, 900.0, 1400.0, 900.0])
    raw[0] = torch.tensor([-5.0, -7.0, 1000.0, 600.0])
    raw[1] = torch.tensor([1000.0, 600.0, 1001.0, 601.0])
    raw[2] = torch.tensor([10.0, 10.0, 10.001, 20.0])  # w == 1e-3 (fp32) edge: strict >
    raw[3] = torch.tensor([10.0, 10.0, 10.0 + 2e-3, 20.0])
    clipped = R.box_clip(raw, (600, 1000))
    out["clip.in"] = raw.numpy()
    out["clip.image_size"] = np.asarray([600, 1000])
    out["clip.out"] = clipped.numpy()
    for t in (1e-3, 16.0):
        kept, idx = R.remove_small_box(clipped, t)
        out[f"small.{t}.index"] = idx.numpy()
        out[f"small.{t}.box"] = kept.numpy()
    save("decode.npz", **out)


def gen_decode_grad():
    """Gradients the reference's own autograd produces through utils.box_regression (utils.py:173-208):
    the reference never detaches its proposals (rpn.py:75,85), so decode is on its backward path."""
    g = torch.Generator().manual_seed(4242)
    out = {}
    for K in (1, 3):
        n = 512
        xy = torch.rand(n, 2, generator=g) * 400
        wh = torch.rand(n, 2, generator=g) * 300 + 1
        box = torch.cat([xy, xy + wh], 1).requires_grad_(True)
        regs = torch.randn(n, 4 * K, generator=g)
        regs[:, 0::4] *= 0.1
        regs[:, 1::4] *= 0.1
        regs[:, 2::4] *= 0.2
        regs[:, 3::4] *= 0.2
        regs[0, 2::4] = 9.0                      # above the clamp: no gradient to dw
        regs[1, 3::4] = 4.135166556742356        # fp32(log(1000/16)) rounds just above the clamp
        regs[2, 2::4] = 4.0                      # below the clamp
        regs[3, :] = 0.0
        regs = regs.requires_grad_(True)
        upstream = torch.randn(n, 4 * K, generator=g)
        dec = R.box_regression(box, regs)
        dec.backward(upstream)
        out[f"k{K}.box"] = box.detach().numpy()
        out[f"k{K}.regs"] = regs.detach().numpy()
        out[f"k{K}.upstream"] = upstream.numpy()
        out[f"k{K}.grad_regs"] = regs.grad.numpy()
        out[f"k{K}.grad_box"] = box.grad.numpy()
    save("decode_grad.npz", **out)


def gen_losses():
    """the sampled-gather + loss tails of RPN.forward (rpn.py:122-137) and ROIHead.forward
    (roi_head.py:141-152) -- torch's F.binary_cross_entropy_with_logits / F.cross_entropy and the
    reference's utils.smooth_l1_loss (beta = 1/9, mean over ALL gathered elements: quirk Q7), with the
    gradients torch autograd sends back to the dense head outputs."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9001)
    out = {}
    # --- RPN: B images x A anchors, S sampled anchors per image (negatives first, rpn.py:122)
    B, A, S = 2, 3000, 256
    scores = (torch.randn(B, A, generator=g) * 3).requires_grad_(True)
    regs = torch.randn(B, A, 4, generator=g).requires_grad_(True)
    labels = torch.full((B, A), -1.0)
    reg_labels = torch.randn(B, A, 4, generator=g) * 0.3
    reg_labels[0, :50] *= 0.01                       # |x| < beta branch
    index = []
    for b in range(B):
        perm = torch.randperm(A, generator=g)
        neg, pos = perm[:S - 40 - 10 * b], perm[S - 40 - 10 * b:S]
        labels[b, neg] = 0.0
        labels[b, pos] = 1.0
        index.append(torch.cat([neg, pos]))
    scores.data[0, index[0][0]] = 40.0               # saturated logits
    scores.data[0, index[0][1]] = -40.0
    cls_y = torch.cat([scores[b][index[b]] for b in range(B)])
    cls_l = torch.cat([labels[b][index[b]] for b in range(B)])
    reg_y = torch.cat([regs[b][index[b]] for b in range(B)])
    reg_l = torch.cat([reg_labels[b][index[b]] for b in range(B)])
    cls_loss = F.binary_cross_entropy_with_logits(cls_y, cls_l)
    reg_loss = R.smooth_l1_loss(reg_y, reg_l)
    (cls_loss * 1.5 + reg_loss * 0.75).backward()
    out.update({"rpn.scores": scores.detach().numpy(), "rpn.regs": regs.detach().numpy(), "rpn.labels": labels.numpy(),
                "rpn.reg_labels": reg_labels.numpy(),
                "rpn.index": torch.cat([index[b] + b * A for b in range(B)]).numpy(),
                "rpn.upstream": np.asarray([1.5, 0.75], np.float32),
                "rpn.cls_loss": cls_loss.detach().numpy(), "rpn.reg_loss": reg_loss.detach().numpy(),
                "rpn.grad_scores": scores.grad.numpy(), "rpn.grad_regs": regs.grad.numpy()})
    # --- RoI head: N sampled proposals, Kc classes (0 = background), per-class regression
    N, Kc = 384, 21
    logits = (torch.randn(N, Kc, generator=g) * 2).requires_grad_(True)
    pred = torch.randn(N, Kc * 4, generator=g).requires_grad_(True)
    cls_label = torch.zeros(N)
    pos = torch.randperm(N, generator=g)[:90]
    cls_label[pos] = torch.randint(1, Kc, (90,), generator=g).float()
    reg_label = torch.randn(N, 4, generator=g) * 0.5
    ce = F.cross_entropy(logits, cls_label.long())
    pi = torch.where(torch.gt(cls_label, 0))[0]
    rl = R.smooth_l1_loss(pred.view(N, -1, 4)[pi, cls_label[pi].long()], reg_label[pi])
    (ce * 0.5 + rl * 2.0).backward()
    out.update({"roi.logits": logits.detach().numpy(), "roi.pred": pred.detach().numpy(), "roi.cls_label": cls_label.numpy(),
                "roi.reg_label": reg_label.numpy(), "roi.upstream": np.asarray([0.5, 2.0], np.float32),
                "roi.cls_loss": ce.detach().numpy(), "roi.reg_loss": rl.detach().numpy(),
                "roi.grad_logits": logits.grad.numpy(), "roi.grad_pred": pred.grad.numpy()})
    save("losses.npz", **out)


def gen_transform():
    """batch assembly before the path -- GeneralizedTransform.forward
    (generalized_transform.py:44-83: bilinear resize to max_size, GT rescale, pad to a multiple of 32 via
    utils.batched_images, utils.py:339-372), bugs included (inverse box ratio, boxes flattened to 1-D by
    torch.cat, the `elif` in the max-width scan)."""
    from generalized_transform import GeneralizedTransform
    g = torch.Generator().manual_seed(31337)
    out = {}
    cases = {"a": (80, 96, [(3, 37, 50), (3, 60, 41), (3, 33, 33)]),      # (min_size, max_size, image shapes)
             "b": (64, 64, [(3, 64, 48), (3, 20, 64)]),                   # scale 1 on one side
             "c": (100, 150, [(1, 11, 7), (1, 9, 12)]),                   # strong upsample, one channel
             "d": (100, 150, [(1, 11, 7)])}                               # single image: the `elif` leaves width 0
    for name, (mn, mx, shapes) in cases.items():
        images = [torch.rand(*s, generator=g) for s in shapes]
        targets = []
        for s in shapes:
            m = 3
            xy = torch.rand(m, 2, generator=g) * torch.tensor([s[2] / 2.0, s[1] / 2.0])
            wh = torch.rand(m, 2, generator=g) * torch.tensor([s[2] / 2.0, s[1] / 2.0]) + 1
            targets.append({"box": torch.cat([xy, xy + wh], 1)})
        for i, (im, t) in enumerate(zip(images, targets)):
            out[f"{name}.image{i}"] = im.numpy().copy()
            out[f"{name}.box{i}"] = t["box"].numpy().copy()
        out[f"{name}.sizes"] = np.asarray([mn, mx, len(shapes)])
        try:
            batch, new_targets = GeneralizedTransform(mn, mx)(list(images), targets)
        except RuntimeError:
            out[f"{name}.raises"] = np.asarray([1])      # the reference itself fails on this input
            continue
        out[f"{name}.raises"] = np.asarray([0])
        out[f"{name}.batch"] = batch.numpy()
        for i, t in enumerate(new_targets):
            out[f"{name}.new_box{i}"] = t["box"].numpy()
    # batched_images alone (utils.py:339-372)
    imgs = [torch.rand(3, 30, 45, generator=g), torch.rand(3, 50, 20, generator=g), torch.rand(3, 40, 70, generator=g)]
    for i, im in enumerate(imgs):
        out[f"pad.image{i}"] = im.numpy()
    out["pad.batch"] = R.batched_images(list(imgs), 32).numpy()
    save("transform.npz", **out)


def gen_sampling():
    out = {}
    cases = [(256, 0.5, 40, 18000), (256, 0.5, 300, 18000), (128, 0.25, 10, 1500), (512, 0.25, 0, 60),
             (256, 0.5, 3, 5)]
    for i, (S, f, npos, nneg) in enumerate(cases):
        torch.manual_seed(1234 + i)
        pf, nf = R.get_sampling_factor(S, f, npos, nneg)
        out[f"c{i}.args"] = np.asarray([S, f, npos, nneg], np.float64)
        out[f"c{i}.seed"] = np.asarray(1234 + i)
        out[f"c{i}.pos"] = pf.numpy()
        out[f"c{i}.neg"] = nf.numpy()
    save("sampling.npz", **out)


def tie_free_scores(g, shape_bahw):
    B, A_, H, W = shape_bahw
    n = A_ * H * W
    s = torch.stack([((torch.randperm(n, generator=g).float() / n) - 0.5) * 8 for _ in range(B)])
    return s.reshape(B, H, W, A_).permute(0, 3, 1, 2).contiguous()


def gen_rpn_stages():
    """Replay RPN/rpn.py:55-83 (everything before the torchvision nms call) with the
    reference's own utils functions and torch.topk on tie-free logits."""
    g = torch.Generator().manual_seed(4242)
    info = (2, 192, 320)
    fsz = [(12, 20), (6, 10)]
    ag = AnchorGenerator()
    feats = [torch.zeros(info[0], 1, h, w) for h, w in fsz]
    list_anchors = ag(info, feats)
    list_cls = [tie_free_scores(g, (info[0], 9, h, w)) for h, w in fsz]
    # make scores distinct ACROSS levels too: shift level 1 by an irrational-ish offset
    list_cls[1] = list_cls[1] + 0.000123
    list_regs = []
    for h, w in fsz:
        r = torch.randn(info[0], 36, h, w, generator=g)
        r[:, 0::4] *= 0.1
        r[:, 1::4] *= 0.1
        r[:, 2::4] *= 0.2
        r[:, 3::4] *= 0.2
        list_regs.append(r)
    num_pre_nms, min_size = 1000, 1e-3
    anchors = torch.cat([a.flatten(1, 2) for a in list_anchors], dim=1)
    cls = torch.cat([c.permute(0, 2, 3, 1).flatten(1, 3) for c in list_cls], dim=1)
    reg = torch.cat([r.permute(0, 2, 3, 1).flatten(1, 2) for r in list_regs], dim=1)
    assert all(len(torch.unique(cls[b])) == cls.shape[1] for b in range(info[0])), "scores must be tie-free"
    out = {"info": np.asarray(info), "fsz": np.asarray(fsz), "num_pre_nms": np.asarray(num_pre_nms),
           "min_size": np.asarray(min_size)}
    for l in range(len(fsz)):
        out[f"cls{l}"] = list_cls[l].numpy()
        out[f"reg{l}"] = list_regs[l].numpy()
    for b, (image_score, image_reg, image_anchor) in enumerate(zip(cls, reg, anchors)):
        k = min(num_pre_nms, image_score.size(0))
        _, top_index = torch.topk(image_score, k=k)
        top_cls = image_score[top_index]
        top_reg = image_reg.view(-1, 4)[top_index]
        top_anchor = image_anchor[top_index]
        proposal = R.box_regression(top_anchor, top_reg)
        clipped = R.box_clip(proposal, (info[1], info[2]))
        kept, index = R.remove_small_box(clipped, min_size)
        out[f"b{b}.top_index"] = top_index.numpy()
        out[f"b{b}.top_score"] = top_cls.numpy()
        out[f"b{b}.decoded"] = proposal.numpy()
        out[f"b{b}.clipped"] = clipped.numpy()
        out[f"b{b}.small_index"] = index.numpy()
        out[f"b{b}.filtered"] = kept.numpy()
        out[f"b{b}.filtered_score"] = top_cls[index].numpy()
    save("rpn_stages.npz", **out)


def gen_train_sampling():
    """the ORCHESTRATION of the two training halves.  rpn.py and roi_head.py cannot be
    imported (torchvision), so their training branches are replayed here statement by statement with the
    reference's own utils functions on CPU under a fixed torch.manual_seed: per image tagging_box,
    box_deviation, where(), get_sampling_factor (positives drawn first, utils.py:310-311), the
    NEGATIVES-FIRST concat (rpn.py:122, roi_head.py:121), the gathers and the two losses."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(8675309)
    out = {}
    # ---------------- RPN.forward training half: rpn.py:90-137
    info, fsz = (2, 192, 320), [(12, 20), (6, 10)]
    num_sample, pos_fraction, neg_thr, pos_thr, seed = 64, 0.5, 0.3, 0.7, 2024
    ag = AnchorGenerator()
    list_anchors = ag(info, [torch.zeros(info[0], 1, h, w) for h, w in fsz])
    list_cls = [torch.randn(info[0], 9, h, w, generator=g) for h, w in fsz]
    list_regs = [torch.randn(info[0], 36, h, w, generator=g) * 0.3 for h, w in fsz]
    flat = torch.cat(list_anchors, dim=1).flatten(1, 2)[0]
    # ground truth = jittered copies of anchors (so that IoU > 0.7 positives exist): 10 boxes for image 0
    # (more positives than num_sample * pos_fraction), 2 for image 1 (fewer: quirk Q9, no negative top-up)
    list_targets = []
    for m, lo, hi in ((10, 33, 10 ** 6), (2, 1, 31)):
        while True:  # draw until the image has the wanted number of positives
            pick = flat[torch.randperm(flat.shape[0], generator=g)[:m]]
            gt = pick + torch.randn(m, 4, generator=g) * (1.5 if m > 2 else 8.0)
            n_pos = int((R.tagging_box(flat, gt, neg_thr, pos_thr) == 1).sum())
            if lo <= n_pos <= hi:
                break
        list_targets.append({"box": gt.tolist()})
    batch_size = info[0]
    l_cls = [c.permute(0, 2, 3, 1).flatten(1, 3) for c in list_cls]                                  # rpn.py:90-91
    l_regs = [r.permute(0, 2, 3, 1).flatten(1, 2).reshape(batch_size, -1, 4) for r in list_regs]     # rpn.py:92-93
    anchors = torch.cat(list_anchors, dim=1).flatten(1, 2)                                           # rpn.py:96
    scores = torch.cat(l_cls, dim=1)
    regs = torch.cat(l_regs, dim=1)
    torch.manual_seed(seed)
    l_reg_label, l_reg_pred, l_cls_label, l_cls_pred = [], [], [], []
    for b, (anchor, score, reg, target) in enumerate(zip(anchors, scores, regs, list_targets)):      # rpn.py:105
        ground_truth = torch.tensor(target["box"], dtype=torch.float32)
        cls_label = R.tagging_box(anchor, ground_truth, neg_thr, pos_thr)
        reg_label = R.box_deviation(anchor, ground_truth)
        neg_index = torch.where(torch.eq(cls_label, 0))[0]
        pos_index = torch.where(torch.eq(cls_label, 1))[0]
        pos_factor, neg_factor = R.get_sampling_factor(num_sample, pos_fraction, pos_index.size(0), neg_index.size(0))
        random_index = torch.cat([neg_index[neg_factor], pos_index[pos_factor]])                     # rpn.py:122
        l_reg_label.append(reg_label[random_index])
        l_cls_label.append(cls_label[random_index])
        l_reg_pred.append(reg[random_index])
        l_cls_pred.append(score[random_index])
        out[f"rpn.b{b}.gt"] = ground_truth.numpy()
        out[f"rpn.b{b}.n_pos_neg"] = np.asarray([pos_index.size(0), neg_index.size(0)])
        out[f"rpn.b{b}.random_index"] = random_index.numpy()
        out[f"rpn.b{b}.cls_label"] = l_cls_label[-1].numpy()
        out[f"rpn.b{b}.reg_label"] = l_reg_label[-1].numpy()
    cls_loss = F.binary_cross_entropy_with_logits(torch.cat(l_cls_pred), torch.cat(l_cls_label))     # rpn.py:136
    reg_loss = R.smooth_l1_loss(torch.cat(l_reg_pred), torch.cat(l_reg_label))                       # rpn.py:137
    assert out["rpn.b0.n_pos_neg"][0] > num_sample * pos_fraction > out["rpn.b1.n_pos_neg"][0] > 0
    out.update({"rpn.info": np.asarray(info), "rpn.fsz": np.asarray(fsz), "rpn.seed": np.asarray(seed),
                "rpn.params": np.asarray([num_sample, pos_fraction, neg_thr, pos_thr], np.float64),
                "rpn.cls_loss": cls_loss.numpy(), "rpn.reg_loss": reg_loss.numpy()})
    for l in range(len(fsz)):
        out[f"rpn.cls{l}"] = list_cls[l].numpy()
        out[f"rpn.reg{l}"] = list_regs[l].numpy()
    # ---------------- ROIHead.forward training half: roi_head.py:99-124, 141-152
    num_sample, pos_fraction, neg_thr, pos_thr, seed = 64, 0.25, 0.5, 0.5, 77
    weights = [10.0, 10.0, 5.0, 5.0]
    Kc = 21
    list_props, list_targets = [], []
    for n_prop, m, n_near in ((300, 6, 60), (180, 2, 5)):
        gt = rand_gt(g, m, W=320.0, H=192.0) * torch.tensor([0.3, 0.3, 0.3, 0.3]) + torch.tensor([20.0, 20.0, 60.0, 60.0])
        xy = torch.rand(n_prop, 2, generator=g) * torch.tensor([260.0, 150.0])
        wh = torch.rand(n_prop, 2, generator=g) * 80 + 4
        props = torch.cat([xy, xy + wh], 1)
        near = gt[torch.randint(0, m, (n_near,), generator=g)] + torch.randn(n_near, 4, generator=g) * 2.0
        props[torch.randperm(n_prop, generator=g)[:n_near]] = near
        list_props.append(props)
        list_targets.append({"class": torch.randint(1, Kc, (m,), generator=g).float().tolist(), "box": gt.tolist()})
    torch.manual_seed(seed)
    l_cls_label, l_reg_label, l_props = [], [], []
    for b, (proposal, target) in enumerate(zip(list_props, list_targets)):                           # roi_head.py:99
        labels = torch.as_tensor(target["class"])
        ground_truth = torch.as_tensor(target["box"], dtype=torch.float32)
        cls_label = R.tagging_box(proposal, ground_truth, neg_thr, pos_thr, labels)
        reg_label = R.box_deviation(proposal, ground_truth, weights)
        pos_index = torch.where(torch.gt(cls_label, 0))[0]
        neg_index = torch.where(torch.eq(cls_label, 0))[0]
        pos_factor, neg_factor = R.get_sampling_factor(num_sample, pos_fraction, pos_index.size(0), neg_index.size(0))
        random_index = torch.cat([neg_index[neg_factor], pos_index[pos_factor]])                     # roi_head.py:121
        l_cls_label.append(cls_label[random_index])
        l_reg_label.append(reg_label[random_index])
        l_props.append(proposal[random_index])
        out[f"roi.b{b}.proposals"] = proposal.numpy()
        out[f"roi.b{b}.gt"] = ground_truth.numpy()
        out[f"roi.b{b}.class"] = labels.numpy()
        out[f"roi.b{b}.n_pos_neg"] = np.asarray([pos_index.size(0), neg_index.size(0)])
        out[f"roi.b{b}.random_index"] = random_index.numpy()
        out[f"roi.b{b}.cls_label"] = l_cls_label[-1].numpy()
        out[f"roi.b{b}.reg_label"] = l_reg_label[-1].numpy()
        out[f"roi.b{b}.sampled_proposals"] = l_props[-1].numpy()
    assert out["roi.b0.n_pos_neg"][0] > num_sample * pos_fraction > out["roi.b1.n_pos_neg"][0] > 0
    cls_label = torch.cat(l_cls_label)                                                               # roi_head.py:141
    reg_label = torch.cat(l_reg_label)
    N = cls_label.shape[0]
    predict_cls = torch.randn(N, Kc, generator=g) * 2        # what the (out of scope) dense heads would emit
    predict_reg = (torch.randn(N, Kc * 4, generator=g) * 0.5).view(N, -1, 4)
    ce = F.cross_entropy(predict_cls, cls_label.long())                                              # roi_head.py:144
    pos_index = torch.where(torch.gt(cls_label, 0))[0]
    rl = R.smooth_l1_loss(predict_reg[pos_index, cls_label[pos_index].long()], reg_label[pos_index])  # :147-152
    out.update({"roi.seed": np.asarray(seed), "roi.weights": np.asarray(weights, np.float32),
                "roi.params": np.asarray([num_sample, pos_fraction, neg_thr, pos_thr], np.float64),
                "roi.predict_cls": predict_cls.numpy(), "roi.predict_reg": predict_reg.reshape(N, -1).numpy(),
                "roi.cls_loss": ce.numpy(), "roi.reg_loss": rl.numpy()})
    save("train_sampling.npz", **out)


def _seed_torchvision_with_the_oracle():
    """torchvision is not installed here and the reference does not vendor it.  rpn.py:18,
    roi_head.py:26 and roi_pooling.py:18 import two names from torchvision.ops; a module object that provides
    them -- implemented by the CPU oracle's restatement of the published algorithms -- lets the reference's
    files be imported and RUN, so that their orchestration (concat order, views, filter order, slicing) is
    recorded from the files themselves.  The arithmetic of nms / roi_align stays what the oracle restates:
    this route does not pin it (closed_form.npz is the independent check for those)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle as orc

    def nms(boxes, scores, iou_threshold):
        keep = orc.nms(boxes.detach().numpy().astype(np.float32), scores.detach().numpy().astype(np.float32),
                       float(iou_threshold))
        return torch.as_tensor(np.asarray(keep, np.int64))

    class MultiScaleRoIAlign(torch.nn.Module):
        def __init__(self, featmap_names, output_size, sampling_ratio):
            super().__init__()
            self.featmap_names, self.output_size, self.sampling_ratio = featmap_names, output_size, sampling_ratio

        def forward(self, x, boxes, image_shapes):
            feats = [v.detach().numpy() for k, v in x.items() if k in self.featmap_names]
            out, _ = orc.multiscale_roi_align(feats, [b.detach().numpy() for b in boxes], image_shapes,
                                              tuple(self.output_size), self.sampling_ratio)
            return torch.as_tensor(out)

    tv, ops = types.ModuleType("torchvision"), types.ModuleType("torchvision.ops")
    ops.nms, ops.MultiScaleRoIAlign = nms, MultiScaleRoIAlign
    tv.ops = ops
    sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, ops
    return orc


class _FixedHead(torch.nn.Module):
    """Stands in for the (out of scope) conv head handed to the reference's RPN: returns recorded outputs."""

    def __init__(self, cls, regs):
        super().__init__()
        self.c, self.r = cls, regs

    def forward(self, feats):
        return list(self.c), list(self.r)


def gen_reference_modules():
    """fixtures that the reference's own rpn.py / roi_head.py / roi_pooling.py produced by running."""
    orc = _seed_torchvision_with_the_oracle()
    sys.path.insert(0, os.path.join(REF, "ROIHead"))
    import rpn as ref_rpn                      # /root/reference/code/FasterRCNN/RPN/rpn.py
    import roi_head as ref_roi_head            # /root/reference/code/FasterRCNN/ROIHead/roi_head.py
    import roi_pooling as ref_roi_pooling      # /root/reference/code/FasterRCNN/ROIHead/roi_pooling.py
    for m, f in ((ref_rpn, "RPN/rpn.py"), (ref_roi_head, "ROIHead/roi_head.py"), (ref_roi_pooling, "ROIHead/roi_pooling.py")):
        assert os.path.abspath(m.__file__) == os.path.join(REF, f), m.__file__
    out = {}
    # ---- RPN.forward, training mode, on the inputs of train_sampling.npz (rpn.py:46-142 in one go: proposals of
    # the eval half AND the losses of the training half under the same seed).  The transcription in
    # gen_train_sampling must give the very same losses: that checks the transcription against the file.
    ts = np.load(os.path.join(HERE, "train_sampling.npz"))
    info = tuple(int(v) for v in ts["rpn.info"])
    fsz = [tuple(int(v) for v in hw) for hw in ts["rpn.fsz"]]
    S, f, neg, pos = ts["rpn.params"]
    head = _FixedHead([torch.as_tensor(ts[f"rpn.cls{l}"]) for l in range(len(fsz))],
                      [torch.as_tensor(ts[f"rpn.reg{l}"]) for l in range(len(fsz))])
    params = dict(anchor_generator=AnchorGenerator(), rpn_head=head, num_pre_nms=500, num_post_nms=100,
                  nms_threshold=0.7, proposal_min_size=1e-3, pos_threshold=float(pos), neg_threshold=float(neg),
                  num_sample=int(S), pos_fraction=float(f))
    targets = [{"box": ts[f"rpn.b{b}.gt"].tolist()} for b in range(info[0])]
    feats = [torch.zeros(info[0], 1, h, w) for h, w in fsz]
    rpn = ref_rpn.RPN(params).train()
    torch.manual_seed(int(ts["rpn.seed"]))
    props, loss = rpn(info, feats, targets)
    assert abs(float(loss["cls_loss"]) - float(ts["rpn.cls_loss"])) < 1e-6, "transcription and rpn.py disagree"
    assert abs(float(loss["reg_loss"]) - float(ts["rpn.reg_loss"])) < 1e-6, "transcription and rpn.py disagree"
    out.update({"rpn_train.pre_post": np.asarray([500, 100]), "rpn_train.cls_loss": loss["cls_loss"].detach().numpy(),
                "rpn_train.reg_loss": loss["reg_loss"].detach().numpy()})
    for b, p_ in enumerate(props):
        out[f"rpn_train.props{b}"] = p_.detach().numpy()
    # ---- RPN.forward, eval mode, on the tie-free inputs of rpn_stages.npz (rpn.py:46-85)
    rs = np.load(os.path.join(HERE, "rpn_stages.npz"))
    info = tuple(int(v) for v in rs["info"])
    fsz = [tuple(int(v) for v in hw) for hw in rs["fsz"]]
    head = _FixedHead([torch.as_tensor(rs[f"cls{l}"]) for l in range(len(fsz))],
                      [torch.as_tensor(rs[f"reg{l}"]) for l in range(len(fsz))])
    params = dict(params, rpn_head=head, num_pre_nms=int(rs["num_pre_nms"]), num_post_nms=60,
                  proposal_min_size=float(rs["min_size"]))
    rpn = ref_rpn.RPN(params).eval()
    props, loss = rpn(info, [torch.zeros(info[0], 1, h, w) for h, w in fsz], None)
    assert loss is None
    out["rpn_eval.post_thr"] = np.asarray([60, 0.7])
    for b, p_ in enumerate(props):
        # the file's result = the transcription's filtered boxes at the oracle's keeps
        keep = orc.nms(rs[f"b{b}.filtered"], rs[f"b{b}.filtered_score"], 0.7)[:60]
        assert np.array_equal(p_.numpy(), rs[f"b{b}.filtered"][keep]), "rpn_stages transcription and rpn.py disagree"
        out[f"rpn_eval.props{b}"] = p_.numpy()
    # ---- ROIHead.forward (eval: the file forces it, roi_head.py:90) with recorded predictor outputs: lines 127-135,
    # 167-229 of the file itself, incl. ROIPooling -> MultiScaleRoIAlign on a two-level pyramid
    g = torch.Generator().manual_seed(60221023)
    K, Cf = 7, 3
    info = (3, 256, 320)
    sizes = [(250, 300), (256, 320), (200, 310)]
    nprops = [150, 1, 333]
    feats = [torch.rand(3, Cf, 64, 80, generator=g), torch.rand(3, Cf, 32, 40, generator=g)]
    props = []
    for n in nprops:
        cx, cy = torch.rand(n, generator=g) * 320, torch.rand(n, generator=g) * 256
        w = 16 * 2 ** (torch.rand(n, generator=g) * 4)
        h = 16 * 2 ** (torch.rand(n, generator=g) * 4)
        props.append(torch.stack([(cx - w / 2).clamp(0, 320), (cy - h / 2).clamp(0, 256), (cx + w / 2).clamp(0, 320),
                                  (cy + h / 2).clamp(0, 256)], 1))
    N = sum(nprops)
    logits = torch.randn(N, K, generator=g) * 2.5
    regs = torch.randn(N, K * 4, generator=g) * 0.3
    regs[5, 10] = 9.0                                          # above the dw clamp
    regs[7, 8:12] = torch.tensor([0.0, 0.0, -40.0, -40.0])     # collapses: dropped by the small-box filter
    seen = {}

    class Flat(torch.nn.Module):
        def forward(self, x):
            seen["roi_features"] = x
            return x.flatten(1)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return logits, regs

    hp = dict(pos_threshold=0.5, neg_threshold=0.5, num_sample=128, pos_fraction=0.25,
              regression_weights=[10., 10., 5., 5.], score_thresh=0.05, proposal_min_size=1e-3, nms_threshold=0.5,
              num_post_nms=40, roi_pooling=ref_roi_pooling.ROIPooling(['1', '2'], [7, 7], 2), mlp_head=Flat(),
              predictor=Fixed())
    roi_head = ref_roi_head.ROIHead(hp)
    result, roi_loss = roi_head(props, feats, info, None, sizes)
    assert roi_loss == {} and len(result) == 3 and sum(len(r["score"]) for r in result) > 20
    out.update({"roi_eval.info": np.asarray(info), "roi_eval.sizes": np.asarray(sizes), "roi_eval.K": np.asarray(K),
                "roi_eval.logits": logits.numpy(), "roi_eval.regs": regs.numpy(),
                "roi_eval.params": np.asarray([0.05, 1e-3, 0.5, 40], np.float64),
                "roi_eval.roi_features": seen["roi_features"].numpy()})
    for l, f_ in enumerate(feats):
        out[f"roi_eval.feat{l}"] = f_.numpy()
    for i, (p_, r) in enumerate(zip(props, result)):
        out[f"roi_eval.props{i}"] = p_.numpy()
        out[f"roi_eval.b{i}.proposal"] = r["proposal"].numpy()
        out[f"roi_eval.b{i}.score"] = r["score"].numpy()
        out[f"roi_eval.b{i}.label"] = r["label"].numpy()
    save("ref_modules.npz", **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: goldens can only be generated in the build container")
    gen_anchors()
    gen_targets()
    gen_decode()
    gen_decode_grad()
    gen_losses()
    gen_transform()
    gen_sampling()
    gen_rpn_stages()
    gen_train_sampling()
    gen_reference_modules()
