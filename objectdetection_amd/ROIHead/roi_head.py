"""ROIHead with the reference's interface (ROIHead/roi_head.py:42-231).

Hot-path parts run on the gfx950 kernels: proposal-target creation (roi_head.py:98-125: fused
IoU / label / encode for the whole batch), RoI pooling (roi_head.py:127-130) and the eval
post-process (roi_head.py:167-229: decode, clip, score / size / label filters, class-agnostic
NMS).  The dense heads (mlp_head, predictor) and the losses stay stock PyTorch.

Quirk Q8: the reference calls self.train(mode=False) at roi_head.py:90, so its training branch is
dead code.  params['force_eval'] (default True) reproduces that; pass False to run the training
branch as written.
"""
import torch
import torch.nn.functional as F
from torch.nn import Module

from .. import ops
from ..utils import remove_small_box


class ROIHead(Module):
    def __init__(self, params):
        super().__init__()
        self.roi_pooling = params['roi_pooling']
        self.mlp_head = params['mlp_head']
        self.predictor = params['predictor']
        self.pos_threshold = params['pos_threshold']
        self.neg_threshold = params['neg_threshold']
        self.num_sample = params['num_sample']
        self.pos_fraction = params['pos_fraction']
        self.regression_weights = params['regression_weights']
        self.score_thresh = params['score_thresh']
        self.proposal_min_size = params['proposal_min_size']
        self.nms_threshold = params['nms_threshold']
        self.num_post_nms = params['num_post_nms']
        self.force_eval = params.get('force_eval', True)
        self._sampled = None

    @property
    def last_sampled(self):
        """What the last training forward sampled (roi_head.py:113-124): per image the random_index (negatives
        first) and the gathered labels / encodings / proposals -- observable for the sampling-order tests."""
        if self._sampled is None:
            return None
        flat, sizes, P, cls_label, reg_label, list_proposals = self._sampled
        out, at = [], 0
        for b, n in enumerate(sizes):
            out.append({"random_index": flat[at:at + n] - b * P, "cls_label": cls_label[at:at + n],
                        "reg_label": reg_label[at:at + n], "proposal": list_proposals[b]})
            at += n
        return out

    # -- proposal targets: roi_head.py:99-110, batched -------------------------------------------
    def _proposal_targets_padded(self, list_rpn_proposals, list_targets, device):
        """-> props (B,P,4) zero-padded, n (B) rows per image (device int64), cls_label (B,P), reg_label (B,P,4)."""
        B = len(list_rpn_proposals)
        P = max(p.shape[0] for p in list_rpn_proposals)
        m = [len(t['box']) for t in list_targets]
        M = max(m)
        props = torch.zeros((B, P, 4), dtype=torch.float32, device=device)
        gt = torch.zeros((B, M, 4), dtype=torch.float32)
        lab = torch.zeros((B, M), dtype=torch.float32)
        for i, (p, t) in enumerate(zip(list_rpn_proposals, list_targets)):
            props[i, :p.shape[0]] = p.detach()
            gt[i, :m[i]] = torch.as_tensor(t['box'], dtype=torch.float32).reshape(-1, 4)
            cls = torch.as_tensor(t['class'], dtype=torch.float32).reshape(-1)
            lab[i, :cls.shape[0]] = cls
        out = ops.iou_label_encode(props, gt.to(device), torch.tensor(m, dtype=torch.int32, device=device),
                                   self.neg_threshold, self.pos_threshold, lab.to(device), self.regression_weights,
                                   want=("label", "dev"))
        n = torch.tensor([p.shape[0] for p in list_rpn_proposals], device=device)
        return props, n, out["label"], out["dev"]

    def proposal_targets(self, list_rpn_proposals, list_targets, device):
        """-> list over images of (cls_label (P_i,), reg_label (P_i,4))."""
        _, _, label, dev_ = self._proposal_targets_padded(list_rpn_proposals, list_targets, device)
        return [(label[i, :p.shape[0]], dev_[i, :p.shape[0]]) for i, p in enumerate(list_rpn_proposals)]

    def forward(self, list_rpn_proposals, list_features, batch_image_info, list_targets, list_image_size):
        if self.force_eval:
            self.train(mode=False)  # roi_head.py:90 (Q8)
        image_height = batch_image_info[1]
        image_width = batch_image_info[2]
        device = list_rpn_proposals[0].device

        list_cls_label, list_reg_label, list_proposals = [], [], []
        if self.training:
            # roi_head.py:99-124 for the whole batch: one fused target launch, then ops.sample_rows (one host read,
            # the reference's randperm order per image) and three gathers -- no per-image device work
            props, n_rows, label, reg = self._proposal_targets_padded(list_rpn_proposals, list_targets, device)
            valid = torch.arange(props.shape[1], device=device)[None, :] < n_rows[:, None]
            flat_index, sizes = ops.sample_rows(torch.gt(label, 0) & valid, torch.eq(label, 0) & valid,
                                                self.num_sample, self.pos_fraction)
            cls_label = label.reshape(-1)[flat_index]
            reg_label = reg.reshape(-1, 4)[flat_index]
            list_proposals = list(torch.split(props.reshape(-1, 4)[flat_index], sizes))
            list_cls_label, list_reg_label = [cls_label], [reg_label]
            self._sampled = (flat_index, sizes, props.shape[1], cls_label, reg_label, list_proposals)
            list_rpn_proposals = list_proposals

        dict_feature = dict(zip([str(i + 1) for i in range(len(list_features))], list_features))
        roi_features = self.roi_pooling(dict_feature, list_rpn_proposals, [(image_height, image_width)])
        mlp_features = self.mlp_head(roi_features)
        predict_cls, predict_reg = self.predictor(mlp_features)
        predict_reg = predict_reg.view(predict_cls.size(0), -1, 4)

        roi_loss = {}
        result = []
        if self.training:
            cls_label = torch.cat(list_cls_label)
            reg_label = torch.cat(list_reg_label)
            # roi_head.py:144-152: cross-entropy, the positives' per-class rows and smooth-L1 in one launch
            cls_loss, reg_loss = ops.RoiLoss.apply(predict_cls, predict_reg, cls_label, reg_label)
            roi_loss = {"cls_loss": cls_loss, "reg_loss": reg_loss}
        else:
            # roi_head.py:167-229
            num_class = predict_cls.size(-1)
            predict_score = F.softmax(predict_cls, -1)
            # One launch decodes / clips / filters every (proposal, class) candidate of the batch
            # (:178-209), one sort + gather + NMS pass handles all images (:215), and the host reads two
            # small count vectors instead of synchronising four times per image.
            num_per_image = [p.size(0) for p in list_rpn_proposals]
            B = len(list_rpn_proposals)
            row_off = torch.tensor([0] + list(torch.tensor(num_per_image).cumsum(0).tolist()), dtype=torch.int32,
                                   device=device)
            sizes = torch.tensor([[float(s[0]), float(s[1])] for s in list_image_size], dtype=torch.float32,
                                 device=device)
            boxes, scores, labels, count = ops.roi_detect_filter(
                torch.cat(list_rpn_proposals), predict_reg.detach(), predict_score.detach(), row_off, sizes,
                max(num_per_image) if num_per_image else 0, self.score_thresh, self.proposal_min_size)
            counts = count.tolist()                                   # host read 1: candidates per image
            vmax = max(counts) if counts else 0
            if vmax > 0:
                valid = torch.arange(vmax, device=device)[None, :] < count[:, None]
                padded = torch.where(valid, scores[:, :vmax], torch.full_like(scores[:, :vmax], float("-inf")))
                _, order = ops.topk_desc(padded, vmax, want_values=False)          # descending score, ties by index
                sorted_boxes = ops.gather_boxes(boxes, order, count)
                keep, n_keep = ops.nms_sorted(sorted_boxes, count, self.nms_threshold, self.num_post_nms)
                n_kept = n_keep.tolist()                              # host read 2: detections per image
                # four batched gathers; the per-image lists are views of their results (keep rows beyond
                # n_keep are unspecified: clamped, then cut off by the views)
                top = max(max(n_kept), 1)
                index = torch.gather(order, 1, keep[:, :top].clamp(0, vmax - 1).long()).long()
                det_boxes = torch.gather(boxes, 1, index[:, :, None].expand(-1, -1, 4))
                det_scores = torch.gather(scores, 1, index)
                det_labels = torch.gather(labels, 1, index).long()
            for i in range(B):
                n = n_kept[i] if vmax > 0 else 0
                if n > 0:
                    result.append({'proposal': det_boxes[i, :n], 'score': det_scores[i, :n],
                                   'label': det_labels[i, :n]})
                else:
                    result.append({'proposal': torch.empty((0, 4), dtype=torch.float32, device=device),
                                   'score': torch.empty((0,), dtype=torch.float32, device=device),
                                   'label': torch.empty((0,), dtype=torch.int64, device=device)})
        return result, roi_loss
