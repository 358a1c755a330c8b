"""RPN stage with the reference's interface (RPN/rpn.py:31-142).

`RPN(dict_params).forward(batch_image_info, list_features, list_targets)` returns
`(list_nms_proposals, rpn_loss)` exactly like the reference, but the whole proposal path of
RPN/rpn.py:54-85 runs as six batched kernel launches (flatten -> top-k -> decode/clip/filter ->
NMS mask -> NMS scan -> gather) instead of three per-image Python loops, and the anchor-target
path of RPN/rpn.py:105-112 is one fused IoU/label/encode launch for the batch.

`proposals_padded` is the sync-free form used by the benchmark and by ROIHead's batched path: it
returns the padded (B, num_post_nms, 4) tensor and the per-image counts without touching the host.
"""
import torch
from torch.nn import Module

from .. import ops


class RPN(Module):
    def __init__(self, dict_params):
        super().__init__()
        self.anchor_generator = dict_params['anchor_generator']
        self.rpn_head = dict_params['rpn_head']
        self.num_pre_nms = dict_params['num_pre_nms']
        self.num_post_nms = dict_params['num_post_nms']
        self.nms_threshold = dict_params['nms_threshold']
        # launches of the round-path NMS bounded by what the previous batches needed (same keep lists)
        self.nms_rounds = ops.NmsRounds()
        self.proposal_min_size = dict_params['proposal_min_size']
        self.pos_threshold = dict_params['pos_threshold']
        self.neg_threshold = dict_params['neg_threshold']
        self.num_sample = dict_params['num_sample']
        self.pos_fraction = dict_params['pos_fraction']
        # workgroups per image for the top-k (ops.topk_desc): 0 = the library's choice; set to 1 when the
        # proposal stages run beside RoIAlign on another stream (pipeline.ProposalRoIPipeline)
        self.topk_max_slices = 0
        self._sampled = None

    @property
    def last_sampled(self):
        """What the last training forward sampled (rpn.py:115-127): per image the random_index (negatives first)
        and the gathered cls / reg labels -- observable so that the sampling order can be checked.  Built on
        demand: the forward itself keeps only the batch's flat index."""
        if self._sampled is None:
            return None
        flat, sizes, labels, reg_labels = self._sampled
        A = labels.shape[1]
        out, at = [], 0
        for b, n in enumerate(sizes):
            idx = flat[at:at + n] - b * A
            out.append({"random_index": idx, "cls_label": labels[b][idx], "reg_label": reg_labels[b][idx]})
            at += n
        return out

    # -- anchors ---------------------------------------------------------------------------------
    def _flat_anchors(self, batch_image_info, list_features):
        """(A,4) anchors shared by the batch when the generator returns stride-0 expands (ours and
        the reference's do, anchor_generator.py:119); otherwise a contiguous (B,A,4)."""
        list_anchors = self.anchor_generator(batch_image_info, list_features)
        if all(a.stride(0) == 0 or a.shape[0] == 1 for a in list_anchors):
            return list_anchors, torch.cat([a[0].reshape(-1, 4) for a in list_anchors], dim=0)
        return list_anchors, torch.cat([a.flatten(1, 2) for a in list_anchors], dim=1).contiguous()

    # -- proposal half: RPN/rpn.py:54-85 ----------------------------------------------------------
    def proposals_from_head(self, batch_image_info, anchors, list_cls, list_regs):
        """-> (props (B,P,4) zero-padded, n_keep (B) int32) on device, no host sync."""
        A = sum(c.shape[1] * c.shape[2] * c.shape[3] for c in list_cls)
        k = min(self.num_pre_nms, A)                                           # rpn.py:65
        info = batch_image_info
        if ops.rpn_in_place_supported(list_cls, k, self.topk_max_slices):
            # rpn.py:56-57 never materialised, and rpn.py:65-85 as ONE library call: top-k and decode read the head's
            # NCHW outputs where they lie, NMS and the gather follow on the same stream, intermediates in one
            # workspace kept across steps
            # (one workspace per device and stream: calls on a stream are ordered behind each other, calls on two
            # streams are not, and would overwrite each other's intermediates in a shared one)
            cache = self.__dict__.setdefault("_path_ws", {})
            key = (anchors.device.index, torch.cuda.current_stream(anchors.device).cuda_stream)
            props, n_keep, cache[key] = ops.rpn_proposals_nchw(
                list_cls, list_regs, anchors, k, info[1], info[2], self.proposal_min_size, self.nms_threshold,
                self.num_post_nms, max_slices=self.topk_max_slices, rounds=self.nms_rounds,
                workspace=cache.get(key))
            return props, n_keep
        scores, deltas = ops.rpn_flatten(list_cls, list_regs)               # rpn.py:56-57
        top_val, top_idx = ops.topk_desc(scores, k, max_slices=self.topk_max_slices)   # rpn.py:66-67
        boxes, _, _, count = ops.decode_clip_filter(                       # rpn.py:68-69,75,81-83
            anchors, deltas, top_idx, top_val, info[1], info[2], self.proposal_min_size)
        keep, n_keep = ops.nms_sorted(boxes, count, self.nms_threshold, self.num_post_nms,  # rpn.py:84
                                      rounds=self.nms_rounds)
        props = ops.gather_boxes(boxes, keep, n_keep)                          # rpn.py:85
        return props, n_keep

    def proposals_padded(self, batch_image_info, list_features):
        _, anchors = self._flat_anchors(batch_image_info, list_features)
        list_cls, list_regs = self.rpn_head(list_features)
        return self.proposals_from_head(batch_image_info, anchors, list_cls, list_regs)

    # -- anchor-target half: RPN/rpn.py:88-133 ----------------------------------------------------
    def anchor_targets(self, anchors, list_targets, device):
        """Batched tagging_box + box_deviation (rpn.py:105-112) -> label (B,A), reg_label (B,A,4)."""
        B = len(list_targets)
        m = [len(t['box']) for t in list_targets]
        gt = torch.zeros((B, max(m), 4), dtype=torch.float32)
        for i, t in enumerate(list_targets):
            gt[i, :m[i]] = torch.tensor(t['box'], dtype=torch.float32).reshape(-1, 4)
        gt = gt.to(device)
        m_dev = torch.tensor(m, dtype=torch.int32, device=device)
        out = ops.iou_label_encode(anchors, gt, m_dev, self.neg_threshold, self.pos_threshold, None, None,
                                   want=("label", "dev"))
        return out["label"], out["dev"]

    def forward(self, batch_image_info, list_features, list_targets):
        batch_size = batch_image_info[0]
        device = list_features[0].device
        list_anchors, anchors = self._flat_anchors(batch_image_info, list_features)
        list_cls, list_regs = self.rpn_head(list_features)

        props, n_keep = self.proposals_from_head(batch_image_info, anchors, list_cls, list_regs)
        counts = n_keep.tolist()  # the reference's ragged list output needs the sizes on the host
        list_nms_proposals = [props[b, :counts[b]] for b in range(batch_size)]

        rpn_loss = None
        if self.training:
            # stock, differentiable views of the head outputs for the loss (rpn.py:90-98)
            scores = torch.cat([c.permute(0, 2, 3, 1).flatten(1, 3) for c in list_cls], dim=1)
            regs = torch.cat([r.permute(0, 2, 3, 1).flatten(1, 2).reshape(batch_size, -1, 4) for r in list_regs],
                             dim=1)
            labels, reg_labels = self.anchor_targets(anchors, list_targets, device)
            # rpn.py:115-122 for the whole batch (ops.sample_rows): one host read, the reference's randperm order
            flat_index, sizes = ops.sample_rows(torch.eq(labels, 1), torch.eq(labels, 0), self.num_sample,
                                                self.pos_fraction)
            self._sampled = (flat_index, sizes, labels, reg_labels)
            # rpn.py:123-137: the four gathers, BCE-with-logits and smooth-L1 (Q7) are one launch, with
            # gradients to the dense head outputs (ops.RpnSampledLoss)
            cls_loss, reg_loss = ops.RpnSampledLoss.apply(scores, regs, labels, reg_labels, flat_index)
            rpn_loss = {"cls_loss": cls_loss, "reg_loss": reg_loss}
        return list_nms_proposals, rpn_loss
