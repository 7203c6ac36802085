"""Hungarian matcher + set criterion on the GPU.

Mirrors /root/reference/ModelComponents/losses_and_metrics.py: default weights (8-11),
MatchingLoss (75-161), MatchingMetric (164-192), MatchingMask (195-212), CostArray (215-225),
MatchingAssignment (228-251).  The reference builds [B,M,N,C] broadcast tensors and hops to the
host for scipy; here one kernel builds the [B,M,N] cost matrix, one workgroup per image solves
the assignment exactly in LDS (csrc/matcher.hip), and one kernel reduces the masked losses and
writes the sparse gradient - no host synchronisation anywhere.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K
from .engine import Layer, current_tape

DEFAULT_CATEGORY_WEIGHT = 1000.0
DEFAULT_BOX_WEIGHT = 1.0
DEFAULT_ATTRIBUTE_WEIGHT = 100.0
DEFAULT_EXIST_WEIGHT = 100.0


class MatchingAssignment(Layer):
    """Bipartite assignment.  Returns the match vector (int32 [B,M], prediction index per object
    or -1); ``mask()`` expands it to the reference's {0,1} mask [B,M,N]."""

    def __init__(self, name="MatchingAssignment", **kwargs):
        super().__init__(name=name, **kwargs)
        self.built = True

    def call(self, cost_array, num_objects, training=False):
        return K.lsa(cost_array, num_objects)

    def __call__(self, cost_array, num_objects, **kw):
        return self.call(cost_array, num_objects)

    @staticmethod
    def mask(match, num_preds):
        return K.match_to_mask(match, num_preds)

    @staticmethod
    def validate(match, num_objects, num_preds):
        """scipy raises ValueError for NaN/-inf or infeasible matrices; the kernel leaves such rows
        at -1.  This check synchronises, so the training loop only calls it when asked to."""
        m = match.cpu().numpy()
        n = num_objects.cpu().numpy()
        for b in range(m.shape[0]):
            want = min(int(n[b]), num_preds, m.shape[1])      # the kernel clamps num_objects to the M padded rows
            if (m[b] >= 0).sum() != want:
                raise ValueError("cost matrix is infeasible or contains invalid numeric entries")


class MatchingMask(Layer):
    def __init__(self, name="MatchingMask", **kwargs):
        super().__init__(name=name, **kwargs)
        self.MatchingAssignment = MatchingAssignment()
        self.built = True

    def __call__(self, inputs, **kw):
        matching_costs, num_objects = inputs
        match = self.MatchingAssignment(matching_costs, num_objects)
        # the reference also returns `assigned_predictions` (max of the mask over objects, 206-207); the
        # loss kernel derives it from the match vector, so only the mask and the vector are returned
        return MatchingAssignment.mask(match, matching_costs.shape[-1]), match


class MatchingLoss(Layer):
    """``call([y_true, y_pred])`` with y_true = [category_ids int32 [B,M], attribute multi-hot
    [B,M,A], bbox [B,M,4], num_objects int32 [B]] and y_pred = [cat [B,N,C], att [B,N,A], box
    [B,N,4]].  Returns (losses, metrics) exactly like the reference: losses = [total, category,
    attribute, box, exist] (each [B]) and metrics = [masked_iou [B]].  When a Tape is recording,
    the gradient of sum_b total_b w.r.t. the three prediction tensors is registered on it."""

    def __init__(self, name="MatchingLoss", category_weight=None, box_weight=None, attribute_weight=None,
                 exist_weight=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.category_weight = float(DEFAULT_CATEGORY_WEIGHT if category_weight is None else category_weight)
        self.box_weight = float(DEFAULT_BOX_WEIGHT if box_weight is None else box_weight)
        self.attribute_weight = float(DEFAULT_ATTRIBUTE_WEIGHT if attribute_weight is None else attribute_weight)
        self.exist_weight = float(DEFAULT_EXIST_WEIGHT if exist_weight is None else exist_weight)
        self.MatchingMask = MatchingMask()
        self.loss_scale = 1.0            # 1/num_replicas under data parallelism
        self.last_match: Optional[torch.Tensor] = None
        self.last_cost: Optional[torch.Tensor] = None
        self.last_num_objects: Optional[torch.Tensor] = None
        self.built = True

    def call(self, inputs, training=False):
        y_true, y_pred = inputs
        category, attribute, bbox, num_objects = y_true
        cat_preds, attribute_preds, box_preds = y_pred
        B, N, Cc = cat_preds.shape
        M = category.shape[1]
        A = attribute_preds.shape[-1]
        d = K.loss_desc(B, M, N, Cc, A, self.category_weight, self.attribute_weight, self.box_weight, self.exist_weight)
        cost = K.cost_matrix(d, cat_preds, attribute_preds, box_preds, category, attribute, bbox, num_objects)
        match = K.lsa(cost, num_objects)
        tape = current_tape()
        losses, d_cat, d_att, d_box = K.set_loss(d, cat_preds, attribute_preds, box_preds, category, attribute, bbox,
                                                 num_objects, match, loss_scale=self.loss_scale, want_grads=tape is not None)
        self.last_match, self.last_cost, self.last_num_objects = match, cost, num_objects
        if tape is not None:
            tape.record([losses], [cat_preds, attribute_preds, box_preds], lambda g: (d_cat, d_att, d_box))
        total, cat_l, att_l, box_l, exist_l, iou = (losses[i] for i in range(6))
        self._losses_tensor = losses
        return [total, cat_l, att_l, box_l, exist_l], [iou]

    def __call__(self, inputs, training=False, **kw):
        return self.call(inputs, training=training)


class MatchingMetric(Layer):
    """IoU of matched pairs (176-192) - computed inside MatchingLoss's kernel; kept for API parity."""

    def __init__(self, name="MatchingMetric", **kwargs):
        super().__init__(name=name, **kwargs)
        self.built = True
