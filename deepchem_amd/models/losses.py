"""The two losses of the GraphConv path behind the reference's ``Loss`` interface
(deepchem/models/losses.py: ``L2Loss`` :76-94, ``SoftmaxCrossEntropy`` :236-259, shape rule :1522-1543).

A loss object hands out a per-element criterion through ``_create_pytorch_loss()``; weighting and the mean
belong to the caller.  ``_gcmi_kind`` is the criterion's number in ``gcmi_loss_fwd_bwd`` (include/gcmi.h), which
``_StandardLoss`` uses in place of criterion + weighting + mean when the tensors live on the GPU."""
import torch
import torch.nn.functional as F


def _make_pytorch_shapes_consistent(output, labels):
    """Give ``output`` and ``labels`` the same rank: the one of lower rank gains trailing unit axes, allowed
    only where the other's extra axes are all of extent 1.  Anything else is the reference's ValueError."""
    ro, rl = output.dim(), labels.dim()
    if ro == rl:
        return output, labels
    longer, shorter = (output, labels) if ro > rl else (labels, output)
    extra = tuple(longer.shape[shorter.dim():])
    if any(e != 1 for e in extra):
        raise ValueError("Incompatible shapes for outputs and labels: %s versus %s" %
                         (str(tuple(output.shape)), str(tuple(labels.shape))))
    shorter = shorter.reshape(tuple(shorter.shape) + extra)
    return (output, shorter) if ro > rl else (shorter, labels)


def _squared_difference(output, labels):
    return F.mse_loss(output, labels, reduction='none')


def _label_weighted_log_softmax(output, labels):
    return -(labels * F.log_softmax(output, dim=-1)).sum(dim=-1)


class Loss(object):
    """Base of the loss objects.  Subclasses name a criterion ``f(output, labels) -> per-element loss``."""

    _gcmi_kind = None
    _criterion = None

    def _create_pytorch_loss(self):
        criterion = type(self)._criterion
        if criterion is None:
            raise NotImplementedError("Subclasses must implement this")

        def loss(output, labels):
            return criterion(*_make_pytorch_shapes_consistent(output, labels))

        return loss


class L2Loss(Loss):
    """(output - labels)^2, element by element."""

    _gcmi_kind = 1
    _criterion = staticmethod(_squared_difference)


class SoftmaxCrossEntropy(Loss):
    """Cross entropy of label probabilities against softmax(logits) over the last axis; that axis is summed away."""

    _gcmi_kind = 0
    _criterion = staticmethod(_label_weighted_log_softmax)
