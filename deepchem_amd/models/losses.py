"""The two losses of the GraphConv path (deepchem/models/losses.py), as Loss
objects with the reference's interface (``_create_pytorch_loss()`` returns the
per-sample/per-task criterion).  ``_gcmi_kind`` lets ``_StandardLoss`` replace
criterion + weighting + mean by the fused HIP kernel (gcmi_loss_fwd_bwd)."""
import torch


class Loss(object):
    """A per-sample (per-task) loss; weighting and averaging are done by the caller."""

    _gcmi_kind = None

    def _create_pytorch_loss(self):
        raise NotImplementedError("Subclasses must implement this")


def _make_pytorch_shapes_consistent(output, labels):
    """Pad the shorter shape with trailing 1s (deepchem/models/losses.py:1522-1543)."""
    shape1, shape2 = tuple(output.shape), tuple(labels.shape)
    len1, len2 = len(shape1), len(shape2)
    if len1 == len2:
        return (output, labels)
    if len1 > len2 and all(i == 1 for i in shape1[len2:]):
        for _ in range(len1 - len2):
            labels = torch.unsqueeze(labels, -1)
        return (output, labels)
    if len2 > len1 and all(i == 1 for i in shape2[len1:]):
        for _ in range(len2 - len1):
            output = torch.unsqueeze(output, -1)
        return (output, labels)
    raise ValueError("Incompatible shapes for outputs and labels: %s versus %s" %
                     (str(shape1), str(shape2)))


class L2Loss(Loss):
    """Squared difference (deepchem/models/losses.py:76-94)."""

    _gcmi_kind = 1

    def _create_pytorch_loss(self):

        def loss(output, labels):
            output, labels = _make_pytorch_shapes_consistent(output, labels)
            return torch.nn.functional.mse_loss(output, labels, reduction='none')

        return loss


class SoftmaxCrossEntropy(Loss):
    """Cross entropy between label probabilities and softmax(logits) over the
    last axis (deepchem/models/losses.py:236-259)."""

    _gcmi_kind = 0

    def _create_pytorch_loss(self):
        ls = torch.nn.LogSoftmax(dim=-1)

        def loss(output, labels):
            output, labels = _make_pytorch_shapes_consistent(output, labels)
            return -torch.sum(labels * ls(output), dim=-1)

        return loss
