"""TEST INFRASTRUCTURE ONLY (CPU oracle; never imported by the product path).

Restatement of the reference's batched-weight layers as the per-member loops the reference runs, in stock torch, dtype
generic (fp64 in the tests).  Pinned by the layer / train-step fixtures the REAL reference produced
(tests/golden/conv_*.npz, lenet_train_step_*.npz, resnet_train_step_f64.npz)."""
import torch
import torch.nn.functional as F


def batched_batch_norm(x, weight, bias, run_mean, run_var, training, momentum, eps):
    """hyper/layers/norm.py:206-222: F.batch_norm per member, stacked.  x [M,N,C,*]; weight / bias [M,C] | None;
    run_mean / run_var [M,C] | None are updated IN PLACE in training mode, as F.batch_norm does."""
    outs = []
    for m in range(x.shape[0]):
        outs.append(F.batch_norm(x[m], None if run_mean is None else run_mean[m], None if run_var is None else run_var[m],
                                 None if weight is None else weight[m], None if bias is None else bias[m],
                                 training, momentum, eps))
    return torch.stack(outs)


def batched_conv2d(x, weight, bias, stride=1, padding=0, dilation=1):
    """hyper/layers/conv.py:434-502: F.conv2d per member (x [N,C,H,W] shared or [M,N,C,H,W]), stacked."""
    outs = []
    for m in range(weight.shape[0]):
        outs.append(F.conv2d(x if x.ndim == 4 else x[m], weight[m], None if bias is None else bias[m], stride, padding, dilation))
    return torch.stack(outs)
