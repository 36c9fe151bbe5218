"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's perceptual loss
(model/model_utils/perceptual_loss.py:46-77) in plain torch.nn.functional, on a caller-supplied state dict with the
reference's keys (``slice1.0.weight`` ... ``slice4.21.bias``).

PARITY UNPINNED with respect to the weights: the reference builds the network from torchvision's ``vgg16()`` and loads
``model/ckp-399.pth`` (:17-24); neither torchvision nor that checkpoint exists in this image or in the reference repository,
so no output of the reference's own module can be produced here.  What this file pins is the ARITHMETIC on given weights:
the slicing of a volume into per-channel (batch, z) images repeated to three channels (:48-52), the layer sequence of
vgg16.features[0:23] (torchvision 0.11.3: conv3x3/ReLU pairs 64-64-M-128-128-M-256-256-256-M-512-512-512), the mean of the
four MSEs (:67-69) and the mean over channels (:71-76).
"""
import torch
import torch.nn.functional as F

# index inside vgg16.features -> slice; 'M' = 2x2 max pool (features[4], [9], [16])
LAYERS = [(1, 0), (1, 2), 'M', (2, 5), (2, 7), 'M', (3, 10), (3, 12), (3, 14), 'M', (4, 17), (4, 19), (4, 21)]
SLICE_END = {2: 0, 7: 1, 14: 2, 21: 3}


def forward_one_view(sd, X):
    """perceptual_loss.py:46-63: X [bs, 1, z, y, x] -> the four feature maps"""
    X = X.permute(0, 2, 1, 3, 4)
    X = X.reshape(-1, *X.shape[2:])
    if X.size(1) == 1:
        X = X.repeat(1, 3, 1, 1)
    outs, h = [], X
    for item in LAYERS:
        if item == 'M':
            h = F.max_pool2d(h, 2, 2)
            continue
        s, idx = item
        h = F.relu(F.conv2d(h, sd[f'slice{s}.{idx}.weight'], sd[f'slice{s}.{idx}.bias'], padding=1))
        if idx in SLICE_END:
            outs.append(h)
    return outs


def perceptual_loss(sd, X1, X2):
    """perceptual_loss.py:65-76"""
    ch = X1.shape[1]
    loss = 0.0
    for idx in range(ch):
        a, b = forward_one_view(sd, X1[:, idx:idx + 1]), forward_one_view(sd, X2[:, idx:idx + 1])
        loss = loss + torch.mean(torch.as_tensor([F.mse_loss(a[i], b[i]) for i in range(4)]))
    return loss / ch
