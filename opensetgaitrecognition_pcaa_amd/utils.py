"""Host-side helpers of the PCAA path with the reference's names
(reference ``utils.py``: ``SeqChamferLoss`` :88-132, ``save_model`` :160-161,
``sample_distant_points`` :216-251, ``openness``)."""
import math

import numpy as np
import torch

from . import functional as F_hip


class SeqChamferLoss(torch.nn.Module):
    """forward(preds[B,C,T,N], gts[B,C,T,N], avg_out=True): sequence Chamfer
    distance over all C features; scalar (mean over B,T) or per-sequence [B].
    Fused forward + analytic backward on the HIP path; gradient flows to
    ``preds`` only, as in the reference's call (PCAA_ablation.py:994)."""

    def __init__(self):
        super().__init__()

    def forward(self, preds, gts, avg_out=True):
        return F_hip.seq_chamfer_loss(preds, gts, avg_out)


def save_model(_model: torch.nn.Module, _path):
    """``torch.save(model.state_dict(), path)`` (reference utils.py:160-161).  The trainer re-points parameters
    into its flat buffers (train.FlatBuffer): saved as they are, every tensor would drag the whole 0.6 GB
    storage into each checkpoint file, so each entry is saved as a compact copy -- same keys, shapes, values."""
    torch.save({k: v.detach().clone() for k, v in _model.state_dict().items()}, _path)


def sample_distant_points(dimension, n, min_dist, sphere_radius, seed=42, verbose=False):
    """Prior centroids: 10 000 points on a sphere in R^dimension, farthest-point
    sample n of them, repeat (drawing a new start index from the same
    generator) until the smallest pairwise distance reaches ``min_dist``.
    Host-side numpy, float64 result ``[n, dimension]`` -- identical draws and
    arithmetic to the reference (utils.py:216-251), golden-pinned for
    n in {2,4,6,8}."""
    gen = np.random.default_rng(seed)
    count = 10000
    cloud = gen.standard_normal(size=(dimension, count))
    cloud /= np.linalg.norm(cloud, axis=0)
    cloud = cloud * sphere_radius
    rows = cloud.T
    reached = 0
    picked_cols = None
    while reached < min_dist:
        if verbose:
            print(reached)
        current = gen.integers(low=0, high=count)
        order = [current]
        closest = np.ones(count) * 1e10
        for _ in range(n - 1):
            closest = np.minimum(closest, np.sum((rows - rows[current]) ** 2, axis=1))
            current = np.argmax(closest)
            order.append(current)
        picked_cols = cloud[:, order]
        pts = torch.tensor(picked_cols.T)
        pair = torch.cdist(pts, pts)
        reached = torch.min(pair[pair > 0])
    return torch.tensor(picked_cols.T)


def openness(n_train_classes, n_test_classes):
    return 1 - math.sqrt(2 * n_train_classes / (n_train_classes + n_test_classes))


def CG_kl_divergence(mu, logvar, mu_k):
    """KL( N(mu, exp(logvar)) || N(mu_k, I) ) averaged over the batch (reference utils.py:72-85, equation (6) of
    "Conditional Gaussian Distribution Learning for Open Set Recognition"): [B,32] device tensors -> scalar."""
    from . import functional as F_hip
    return F_hip.cg_kl_divergence(mu, logvar, mu_k)        # pcaa_orced_kl (forward + its three gradients); raises on CPU tensors
