"""Drop-in PCAA modules backed by the HIP path.

Same constructor signatures, sub-module structure and ``state_dict`` keys as
the reference's ``models.py`` (``PointNetModule`` :6-34, ``DilTempConv1d``
:37-79, ``PointNetBlock`` :82-105, ``TemporalConvolutionBlock`` :108-160,
``CGEncoder`` :232-292, ``CGDecoder`` :340-385, ``CGDiscriminator`` :405-421,
``GaussianMeanLearner`` :424-443), so reference checkpoints (``_E/_G/_D.pt``)
load unchanged and the reference's training loops can construct them.

The ``torch.nn`` layer objects created here only OWN parameters/buffers (and
give the reference's default initialisation, drawn in the same order from
torch's global RNG); none of their ``forward``s is ever called.  All
arithmetic goes through :mod:`.functional` -> C-ABI -> HIP kernels.  There is
no CPU fallback: calling a module without the extension or on a CPU tensor
raises.
"""
import numpy as np
import torch

from . import constants
from . import functional as F_hip

_ELU = torch.nn.ELU


def _check_elu(act):
    if act is not None and not isinstance(act, torch.nn.ELU):
        raise NotImplementedError(
            "the HIP path implements the reference's ELU(alpha=1) activation only")
    if act is not None and (act.alpha != 1.0):
        raise NotImplementedError("ELU alpha must be 1.0")


class PointNetModule(torch.nn.Module):
    """Per-point linear (Conv2d 1x1) + BatchNorm2d + ELU on ``[B,Cin,T,N]``."""

    def __init__(self, in_chs, out_chs, activation=None):
        super().__init__()
        _check_elu(activation)
        self.module = torch.nn.Sequential(
            torch.nn.Conv2d(in_chs, out_chs, (1, 1), stride=1, padding="valid", dilation=1),
            torch.nn.BatchNorm2d(num_features=out_chs),
            activation if activation is not None else _ELU(),
        )

    def forward(self, x):
        return F_hip.pointnet_stack(x, [self], self.training)


class DilTempConv1d(torch.nn.Module):
    """Causal dilated Conv1d (pad 2d both sides, drop the last 2d outputs) +
    BatchNorm1d + ELU on ``[B,Cin,T]``."""

    def __init__(self, in_chs, out_chs, dilation, kernel_size=3, stride=1, use_bias=True,
                 activation=None):
        super().__init__()
        _check_elu(activation)
        if kernel_size != 3 or stride != 1:
            raise NotImplementedError("HIP path: kernel_size=3, stride=1 (the only use in the reference)")
        if out_chs % 4 != 0 or 256 % (out_chs // 4) != 0:
            # the BatchNorm / ELU passes move 16-byte quads of channels, a power-of-two number of them per 256-thread
            # workgroup (every width of the reference -- 16 ... 512, constants.py:37 -- qualifies)
            raise NotImplementedError("HIP path: DilTempConv1d out_chs must be a multiple of 4 and a power of two times 4 up "
                                      f"to 1024 (got {out_chs})")
        self.dilation = int(dilation)
        self.padding = int(np.floor((kernel_size - 1) * dilation))
        # bias=True regardless of use_bias, as in the reference (models.py:67)
        self.conv1d = torch.nn.Conv1d(in_chs, out_chs, kernel_size=kernel_size, stride=stride,
                                      padding=self.padding, dilation=dilation, bias=True)
        self.activation = activation if activation is not None else _ELU()
        self.batch_norm = torch.nn.BatchNorm1d(out_chs)

    def forward(self, x):
        return F_hip.dtc_stack(x, [self], self.training)


class PointNetBlock(torch.nn.Module):
    def __init__(self):
        super().__init__()
        d = constants.POINTNET_OUT_DIM
        self.pointnet1 = PointNetModule(in_chs=constants.NFEATURES, out_chs=d // 2)
        self.pointnet2 = PointNetModule(in_chs=d // 2, out_chs=d // 2)
        self.pointnet3 = PointNetModule(in_chs=d // 2, out_chs=d)
        self.pointnet4 = PointNetModule(in_chs=d, out_chs=d)

    def layers(self):
        return [self.pointnet1, self.pointnet2, self.pointnet3, self.pointnet4]

    def forward(self, x):
        return F_hip.pointnet_stack(x, self.layers(), self.training)


class TemporalConvolutionBlock(torch.nn.Module):
    DILATIONS = (1, 2, 4, 1, 2, 4)

    def __init__(self):
        super().__init__()
        chans = [constants.POINTNET_OUT_DIM] + list(constants.DTC_FILTERS)
        for i, d in enumerate(self.DILATIONS):
            setattr(self, f"dtc{i + 1}",
                    DilTempConv1d(in_chs=chans[i], out_chs=chans[i + 1], dilation=d, kernel_size=3))

    def layers(self):
        return [getattr(self, f"dtc{i}") for i in range(1, 7)]

    def forward(self, x):
        return F_hip.dtc_stack(x, self.layers(), self.training)


class CGEncoder(torch.nn.Module):
    """forward(x[B,C,T,N]) -> (out_classes[B,K], sup_fv[B,32])."""

    def __init__(self, n_out_labels, nmax_points=constants.NMAX, use_projection_head=False):
        super().__init__()
        self.use_projection_head = use_projection_head
        self.nmax_points = nmax_points
        self.pc_block = PointNetBlock()
        # parameter-free pooling layers kept for attribute parity (models.py:242-249)
        self.glob_avg_pool1 = torch.nn.AvgPool2d(kernel_size=(1, nmax_points))
        self.tc_block = TemporalConvolutionBlock()
        self.glob_avg_pool2 = torch.nn.AvgPool1d(kernel_size=constants.NSTEPS)
        lat = constants.SUP_LATENT_DIM
        self.MLP_sup1 = torch.nn.Sequential(
            torch.nn.Linear(in_features=constants.DTC_FILTERS[-1], out_features=lat), _ELU())
        head_out = lat // 2 if use_projection_head else lat
        if use_projection_head:
            self.MLP_head = torch.nn.Sequential(
                torch.nn.Linear(in_features=lat, out_features=head_out), _ELU())
        self.MLP_sup2 = torch.nn.Sequential(
            torch.nn.Linear(in_features=head_out, out_features=n_out_labels), _ELU())

    def forward(self, x):
        return F_hip.cg_encoder(self, x)


class CGDecoder(torch.nn.Module):
    """forward(z[B,input_dim]) -> [B,C,T,N]; bn1..bn4 are registered (they are
    in the reference's state_dict, models.py:353-368) and never used."""

    def __init__(self, input_dim=constants.SUP_LATENT_DIM, nmax_points=constants.NMAX):
        super().__init__()
        S = constants.NSTEPS * constants.NFEATURES * nmax_points
        self.decoder_mlp_size = S
        self.nmax_points = nmax_points
        self.n_features = constants.NFEATURES
        self.n_steps = constants.NSTEPS
        self.activation = _ELU()
        widths = [input_dim, S // 16, S // 8, S // 4, S // 2, S]
        for i in range(5):
            setattr(self, f"dense{i + 1}",
                    torch.nn.Linear(in_features=widths[i], out_features=widths[i + 1]))
            if i < 4:
                setattr(self, f"bn{i + 1}", torch.nn.BatchNorm1d(widths[i + 1]))

    def dense_layers(self):
        return [getattr(self, f"dense{i}") for i in range(1, 6)]

    def forward(self, x):
        return F_hip.cg_decoder(self, x)


class CGDiscriminator(torch.nn.Module):
    """forward(x[B,32], label[B,K]) -> [B,1] (critic, no sigmoid)."""

    def __init__(self, n_in_labels):
        super().__init__()
        self.n_in_labels = n_in_labels
        self.model = torch.nn.Sequential(
            torch.nn.Linear(constants.SUP_LATENT_DIM + n_in_labels, 64, bias=True), _ELU(),
            torch.nn.Linear(64, 32, bias=True), _ELU(),
            torch.nn.Linear(32, 1, bias=True),
        )

    def forward(self, x, label):
        return F_hip.cg_discriminator(self, x, label)


class GaussianMeanLearner(torch.nn.Module):
    """K->16->32->64->32 with BN1d+ELU (ablation V1 only, models.py:424-443)."""

    def __init__(self, n_in_labels):
        super().__init__()
        self.model = torch.nn.Sequential(
            torch.nn.Linear(n_in_labels, 16, bias=True), torch.nn.BatchNorm1d(16), _ELU(),
            torch.nn.Linear(16, 32, bias=True), torch.nn.BatchNorm1d(32), _ELU(),
            torch.nn.Linear(32, 64, bias=True), torch.nn.BatchNorm1d(64), _ELU(),
            torch.nn.Linear(64, constants.SUP_LATENT_DIM, bias=True),
        )

    def forward(self, x):
        return F_hip.gaussian_mean_learner(self, x)


class ORCEDEncoder(torch.nn.Module):
    """OR-CED baseline encoder (reference models.py:446-505): the CGEncoder trunk, then ``MLP_mu`` / ``MLP_logvar``
    (Linear 512 -> 32, no activation), the reparametrisation ``sup_fv = mu + eps * exp(0.5 logvar)`` with
    ``eps = torch.randn_like(logvar)`` drawn in BOTH train and eval mode, and ``MLP_classification`` (Linear 32 -> K,
    no activation).  forward(x[B,C,T,N]) -> (out_classes, sup_fv, vae_mu, vae_logvar).  The trunk runs on the PointNet /
    temporal-block kernels (functional.encoder_trunk), the three heads and the reparametrisation on
    pcaa_orced_heads_fwd / _bwd (round 3; the normal draw itself is torch's device generator, as in the reference)."""

    def __init__(self, n_out_labels, nmax_points=None):
        super().__init__()
        self.nmax_points = constants.NMAX if nmax_points is None else nmax_points   # the reference hard-wires constants.NMAX
        self.pc_block = PointNetBlock()
        self.glob_avg_pool1 = torch.nn.AvgPool2d(kernel_size=(1, self.nmax_points))
        self.tc_block = TemporalConvolutionBlock()
        self.glob_avg_pool2 = torch.nn.AvgPool1d(kernel_size=constants.NSTEPS)
        lat = constants.SUP_LATENT_DIM
        self.MLP_mu = torch.nn.Sequential(torch.nn.Linear(constants.DTC_FILTERS[-1], lat))
        self.MLP_logvar = torch.nn.Sequential(torch.nn.Linear(constants.DTC_FILTERS[-1], lat))
        self.MLP_classification = torch.nn.Sequential(torch.nn.Linear(lat, n_out_labels))

    def forward(self, x):
        x4 = F_hip.encoder_trunk(self, x)
        # the reference's draw (models.py:497: torch.randn_like(vae_logvar), [B, 32] from the device generator, in train
        # AND eval mode); the three heads and the reparametrisation are one launch (csrc/orced.hip)
        eps = torch.randn_like(x4[:, :constants.SUP_LATENT_DIM])
        return F_hip.orced_heads(self, x4, eps)


class ORCEDDecoder(CGDecoder):
    """OR-CED decoder (reference models.py:508-545): the same five-Linear stack as CGDecoder fed by the 32-wide latent,
    bn1..bn4 registered and unused, widths from ``DEC_MLP_SIZE`` (= NSTEPS * NMAX * NFEATURES)."""

    def __init__(self, nmax_points=None):
        super().__init__(input_dim=constants.SUP_LATENT_DIM,
                         nmax_points=constants.NMAX if nmax_points is None else nmax_points)
