"""PCAA training loops on the HIP path, with the reference's call surface:

* ``train_variant4(config, wandb_mode="online", proj_head_on_discriminator=False)``
  -- the paper's PCAA (reference ``PCAA_ablation.py:746-1122``)
* ``train_CGAAE(config)`` / ``train_variant2`` -- base loop without projection
  heads (``train_AAE.py:25-364``, ``PCAA_ablation.py:381-389``)
* ``train_variant1(config, wandb_mode="online")`` -- ablation with the ``GaussianMeanLearner``
  producing the prior centroids (``PCAA_ablation.py:28-378``)

Both drive :class:`PCAATrainer`, which owns the five modules, flat fp32
parameter / gradient / Adam buffers (one fused Adam launch per optimiser, one
RCCL all-reduce per optimiser under data parallelism) and runs one step as a
fixed sequence of HIP launches on the current stream with NO host
synchronisation: losses stay on the device until the caller reads them.
"""
import itertools
import math
import os
import pickle

import numpy as np
import torch

from . import constants
from . import dist as pdist
from . import functional as F_hip
from . import ops
from ._lib import ACT_ELU
from .models import CGDecoder, CGDiscriminator, CGEncoder, GaussianMeanLearner
from .utils import sample_distant_points, save_model

_ALIGN = 64  # floats; keeps every parameter view 256-B aligned inside the flat buffers


_SIDE_STREAMS = {}


# where the decoder's side-stream update is enqueued: "dec_bwd" (right after the decoder backward), "heads" (after the MLP
# heads' backward launch), "pointnet" (before the PointNet backward); re-measured each time the kernels around it change
_SIDE_ADAM_AT = os.environ.get("PCAA_SIDE_ADAM_AT", "dec_bwd")


def _side_streams(device):
    """(adam, critic, wgrad) streams of ``device``, created on first use"""
    if device.type != "cuda":
        return None, None, None
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        # Three streams beside the main one.  Two leaner layouts were measured on one box (round 3, ms/step single |
        # forced 1-rank RCCL): the critic branch sharing the weight-gradient stream 5.74 | 6.50, sharing the Adam stream
        # (both idle while the critic runs) 5.72 | 6.38, separate streams 5.67 | 6.10 -- although a data-parallel process
        # then has five streams (RCCL brings its own) on the device's four hardware queues and rocprofv3 shows the
        # critic on the main stream's queue.  PCAA_STREAM_LAYOUT keeps the other two selectable for re-measurement.
        side, small = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        layout = os.environ.get("PCAA_STREAM_LAYOUT", "separate")
        if layout == "critic_on_wgrad":
            _SIDE_STREAMS[key] = (side, small, small)
        elif layout == "critic_on_side":
            _SIDE_STREAMS[key] = (side, side, small)
        else:
            _SIDE_STREAMS[key] = (side, small, torch.cuda.Stream(device=device))
    return _SIDE_STREAMS[key]


class StepCount:
    """An optimizer step count on the device (int32) with the two bias-correction scalars Adam derives from it
    (``pcaa_adam_advance``), plus its host mirror."""

    def __init__(self, device):
        self.step = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.coef_dev = torch.zeros(2, dtype=torch.float32, device=device)

    def advance(self, lr, b1, b2):
        self.step += 1
        ops.adam_advance_(self.step_dev, self.coef_dev, lr, b1, b2)

    def set(self, step):
        self.step = int(step)
        self.step_dev.fill_(self.step)


class _CompressedWork:
    """An async all-reduce of the bf16 image of a gradient bucket: ``wait()`` (stream-side, like the work's own)
    is followed by widening the sum back into the fp32 bucket, once, on the stream of the first waiter."""

    def __init__(self, work, dst32, src16):
        self.work, self.dst32, self.src16, self.done = work, dst32, src16, False

    def wait(self, widen=True):
        """``widen=False``: the caller consumes the bf16 sum itself (``src16``, FlatBuffer.adam(g16=...)) and marks the
        work done; a later plain wait() then only waits."""
        self.work.wait()
        if not self.done and widen:
            self.dst32.copy_(self.src16)
            self.done = True


class FlatBuffer:
    """Parameters re-pointed into one contiguous fp32 buffer (+ matching
    gradient and Adam moment buffers)."""

    def __init__(self, named_params, device, padded_shapes=None, tail_multiple=None):
        """``padded_shapes``: optional {name: padded shape}: that parameter is stored zero-padded to the
        given shape and exposed as the strided ``[:n0, :n1]`` view of it (same values, same ``state_dict``);
        ``self.padded[name]`` holds the full (parameter, gradient) tensors for kernels that want whole tiles."""
        self.names, self.offsets, self.sizes, self.params = [], [], [], []
        padded_shapes = padded_shapes or {}
        total = 0
        for name, p in named_params:
            self.names.append(name)
            self.offsets.append(total)
            n = int(torch.Size(padded_shapes[name]).numel()) if name in padded_shapes else p.numel()
            self.sizes.append(n)
            self.params.append(p)
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        if tail_multiple is not None:
            # (start_name, multiple): zero tail padding so that the region from that parameter to the end of the
            # buffer splits evenly (sharded optimizer: world x chunks x 64-float pieces)
            start = self.offsets[self.names.index(tail_multiple[0])]
            total = start + (total - start + tail_multiple[1] - 1) // tail_multiple[1] * tail_multiple[1]
        self.total = total
        self.p = torch.zeros(total, dtype=torch.float32, device=device)
        self.g = torch.zeros(total, dtype=torch.float32, device=device)
        self.m = torch.zeros(total, dtype=torch.float32, device=device)
        self.v = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad_views = {}
        self.padded = {}
        with torch.no_grad():
            for name, p, o, n in zip(self.names, self.params, self.offsets, self.sizes):
                if name in padded_shapes:
                    shp = tuple(padded_shapes[name])
                    full_p, full_g = self.p[o:o + n].view(shp), self.g[o:o + n].view(shp)
                    window = tuple(slice(0, d) for d in p.shape)
                    full_p[window].copy_(p.detach())
                    p.data = full_p[window]
                    self.grad_views[name] = full_g[window]
                    self.padded[name] = (full_p, full_g)
                    continue
                self.p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.p[o:o + n].view(p.shape)
                self.grad_views[name] = self.g[o:o + n].view(p.shape)
        # The optimizer's step count also lives on the device (int32 + the two bias-correction scalars
        # derived from it by pcaa_adam_advance), so that a captured hipGraph of the train step replays
        # without per-step host arguments; ``step`` is the host mirror (replays bump it).
        self.count = StepCount(device)
        # Parameters that only receive a gradient on SUPERVISED steps (MLP_head / MLP_sup2, see
        # PCAATrainer._sup_range) keep their own count: torch.optim.Adam's ``step`` state is per parameter
        # and is not advanced while the parameter's ``.grad`` is None.
        self.sup_count = StepCount(device)

    # the main counter under its historical names
    @property
    def step(self):
        return self.count.step

    @step.setter
    def step(self, v):
        self.count.step = int(v)

    @property
    def step_dev(self):
        return self.count.step_dev

    @property
    def coef_dev(self):
        return self.count.coef_dev

    def advance(self, lr, b1, b2):
        """Begin the next optimizer step (once per step, before any adam() range of it)."""
        self.count.advance(lr, b1, b2)

    def set_step(self, step):
        self.count.set(step)

    def adam(self, lr, b1, b2, eps=1e-8, grad_scale=1.0, lo=0, hi=None, advance=True, max_blocks=0, count=None, g16=None):
        """One Adam update of the elements [lo, hi) (default: everything).  ``advance=False``:
        a further range of the SAME optimizer step (the step count is shared).  ``count``: the
        :class:`StepCount` whose bias corrections apply (default: the buffer's main one)."""
        if advance:
            self.advance(lr, b1, b2)
        hi = self.total if hi is None else hi
        if hi > lo and g16 is not None:
            # ``g16``: a bf16 image of the whole gradient buffer (same element offsets) holding this range's gradient
            ops.adam_step_dev_g16_(self.p[lo:hi], g16[lo:hi], self.m[lo:hi], self.v[lo:hi], b1, b2, eps,
                                   (count or self.count).coef_dev, grad_scale, max_blocks)
        elif hi > lo:
            ops.adam_step_dev_(self.p[lo:hi], self.g[lo:hi], self.m[lo:hi], self.v[lo:hi], b1, b2, eps,
                               (count or self.count).coef_dev, grad_scale, max_blocks)


class PCAATrainer:
    """One process = one GPU.  ``variant`` "v4" (projection heads, the paper's
    PCAA), "base" (train_CGAAE / variant 2) or "v1" (variant 4's networks without the inert discriminator
    head, prior centroids = GaussianMeanLearner(one_hot), PCAA_ablation.py:28-378).

    Variant 1 as the reference EXECUTES it: ``z = Variable(z0 + mus)`` (:186) detaches, so the mean learner
    never receives a gradient -- its parameters stay at their initial values (they sit in optimizer_D with
    ``grad=None``), only its BatchNorm running statistics move (golden: tests/golden/v1_*.npz).
    ``learn_centroids=True`` is the variant's stated intent instead (NOT the reference's behaviour): z stays
    attached, ``pcaa_disc_wgan_gp`` returns d(d_loss)/dz and the learner is trained by optimizer_D."""

    _COMPRESS_MIN = 1 << 16       # gradient buckets below this many elements cross the wire as fp32

    def __init__(self, config, n_classes=None, device="cuda", variant="v4", precision=None,
                 process_group=None, sync_bn=False, learn_centroids=False, dp_zero=False, grad_compress=None,
                 force_collectives=False, fused_decoder_update=True, dp_gather=False, emulate_world=0):
        """``fused_decoder_update`` (single process, every precision mode since round 5; "all" is the older spelling of
        True): the decoder's wide weight gradients are formed and consumed by one kernel per layer that applies Adam in
        place (pcaa_skinny_linear_wgrad_adam; the parity modes "fp32" / "fp16x3" use its fp32-product form) -- those
        gradients never exist in ``flat_g.g`` (``self.gradless_ranges`` lists the [lo, hi) element ranges the last step
        left without one); pass False to keep them (gradient inspection, clipping, the parity tests that read them: the
        resulting parameters are bit-identical either way at the op level, and the trainer-level bridge holds the fused
        run to the gate two unfused runs meet, tests/test_round2_parity.py).
        Data-parallel options (``process_group`` given): ``dp_zero`` True = sharded decoder optimizer
        (reduce-scatter, Adam on 1/world of the decoder, all-gather; default off); ``grad_compress="bf16"`` = the decoder's gradient buckets cross the wire as bf16 (half the
        bytes; fp32 master gradients, moments and weights; each bucket is rounded once before the sum and the sum
        is accumulated in bf16 by the collective: relative error of a reduced element ~2^-8).
        ``force_collectives`` issues the collectives on a 1-rank group too (exercises the RCCL calls on one GPU).
        ``dp_gather`` (round 5; bf16 mode with the fused update): the wide decoder layers exchange their two small
        weight-gradient operands -- dz [B, out] and x [B, in], all-gathered over the ranks -- instead of the gradient
        (4 world B (in + out) bytes against 4 in out: ~10x less on the wire), and every rank forms the GLOBAL gradient
        inside the fused weight-gradient + Adam kernel (pcaa_skinny_linear_wgrad_adam_rows): the update stays 24 B per
        parameter and no Adam pass over the decoder follows the exchange.  Mathematically the all-reduce scheme's step
        (sum over ranks of dz_r^T x_r = stacked-rows product).  Round 6: the two operands travel as ONE packed bf16 chunk
        per rank and layer (ops.pack_rows_t16: transposed, 64 batch rows, the rounding the weight-gradient kernels apply
        anyway) -- half the bytes, one all-gather per layer -- and pcaa_skinny_linear_wgrad_adam_t16 contracts over the
        ranks' chunks.  Applies while world <= 8 and B <= 64 (else the all-reduce scheme runs; ``dp_scheme`` says which).
        ``emulate_world=W`` (round 6; no process group): this process runs ONE RANK'S PROGRAM of a W-rank job on its own --
        gradient scale 1/W, W * B stacked rows in the gathered update, every collective replaced by a device operation of
        the same bytes on a stream of its own (dist.EmulatedExchange; peers' gathered rows can be staged there).  It
        measures what the rank's GPU does at that world size; the wire is not in it."""
        self.cfg = dict(config)
        self.K = n_classes if n_classes is not None else len(config["TRAIN_CLASSES"])
        self.N = config["NMAX"]
        self.C = constants.NFEATURES
        self.T = constants.NSTEPS
        self.L = config["SUP_LATENT_DIM"]
        self.variant = variant
        self.device = torch.device(device)
        self.precision = precision
        self.pg = process_group
        self.world = 1
        # SyncBN group: installed in functional for the duration of this trainer's train-mode work only
        # (_sync_bn()), so a later trainer or module call in the same process never all-reduces over it
        self._sync_bn_group = None
        self._xchg = None               # the step's exchanges (dist.GroupExchange; emulate_world: dist.EmulatedExchange)
        if process_group is not None:
            if emulate_world:
                raise ValueError("PCAATrainer: emulate_world runs without a process group")
            self._xchg = pdist.GroupExchange(process_group)
            self.world = self._xchg.world
            if sync_bn:
                self._sync_bn_group = process_group
        elif emulate_world:
            if sync_bn or dp_zero:
                raise ValueError("PCAATrainer: emulate_world covers the all-reduce and gathered-operand schemes without SyncBN")
            self._xchg = pdist.EmulatedExchange(int(emulate_world), self.device)
            self.world = int(emulate_world)
            force_collectives = True
        # what the LAST step did with the decoder's gradients: "none" (single process), "allreduce", "zero", "gather"
        self.dp_scheme = "none"
        self._dp_zero_arg = bool(dp_zero)
        self._dp_gather = bool(dp_gather)
        if self._dp_gather and self._dp_zero_arg:
            raise ValueError("PCAATrainer: dp_gather and dp_zero are alternatives")
        self._gather_bufs = {}          # (layer, world) -> (gathered packed chunks [world, (N + K) * 64] bf16, own chunk)
        self._force_collectives = bool(force_collectives)
        self.fused_decoder_update = bool(fused_decoder_update)
        # the parity modes too (fp32-product kernels); tests that read the decoder's weight gradients pass False
        self.fused_exact = bool(fused_decoder_update)
        if grad_compress not in (None, "bf16"):
            raise ValueError("grad_compress must be None or 'bf16'")
        self.grad_compress = grad_compress
        self.comm = {"collectives": 0, "payload_bytes": 0}     # of the LAST step (gradient / parameter exchanges)
        # bench.py's data-parallel legs: with ``time_comm`` set, every step appends ONE LIST of timing-event pairs on the
        # main stream, one pair around every place where that stream waits for an exchange: (1) from just before the
        # (synchronous) all-reduce of the encoder + head gradients to just after the waits for the decoder buckets;
        # (2) ZeRO: the waits for the all-gathers of the updated decoder shards (they come after the encoder's Adam);
        # (3) SyncBN: each synchronous statistics all-reduce (functional._sync_stats); (4) the join with the side stream's
        # decoder update, which is where a late all-gather of the gathered-operand scheme is paid.  The sum over a step's pairs is
        # what the step pays for communication it could not hide ("exposed"), measured where it is paid.
        # exposed_comm_us() turns the record into microseconds per step.
        self.time_comm = False
        self.comm_events = []
        self._step_events = None
        if variant not in ("v4", "base", "v1", "v3"):
            raise ValueError(f"PCAATrainer: unknown variant {variant!r}")
        head = variant in ("v4", "v1")
        self.learn_centroids = bool(learn_centroids) and variant == "v1"
        # construction order = the reference's (PCAA_ablation.py:764-786): same draws from torch's RNG
        self.encoder = CGEncoder(n_out_labels=self.K, use_projection_head=head, nmax_points=self.N).to(self.device).float()
        # variant 3 has no decoder at all (PCAA_ablation.py:404-415)
        self.decoder = None if variant == "v3" else \
            CGDecoder(input_dim=self.L * 2 if head else self.L, nmax_points=self.N).to(self.device).float()
        if variant != "v1":
            self.discriminator = CGDiscriminator(self.K).to(self.device).float()
        self.decoder_projection_head = self.discriminator_projection_head = self.mean_learner = None
        if variant == "v1":
            # PCAA_ablation.py:44-64: encoder, decoder, decoder head, discriminator, mean learner
            self.decoder_projection_head = torch.nn.Sequential(
                torch.nn.Linear(self.L, self.L * 2), torch.nn.ELU()).to(self.device).float()
            self.discriminator = CGDiscriminator(self.K).to(self.device).float()
            self.mean_learner = GaussianMeanLearner(self.K).to(self.device).float()
        elif head:
            self.decoder_projection_head = torch.nn.Sequential(
                torch.nn.Linear(self.L, self.L * 2), torch.nn.ELU()).to(self.device).float()
            self.discriminator_projection_head = torch.nn.Sequential(
                torch.nn.Linear(self.L * 2, self.L), torch.nn.ELU()).to(self.device).float()
        self.discriminator_means = None
        self._flat_ready = False
        self.gradless_ranges = []      # see _step: ranges of flat_g.g the last step left without a gradient
        self._graphs = {}
        # raised on the device by a label outside [0, K) (torch's CrossEntropyLoss raises there); read by check()
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._sup_split = False        # an unsupervised step has happened: MLP_head / MLP_sup2 count separately

    # ------------------------------------------------------------------ setup
    def set_prior_means(self, means):
        self.discriminator_means = means.float().to(self.device).contiguous()

    def sample_prior_means(self):
        self.set_prior_means(sample_distant_points(dimension=self.L, n=self.K, min_dist=10, sphere_radius=10))
        return self.discriminator_means

    def _sync_bn(self):
        """Context: BatchNorm statistics are all-reduced over this trainer's group inside (no-op without SyncBN)."""
        return F_hip.sync_bn_group(self._sync_bn_group)

    def betas_g(self):
        """optimizer_G's betas: (B1, B2) -- except variant 3, whose G optimizer is built with (B1, B1)
        (PCAA_ablation.py:455; a typo of the reference that its runs executed, so it is what is reproduced)."""
        c = self.cfg
        return (c["B1"], c["B1"]) if self.variant == "v3" else (c["B1"], c["B2"])

    def modules(self):
        d = {"E": self.encoder, "D": self.discriminator}
        if self.decoder is not None:
            d = {"E": self.encoder, "G": self.decoder, "D": self.discriminator}
        if self.decoder_projection_head is not None:
            d["GPH"] = self.decoder_projection_head
        if self.discriminator_projection_head is not None:
            d["DPH"] = self.discriminator_projection_head
        if self.mean_learner is not None:
            d["ML"] = self.mean_learner
        return d

    def finalize(self):
        """Build the flat buffers (call after loading/filling weights)."""
        g_named = [("E." + n, p) for n, p in self.encoder.named_parameters()]
        if self.decoder_projection_head is not None:
            g_named += [("GPH." + n, p) for n, p in self.decoder_projection_head.named_parameters()]
        # bn1..bn4 of the decoder never receive a gradient: torch.optim.Adam skips them
        # order inside the buffer: first layer and all biases, then the wide weights -- with the fused
        # weight-gradient + Adam kernels what is left for the plain Adam pass is one contiguous range
        if self.decoder is not None:
            dec_named = [("G." + n, p) for n, p in self.decoder.named_parameters() if n.startswith("dense")]
            is_small = lambda n: n.startswith("G.dense1.") or n.endswith(".bias")
            g_named += [(n, p) for n, p in dec_named if is_small(n)] + [(n, p) for n, p in dec_named if not is_small(n)]
        # Decoder widths that are not multiples of 64 (N=150, the reference's default: 1125 ... 18000) keep the
        # weight-streaming kernels away -- rows of 1125 floats are not even 16-B aligned.  The weights are then
        # STORED zero-padded to multiples of 64 in both dimensions and the decoder runs in the padded widths
        # (functional.decoder_forward): padded outputs are ELU(0) = 0, padded rows / columns get exactly zero
        # gradients (their operands are zero), so Adam leaves the padding at zero and the [:N,:K] windows the
        # modules expose are bit-for-bit what an unpadded run of the same kernels would hold.
        pads = {}
        if self.decoder is not None:
            r64 = lambda v: (v + 63) // 64 * 64
            lins = self.decoder.dense_layers()
            if any(l.weight.shape[0] % 64 for l in lins):
                for i, l in enumerate(lins, start=1):
                    n_out, n_in = l.weight.shape
                    kp = n_in if i == 1 else r64(n_in)             # the first layer's input is the latent itself
                    pads[f"G.dense{i}.weight"] = (r64(n_out), kp)
                    pads[f"G.dense{i}.bias"] = (r64(n_out),)
        # Data-parallel with a sharded decoder optimizer (PCAA_DP_ZERO=1, ZeRO-1 style): the decoder gradients are
        # reduce-SCATTERED, every rank runs Adam on its 1/world slice only (the 0.78 ms, 4.4 GB update shrinks by the
        # world size) and the updated parameters are all-gathered -- the same bytes on the wire as the all-reduce.
        self._zero = self._dp_zero_arg and self.pg is not None
        self._zero_chunks = 4
        tail = None
        if self.decoder is None:
            self._zero = False
        if self._zero:
            first_dec = next(n for n, _ in g_named if n.startswith("G."))
            tail = (first_dec, pdist.zero_tail_multiple(self.world, self._zero_chunks, _ALIGN))
        self.flat_g = FlatBuffer(g_named, self.device, padded_shapes=pads, tail_multiple=tail)
        if pads:
            from types import SimpleNamespace
            self.decoder._pcaa_pad = [
                SimpleNamespace(weight=self.flat_g.padded[f"G.dense{i}.weight"][0],
                                bias=self.flat_g.padded[f"G.dense{i}.bias"][0])
                for i in range(1, len(self.decoder.dense_layers()) + 1)]
        # the inert discriminator projection head is in optimizer_D but never gets a gradient
        d_named = [("D." + n, p) for n, p in self.discriminator.named_parameters()]
        if self.learn_centroids:
            # optimizer_D = Adam(chain(mean_learner, discriminator)) (PCAA_ablation.py:104-108); in the reference's
            # executed behaviour the learner's gradients are None (Adam skips them), so it stays out of the buffer
            d_named = [("ML." + n, p) for n, p in self.mean_learner.named_parameters()] + d_named
        self.flat_d = FlatBuffer(d_named, self.device)
        if self.variant == "v1":
            self._zero_means = torch.zeros((self.K, self.L), dtype=torch.float32, device=self.device)
            ml = [i for i, nm in enumerate(self.flat_d.names) if nm.startswith("ML.")]
            self._ml_end = self.flat_d.offsets[len(ml)] if ml and len(ml) < len(self.flat_d.offsets) else 0
        self._d_params = ops._disc_params(self.discriminator)
        self._d_grads = [self.flat_d.grad_views["D." + n] for n, _ in self.discriminator.named_parameters()]
        self._dec_grads = {} if self.decoder is None else {
            n: (self.flat_g.padded["G." + n][1] if "G." + n in self.flat_g.padded else self.flat_g.grad_views["G." + n])
            for n, _ in self.decoder.named_parameters() if n.startswith("dense")}
        # encoder gradients are produced straight into the flat buffer; the split-K products
        # accumulate, so that (leading) region is cleared by one fill per step
        self._enc_grads = {n: self.flat_g.grad_views["E." + n] for n, _ in self.encoder.named_parameters()}
        n_enc = sum(1 for nm in self.flat_g.names if nm.startswith("E."))
        enc_end = self.flat_g.offsets[n_enc] if n_enc < len(self.flat_g.offsets) else self.flat_g.total
        self._enc_region = self.flat_g.g[:enc_end]
        # MLP_head / MLP_sup2 only feed the logits: on an unsupervised step (i % SUPERVISION_FREQUENCY != 0) their
        # ``.grad`` stays None in the reference and torch.optim.Adam skips them entirely -- no update, no moment
        # decay, no advance of their per-parameter step.  They are the last encoder parameters: one contiguous range.
        sup_idx = [i for i, nm in enumerate(self.flat_g.names)
                   if nm.startswith("E.MLP_head.") or nm.startswith("E.MLP_sup2.")]
        assert sup_idx == list(range(sup_idx[0], sup_idx[-1] + 1)) and sup_idx[-1] == n_enc - 1
        self._sup_range = (self.flat_g.offsets[sup_idx[0]], enc_end)
        # projection-head + decoder gradients (>98 % of the bytes) are complete before the encoder
        # backward starts: their all-reduce is issued asynchronously and overlaps it
        self._tail_region = self.flat_g.g[enc_end:]
        # The decoder's parameters (98 % of optimizer_G's bytes: a 0.8 ms HBM-bound Adam pass) are
        # final once its backward -- and, data-parallel, the all-reduce of that region -- is done,
        # long before the encoder backward ends.  Their Adam update runs on a side stream beside the
        # latency-bound temporal-conv / MLP-head backward kernels instead of after everything.
        dec_names = [i for i, nm in enumerate(self.flat_g.names) if nm.startswith("G.")]
        self._dec_start = self.flat_g.offsets[dec_names[0]] if dec_names else self.flat_g.total
        # layer number -> (lo, hi, W, exp_avg, exp_avg_sq) of the weights a fused update may take over (views of
        # the flat buffers in the STORED, i.e. possibly padded, shape)
        self._dec_fused = {}
        if self.decoder is not None:
            fg = self.flat_g
            for i in range(2, len(self.decoder.dense_layers()) + 1):
                nm = f"G.dense{i}.weight"
                k = fg.names.index(nm)
                o, n = fg.offsets[k], fg.sizes[k]
                shp = tuple(fg.padded[nm][0].shape) if nm in fg.padded else tuple(fg.params[k].shape)
                self._dec_fused[i] = (o, (o + n + _ALIGN - 1) // _ALIGN * _ALIGN,      # up to the next parameter's offset
                                      fg.p[o:o + n].view(shp), fg.m[o:o + n].view(shp), fg.v[o:o + n].view(shp))
        # layer -> [bf16 image of the weight, the version pair it stands for]: kept by the fused update, streamed by the
        # decoder's forward / dgrad of single-process bf16 steps (_refresh_w16)
        self._dec_w16 = {}
        self.w16_casts = 0          # images (re)built by a cast pass: stays at one per layer unless the weights change outside
        self._side_adam_blocks = 256
        self._dp_chunks = 4
        # data-parallel bucketing: one all-reduce per decoder layer, issued as soon as that layer's gradient is
        # written (fallback without the wgrad stream: the whole region in 4 pieces after the decoder backward)
        # measured (round 1, same box, ms/step): no side stream 8.08-8.26 | beside the temporal-conv/head backward
        # 7.90 (256 blocks), 7.99 (128) | beside the PointNet backward GEMMs 8.29 (256) .. 9.03 (32): the GEMMs lose
        # more to the extra HBM stream than the update costs on its own; 256 blocks 6.63 | 128: 6.91 | 512: 6.72 | 1024: 6.76
        # The three side streams are made ONCE per device and shared by every trainer of the process: HIP maps streams
        # onto a few hardware queues round-robin, and a later trainer's fresh streams can land on the main stream's queue
        # (or on each other's) -- the fourth trainer built in one process ran 6.0 instead of 5.7 ms/step (round 3,
        # tools/leg_check.py).  Trainers of one process never step concurrently, so sharing is safe.
        self._side, self._aux, self._wg = _side_streams(self.device)
        # _side: the decoder's Adam; _aux: the critic branch of the step (see step()); _wg: the small weight-gradient
        # products of the temporal block / heads (functional._WGRAD_STREAM)
        if self._zero:
            n = (self.flat_g.total - self._dec_start) // self._zero_chunks
            self._zero_len = n                                            # floats per chunk (divisible by world * 64)
            import torch.distributed as dist
            self._zero_slices = pdist.zero_slices(self._dec_start, self.flat_g.total, self.world, self._zero_chunks,
                                                  dist.get_rank(self.pg))
            self._zero_g = [torch.empty(n // self.world, dtype=torch.float32, device=self.device)
                            for _ in range(self._zero_chunks)]            # this rank's reduced gradient slice
            self._zero_p = [torch.empty(n // self.world, dtype=torch.float32, device=self.device)
                            for _ in range(self._zero_chunks)]            # staging of the updated slice for the gather
            if self.grad_compress == "bf16":
                self._zero_g16 = [torch.empty(n // self.world, dtype=torch.bfloat16, device=self.device)
                                  for _ in range(self._zero_chunks)]      # this rank's slice of the bf16 sum
        self._g16_direct = set()
        if self.grad_compress == "bf16" and self._xchg is not None:
            self._g16 = torch.zeros(self.flat_g.total, dtype=torch.bfloat16, device=self.device)
            # the wide decoder layers' weight gradients are produced as bf16 straight into the wire image
            # (ops.skinny_linear_wgrad with a bf16 destination): layer -> (offset, bf16 view in the stored shape)
            self._dec_grads16 = {i: (lo, self._g16[lo:lo + Wv.numel()].view(Wv.shape))
                                 for i, (lo, hi, Wv, mv, vv) in self._dec_fused.items()}
        self._stats_pool = ops.StatsPool(self.device)
        self._flat_ready = True
        if self.pg is not None and self.world > 1:
            self.sync_replicas()

    @torch.no_grad()
    def sync_replicas(self, src_rank=0):
        """Make every rank's replica rank ``src_rank``'s: parameters (the two flat buffers and every parameter
        outside them: decoder bn1..4, the inert discriminator head, an untrained mean learner), all module
        buffers (BatchNorm running statistics, ``num_batches_tracked``), the Adam moments and step counts, and
        the prior centroids.  The reference never seeds, so under torchrun each rank draws its own initial
        weights; without this the averaged gradients would be applied to diverging models."""
        import torch.distributed as dist
        if self.pg is None or self.world == 1:
            return
        src = dist.get_global_rank(self.pg, src_rank)
        flat = []
        for fb in (self.flat_g, self.flat_d):
            flat += [fb.p, fb.m, fb.v, fb.count.step_dev, fb.count.coef_dev, fb.sup_count.step_dev,
                     fb.sup_count.coef_dev]
        lo_hi = [(fb.p.data_ptr(), fb.p.data_ptr() + fb.p.numel() * 4) for fb in (self.flat_g, self.flat_d)]
        rest = []
        for mod in self.modules().values():
            for t in list(mod.parameters()) + list(mod.buffers()):
                if not any(lo <= t.data_ptr() < hi for lo, hi in lo_hi):
                    rest.append(t.data)
        if self.discriminator_means is not None:
            rest.append(self.discriminator_means)
        for t in flat + rest:
            dist.broadcast(t, src=src, group=self.pg)
        for fb in (self.flat_g, self.flat_d):
            for c in (fb.count, fb.sup_count):
                c.step = int(c.step_dev.item())

    def check(self):
        """Raise if any step / validation batch so far carried a label outside [0, K) (one host sync; the loops
        call it once per epoch)."""
        if int(self._err.item()):
            raise IndexError(f"PCAATrainer: a ground-truth label was outside [0, {self.K})")
        # fp16x3 mode: an operand image that left fp16's range (saturated, flagged on the device) voids the step
        ops.range_check(self.device)

    def train(self):
        for m in self.modules().values():
            m.train()

    def eval(self):
        for m in self.modules().values():
            m.eval()

    # ------------------------------------------------------------------ one step
    def _count(self, nbytes, kind="allreduce"):
        """``kind``: "allreduce" (a ring moves 2 (w-1)/w of the payload per rank), "gather" / "scatter" ((w-1)/w of it)"""
        self.comm["collectives"] += 1
        self.comm["payload_bytes"] += int(nbytes)
        self.comm[kind + "_bytes"] = self.comm.get(kind + "_bytes", 0) + int(nbytes)

    def _allreduce(self, t, async_op=False):
        if self._collective() and t.numel():
            in_dec = (t.data_ptr() >= self.flat_g.g.data_ptr() + 4 * self._dec_start
                      and t.data_ptr() < self.flat_g.g.data_ptr() + 4 * self.flat_g.total)
            off = (t.data_ptr() - self.flat_g.g.data_ptr()) // 4 if in_dec else -1
            # a bucket whose weight gradient exists ONLY as the bf16 wire image (_g16_direct) must take the compressed
            # path whatever its size: its fp32 range was never written
            if self.grad_compress == "bf16" and in_dec and (t.numel() >= self._COMPRESS_MIN or off in self._g16_direct):
                # decoder gradient bucket: round to bf16 once, sum on the wire in bf16, widen back into the fp32
                # gradient buffer when the consumer waits for it
                g16 = self._g16[off:off + t.numel()]
                if off not in self._g16_direct:          # else: the weight-gradient kernel wrote the bf16 image itself
                    g16.copy_(t)
                self._count(2 * t.numel())
                work = self._xchg.all_reduce(g16, async_op=async_op)
                if not async_op:
                    t.copy_(g16)
                    return None
                return _CompressedWork(work, t, g16)
            assert off not in self._g16_direct, "a bf16-direct gradient bucket reached the fp32 all-reduce"
            self._count(t.numel() * t.element_size())
            return self._xchg.all_reduce(t, async_op=async_op)
        return None

    def _collective(self):
        """whether this trainer's steps issue collectives (a group of more than one rank, a forced 1-rank group, or the
        emulation of a world)"""
        return self._xchg is not None and (self.world > 1 or self._force_collectives)

    def _advance_g(self, supervise):
        """Begin optimizer_G's next step: the main count always, the supervised-only parameters' count on
        supervised steps (see finalize: _sup_range)."""
        b1, b2 = self.betas_g()
        self.flat_g.advance(self.cfg["LR"], b1, b2)
        if supervise:
            self.flat_g.sup_count.advance(self.cfg["LR"], b1, b2)
        else:
            self._sup_split = True

    def _adam_g(self, lo, hi, supervise, gs, max_blocks=0):
        """Adam over [lo, hi) of optimizer_G's flat buffer for the step _advance_g began.  The MLP_head / MLP_sup2
        range is skipped on unsupervised steps and uses its own step count once the two counts differ (one launch
        as long as every step so far was supervised: the counts, hence the coefficients, are then identical)."""
        b1, b2 = self.betas_g()
        lr, fg = self.cfg["LR"], self.flat_g
        a, b = max(lo, self._sup_range[0]), min(hi, self._sup_range[1])
        if a >= b or (supervise and not self._sup_split):
            fg.adam(lr, b1, b2, grad_scale=gs, lo=lo, hi=hi, advance=False, max_blocks=max_blocks)
            return
        fg.adam(lr, b1, b2, grad_scale=gs, lo=lo, hi=a, advance=False, max_blocks=max_blocks)
        if supervise:
            fg.adam(lr, b1, b2, grad_scale=gs, lo=a, hi=b, advance=False, count=fg.sup_count)
        fg.adam(lr, b1, b2, grad_scale=gs, lo=b, hi=hi, advance=False, max_blocks=max_blocks)

    def _w16_active(self, mode, B, collective):
        """the layers whose weight a fused wgrad+Adam kernel will update in a step of this kind (the only updater that keeps
        the bf16 image): layer -> (W, m, v) views"""
        # PCAA_DEC_W16=1 turns the images on.  Off by default: measured (round 4, same box) the forward / dgrad kernels drop
        # from 0.170 + 0.156 to 0.118 + 0.096 ms per step, but keeping the images current costs more than that -- written
        # from inside the update kernel (64-B partial lines from different CUs) that kernel went 1.06 -> 2.1 ms, as a cast
        # pass behind it the extra 0.9 GB of side-stream traffic: step 5.39-5.44 -> 5.48-5.53 ms, N=256 11.9-12.0 -> 12.3-12.6
        # (profiles/r04_ab_dec_w16.txt).  The step is short of HBM bandwidth, not of decoder time.
        if (self.decoder is None or not self.fused_decoder_update or collective or self._side is None or mode != "bf16"
                or os.environ.get("PCAA_DEC_W16", "0") != "1"):
            return {}
        return {i: t for i, t in self._dec_fused.items() if F_hip._skinny(mode, B, t[2].shape[0], t[2].shape[1])}

    def _refresh_w16(self, mode, B, collective):
        """{layer: bf16 image of its weight} for a step of this kind (functional.decoder_forward / _backward stream the
        images instead of the fp32 matrices: half the bytes, same rounding, same result).  An image stands for the
        (flat-buffer, module-parameter) version pair it was built or last updated under: anything that writes the weights
        through torch (load_state_dict, a broadcast, an all-gather) moves one of the two and the image is rebuilt here;
        the flat Adam kernel, which writes through raw pointers, drops the image explicitly (else-branch below)."""
        images = {}
        if self.decoder is None:
            return images
        params = self.decoder.dense_layers()
        active = self._w16_active(mode, B, collective)
        for i, t in self._dec_fused.items():
            if i not in active:
                if i in self._dec_w16:
                    self._dec_w16[i][1] = None          # this step's update will not keep it
                continue
            Wv = t[2]
            ent = self._dec_w16.get(i)
            if ent is None:
                ent = self._dec_w16[i] = [torch.empty(Wv.shape, dtype=torch.bfloat16, device=Wv.device), None]
            ver = (Wv._version, params[i - 1].weight._version)
            if ent[1] != ver:
                ops.cast_bf16(Wv, want_transposed=False, out=ent[0])
                ent[1] = ver
                self.w16_casts += 1
            images[i] = ent[0]
        return images

    def step(self, pcs, gt, z0, alphas, supervise=True):
        """One iteration of the reference's inner loop (PCAA_ablation.py:882-1021; variant 3: :514-655).
        pcs [B,C,T,N] fp32 (ideally a permuted view of point-major storage),
        gt [B] int64, z0 [B,L] fp32, alphas [B,1] fp32 -- all on the device.
        ``supervise=False`` (``i % SUPERVISION_FREQUENCY != 0``): no cross-entropy term, and the parameters only it
        reaches are left alone by Adam, as in the reference.  Returns a dict of DEVICE tensors (no host sync)."""
        with self._sync_bn():
            try:
                return self._step(pcs, gt, z0, alphas, supervise)
            finally:
                F_hip._SYNC_BN["events"] = None
                if self._step_events:
                    self.comm_events.append(self._step_events)
                self._step_events = None

    def _step(self, pcs, gt, z0, alphas, supervise):
        if not self._flat_ready:
            self.finalize()
        if self.discriminator_means is None and self.variant != "v1":
            raise RuntimeError("prior means not set: call sample_prior_means() / set_prior_means()")
        cfg = self.cfg
        B = pcs.shape[0]
        mode = self.precision or F_hip.get_precision()
        enc, dec = self.encoder, self.decoder
        gs = 1.0 / self.world
        ops.set_stats_pool(self._stats_pool)
        self._stats_pool.begin()
        self._enc_region.zero_()
        self.comm = {"collectives": 0, "payload_bytes": 0}
        F_hip._SYNC_BN["collectives"] = F_hip._SYNC_BN["payload_bytes"] = 0
        step_events = self._step_events = [] if self.time_comm else None
        F_hip._SYNC_BN["events"] = step_events if self._sync_bn_group is not None else None

        # (1) encoder forward (train-mode BatchNorm)
        # (the decoder projection head rides in the launch of the MLP heads)
        logits, sup_fv, st = F_hip.encoder_forward(enc, pcs, True, mode, gph=self.decoder_projection_head)
        F_hip.mark("heads_fwd")
        # (2) cross-entropy, its gradient and the predicted labels in one launch
        sup_loss, dlogits, preds = ops.cross_entropy(logits, gt, want_loss=True, want_grad=supervise,
                                                     grad_scale=1.0, want_preds=True, err_flag=self._err)
        # (3) D-step: prior sample, WGAN-GP loss + closed-form gradients, Adam; then the adversarial
        # term of the G-step with the UPDATED critic.  ~10 launches of one wave per batch row: they run
        # on a second stream beside the HBM-bound decoder forward / Chamfer / decoder backward, which do
        # not depend on the critic, and join where the adversarial gradient enters the G backward.
        adv = float(cfg["ADV_WEIGHT"])

        def critic_branch(outs=(None, None, None)):
            """``outs``: (losses [2], loss_g [], dsup [B, L]) destinations allocated by the caller"""
            if self.variant == "v1":
                # centroids from the mean learner (train-mode BatchNorm over the batch's one-hots, :170)
                _, oh = ops.prior_sample(z0, self._zero_means, gt, self.K)
                if self.learn_centroids:
                    for n, p in self.mean_learner.named_parameters():
                        p.grad = self.flat_d.grad_views["ML." + n]
                    self.flat_d.g[:self._ml_end].zero_()
                    mus = self.mean_learner(oh)
                    z = z0 + mus.detach()
                    dl, _, dz = ops.disc_wgan_gp(z, sup_fv, oh, alphas.reshape(-1).contiguous(), self._d_params,
                                                 cfg["GP_WEIGHT"], grads_out=self._d_grads, want_dz=True, losses_out=outs[0])
                    mus.backward(dz)
                else:
                    with torch.no_grad():
                        z = z0 + self.mean_learner(oh)
                    dl, _ = ops.disc_wgan_gp(z, sup_fv, oh, alphas.reshape(-1).contiguous(), self._d_params,
                                             cfg["GP_WEIGHT"], grads_out=self._d_grads, losses_out=outs[0])
            else:
                z, oh = ops.prior_sample(z0, self.discriminator_means, gt, self.K)
                dl, _ = ops.disc_wgan_gp(z, sup_fv, oh, alphas.reshape(-1).contiguous(), self._d_params,
                                         cfg["GP_WEIGHT"], grads_out=self._d_grads, losses_out=outs[0])
            self._allreduce(self.flat_d.g)
            self.flat_d.adam(cfg["LR"], cfg["B1"], cfg["B2"], grad_scale=gs)
            synth = ops.disc_forward(sup_fv, oh, self._d_params)
            lg = ops.total(synth, -adv / B, out=outs[1])
            gout = torch.full((B,), -adv / B, dtype=torch.float32, device=self.device)
            ds, _, _ = ops.disc_backward(sup_fv, oh, self._d_params, gout, want_dx=True, want_params=False, dx_out=outs[2])
            return dl, lg, ds

        joined = None
        if self._aux is not None:
            main = ops.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            # The three results the main stream consumes are allocated HERE, from the main stream's pool, and written on the
            # aux stream: a tensor allocated on the aux stream and handed to the main one (record_stream(main)) costs an
            # event record ON THE MAIN STREAM when it is freed -- four of them were 18 us of the 40 us step boundary
            # (round 6, probe: 40.1 -> 21.6 us without them).  This way the records fall on the aux stream.
            outs = (torch.empty(2, dtype=torch.float32, device=self.device), torch.empty((), dtype=torch.float32, device=self.device),
                    torch.empty_like(sup_fv))
            for t in outs:
                t.record_stream(self._aux)
            with ops.on_stream(self._aux):
                self._aux.wait_event(fork)
                d_losses, loss_g, dsup = critic_branch(outs)
                joined = torch.cuda.Event()
                joined.record(self._aux)
        else:
            d_losses, loss_g, dsup = critic_branch()

        if self.variant == "v3":
            # no decoder, no reconstruction term: tot = loss_g (+ sup_loss) (PCAA_ablation.py:621-645)
            if joined is not None:
                ops.current_stream().wait_event(joined)
            F_hip.set_wgrad_stream(self._wg)
            try:
                F_hip.encoder_backward(enc, st, dlogits if supervise else None, dsup, gout=self._enc_grads)
            finally:
                F_hip.set_wgrad_stream(None)
            if self._wg is not None:
                ops.current_stream().wait_stream(self._wg)
            self._allreduce(self.flat_g.g)
            self._advance_g(supervise)
            self._adam_g(0, self.flat_g.total, supervise, gs)
            tot = loss_g + (sup_loss if supervise else 0.0)
            return {"d_loss": d_losses[0], "gp": d_losses[1], "rec_loss": None, "loss_g": loss_g,
                    "sup_loss": sup_loss, "tot_loss": tot, "preds": preds, "out_labels": logits, "sup_fvs": sup_fv}

        # (4) G-step forward: decoder + Chamfer (+ fused gradient)
        hproj = st.hproj if self.decoder_projection_head is not None else sup_fv
        w16 = self._refresh_w16(mode, B, self._collective())
        rec, acts = F_hip.decoder_forward(dec, hproj, mode, images=w16)
        F_hip.mark("dec_fwd")
        rec4 = rec.view(B, self.C, self.T, self.N)
        inv_bt = 1.0 / (B * self.T)
        frame_loss, drec = ops.chamfer(rec4, pcs, want_grad=True, grad_scale=inv_bt)
        rec_loss = ops.total(frame_loss, inv_bt)

        F_hip.mark("chamfer")
        # (5) G-step backward (the adversarial gradient w.r.t. sup_fvs seeds the accumulation)
        F_hip.set_wgrad_stream(self._wg)
        if joined is not None and self.decoder_projection_head is None:
            ops.current_stream().wait_event(joined)
            joined = None
        dh = None
        collective = self._collective()
        # Data-parallel: a decoder layer's gradient goes onto the wire as soon as it exists.  The backward
        # produces the 118 M-parameter output layer FIRST (75 % of all gradient bytes), a quarter of a millisecond
        # before the decoder backward is over: its all-reduce is issued from the wgrad stream right behind its
        # weight / bias gradient kernels, the later (smaller) layers follow the same way, and the small rest
        # (first layer, whose bias gradient is on the main stream) goes out after the decoder backward.
        early_buckets = []                         # (lo, hi, work) in flat_g coordinates, in issue order
        layer_hook = None
        zero = self._zero and collective and self._side is not None
        if zero:
            pass
        elif collective and self._wg is not None:
            fg = self.flat_g

            def layer_hook(layer):
                if layer < 2:
                    return
                lo = fg.offsets[fg.names.index(f"G.dense{layer}.weight")]
                nxt = f"G.dense{layer + 1}.weight"
                hi = fg.offsets[fg.names.index(nxt)] if nxt in fg.names else fg.total
                # ordered behind this layer's dW / db kernels, whichever stream the layer's path put them on
                self._wg.wait_stream(ops.current_stream())
                with ops.on_stream(self._wg):
                    early_buckets.append((lo, hi, self._allreduce(fg.g[lo:hi], async_op=True)))

        # Single process: the wide decoder layers' weight gradients go straight into Adam (one kernel per layer forms
        # dW in registers and updates W / exp_avg / exp_avg_sq in place: 24 B per parameter instead of 4 + 28).  The
        # backward only notes each layer's operands; the kernels run on the Adam side stream once the decoder backward
        # -- whose dgrads read the weights they overwrite -- is enqueued (measured, same box, ms/step: unfused 6.27-6.30
        # | behind each layer's own dgrad 6.13-6.18, the decoder backward section 276 -> 634 us | here 6.12-6.14).
        updates, fused_ranges, deferred = None, [], []
        exact_dec = mode in ("fp32", "fp16x3")      # the parity modes: the same fused kernels with fp32 products
        if (self.fused_decoder_update and not collective and self._side is not None
                and (mode == "bf16" or (self.fused_exact and exact_dec))):
            for layer, (lo, hi, Wv, mv, vv) in self._dec_fused.items():
                if F_hip._skinny(mode, B, Wv.shape[0], Wv.shape[1]) or F_hip._skinny_exact(mode, B, Wv.shape[0], Wv.shape[1]):
                    updates = updates or {}
                    updates[layer] = lambda dz2, x, t=(Wv, mv, vv, w16.get(layer)): deferred.append((dz2, x) + t + (None,))
                    fused_ranges.append((lo, hi))
        # Data parallel, dp_gather: the same fused update from the ranks' stacked rows.  A layer's callback runs where the
        # backward has just formed dz2: the operands are packed and their all-gather goes out from there (the collective's own
        # stream picks up behind what is enqueued here; this stream does not wait) and the update kernel follows on the Adam
        # side stream once it is back -- no gradient bucket, no all-reduce, no separate Adam pass for these layers.
        # (round 6: the operands travel PACKED -- one transposed bf16 chunk of 64 batch rows per rank and layer, ops.pack_rows_t16
        # -- half the bytes, one all-gather per layer instead of two, and 16-B fragment loads in the update kernel)
        gather = (collective and self._dp_gather and not zero and mode == "bf16" and self.fused_decoder_update
                  and self._side is not None and self.world <= ops.PACK_MAX_CHUNKS and B <= ops.PACK_ROWS)
        self.dp_scheme = "zero" if zero else ("gather" if gather else ("allreduce" if collective else "none"))
        if gather:
            def gather_update(layer, Wv, mv, vv):
                def cb(dz2, x):
                    key = (layer, self.world)
                    if key not in self._gather_bufs:
                        ce = ops.packed_chunk_elems(Wv.shape[0], Wv.shape[1])
                        self._gather_bufs[key] = (torch.zeros((self.world, ce), dtype=torch.bfloat16, device=self.device),
                                                  torch.empty(ce, dtype=torch.bfloat16, device=self.device))
                    packed_all, own = self._gather_bufs[key]
                    dzc, xc = (dz2 if dz2.stride(1) == 1 else dz2.contiguous()), (x if x.stride(1) == 1 else x.contiguous())
                    self._count(2 * packed_all.numel(), "gather")

                    def send():
                        ops.pack_rows_t16(dzc, xc, out=own)
                        # (flat views: gloo's all-gather wants output and input of the same rank)
                        return self._xchg.all_gather_into_tensor(packed_all.view(-1), own, async_op=True, tag=layer)
                    if self._wg is not None:
                        # pack + all-gather leave from the weight-gradient stream (behind dz2, which this stream has just
                        # formed): the main stream goes straight on to the layer's dgrad
                        self._wg.wait_stream(ops.current_stream())
                        with ops.on_stream(self._wg):
                            dzc.record_stream(self._wg)
                            xc.record_stream(self._wg)
                            work = send()
                    else:
                        work = send()
                    deferred.append((packed_all, None, Wv, mv, vv, None, [work]))
                return cb

            for layer, (lo, hi, Wv, mv, vv) in self._dec_fused.items():
                if F_hip._skinny(mode, B, Wv.shape[0], Wv.shape[1]):
                    updates = updates or {}
                    updates[layer] = gather_update(layer, Wv, mv, vv)
                    fused_ranges.append((lo, hi))
            if layer_hook is not None:
                # the per-layer gradient buckets only for what the gathered update does not take
                plain_hook, taken = layer_hook, set(updates or {})

                def layer_hook(layer):
                    if layer not in taken:
                        plain_hook(layer)
        if not set(w16) <= set(updates or {}):
            # an image is only current if THIS step's update of its weight rewrites it
            raise RuntimeError("PCAATrainer: a bf16 weight image is in use for a layer whose update is not fused")
        dec_grads = self._dec_grads
        self._g16_direct = set()
        if collective and layer_hook is not None and self.grad_compress == "bf16" and mode == "bf16":
            dec_grads = dict(dec_grads)
            for layer, (lo, view16) in self._dec_grads16.items():
                if updates and layer in updates:
                    continue                     # (dp_gather: no gradient of this layer exists in any form)
                # same size rule as _allreduce's compressed path: one predicate decides both
                if F_hip._skinny(mode, B, view16.shape[0], view16.shape[1]) and view16.numel() >= self._COMPRESS_MIN:
                    dec_grads[f"dense{layer}.weight"] = view16
                    self._g16_direct.add(lo)
        # [lo, hi) ranges of flat_g.g that hold NO gradient after this step: the fused weight-gradient + Adam kernels and
        # the bf16-direct wire images never write the fp32 gradient there (round-2 advisor finding: a consumer of
        # grad_views -- clipping, norm logging -- must skip them or construct the trainer with fused_decoder_update=False)
        self.gradless_ranges = sorted(fused_ranges) + sorted(
            (lo, lo + v.numel()) for lo, v in (self._dec_grads16.values() if self._g16_direct else ()) if lo in self._g16_direct)
        if self.decoder_projection_head is not None:
            # the head's own backward (dh -> dsup, dW, db) runs inside the heads' backward launch below
            _, dh = F_hip.decoder_backward(dec, acts, drec, need_dz=True, grads_out=dec_grads, mode=mode,
                                           after_layer=layer_hook, updates=updates, images=w16)
            if joined is not None:
                ops.current_stream().wait_event(joined)
        else:
            _, dsup = F_hip.decoder_backward(dec, acts, drec, need_dz=True, grads_out=dec_grads, dz_init=dsup,
                                             mode=mode, after_layer=layer_hook, updates=updates, images=w16)
        # The decoder's gradients are final here (the projection head's follow with the encoder's: its backward
        # runs in the MLP heads' launch).  Data-parallel: their all-reduce goes out now, in a few chunks (the
        # collectives of one communicator run in order), so that the side-stream Adam of chunk i overlaps the
        # all-reduce of chunk i+1 instead of waiting for all 628 MB.
        F_hip.mark("dec_bwd")
        pending = []                               # (lo, hi, work) in flat_g coordinates
        zero_gather = []
        if zero:
            import torch.distributed as dist
            fg, n = self.flat_g, self._zero_len
            if self._wg is not None:
                ops.current_stream().wait_stream(self._wg)   # the decoder's dW/db were written on that stream
            scatter = []
            z16 = self.grad_compress == "bf16"
            for c in range(self._zero_chunks):
                lo = self._dec_start + c * n
                if z16:
                    # bf16 buckets (round 4): the chunk is rounded once, summed on the wire in bf16, and this rank's
                    # reduced slice goes to Adam as it came off the wire (adam_step_dev_g16_) -- the gradient half of
                    # the exchange moves half the bytes; the updated parameters are gathered in fp32 as before
                    g16 = self._g16[lo:lo + n]
                    g16.copy_(fg.g[lo:lo + n])
                    self._count(2 * n, "scatter")
                    scatter.append(dist.reduce_scatter_tensor(self._zero_g16[c], g16, group=self.pg, async_op=True))
                else:
                    self._count(4 * n, "scatter")
                    scatter.append(dist.reduce_scatter_tensor(self._zero_g[c], fg.g[lo:lo + n], group=self.pg, async_op=True))
            self._advance_g(supervise)

            def launch_zero_adam():
                ready = torch.cuda.Event()
                ready.record(ops.current_stream())
                with ops.on_stream(self._side):
                    self._side.wait_event(ready)
                    for c in range(self._zero_chunks):
                        lo, hi = self._zero_slices[c][2:]                   # this rank's slice of chunk c
                        scatter[c].wait()
                        if z16:
                            ops.adam_step_dev_g16_(fg.p[lo:hi], self._zero_g16[c], fg.m[lo:hi], fg.v[lo:hi], cfg["B1"],
                                                   cfg["B2"], 1e-8, fg.coef_dev, gs, self._side_adam_blocks)
                        else:
                            ops.adam_step_dev_(fg.p[lo:hi], self._zero_g[c], fg.m[lo:hi], fg.v[lo:hi], cfg["B1"], cfg["B2"],
                                               1e-8, fg.coef_dev, gs, self._side_adam_blocks)
                        self._zero_p[c].copy_(fg.p[lo:hi])
                        self._count(4 * n, "gather")
                        zero_gather.append(dist.all_gather_into_tensor(
                            fg.p[self._dec_start + c * n:self._dec_start + (c + 1) * n], self._zero_p[c],
                            group=self.pg, async_op=True))
        elif gather or early_buckets:
            # The wide layers are on the wire already, as gathered operands (no gradient exists) or as per-layer buckets.
            # What is left of the decoder region -- its first layer and the biases in front of the wide weights
            # (finalize), a wide layer neither scheme took, tail padding -- is the region MINUS those ranges, computed
            # as such (round-5 advisor finding: the two branches this replaces each assumed which ranges the other
            # scheme had covered): a few small all-reduces, issued once the wgrad stream's products are in.
            covered = sorted(fused_ranges + [(lo, hi) for lo, hi, _ in early_buckets])
            residual, a = [], self._dec_start
            for lo, hi in covered + [(self.flat_g.total, self.flat_g.total)]:
                if lo > a:
                    residual.append((a, lo))
                a = max(a, hi)
            if residual and self._wg is not None:
                ops.current_stream().wait_stream(self._wg)
            pending = early_buckets + [(lo, hi, self._allreduce(self.flat_g.g[lo:hi], async_op=True)) for lo, hi in residual]
            # (issue order = the order the collectives of one communicator complete in)
        else:
            bounds = [self._dec_start]
            if collective and self._wg is not None:
                ops.current_stream().wait_stream(self._wg)   # the decoder's dW/db were written on that stream
            nchunk = self._dp_chunks if collective else 1
            dec_n = self.flat_g.total - self._dec_start
            for i in range(1, nchunk + 1):
                b = self._dec_start + (dec_n * i // nchunk) // _ALIGN * _ALIGN if i < nchunk else self.flat_g.total
                if b > bounds[-1]:
                    bounds.append(b)
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                pending.append((lo, hi, self._allreduce(self.flat_g.g[lo:hi], async_op=True)))
        early = self._side is not None
        hook = hook_heads = None
        done = []
        if zero:
            early = False
            hook_heads = launch_zero_adam          # beside the temporal block's backward, like the replicated update
        if early:
            self._advance_g(supervise)

            def launch_side_adam():
                ready = torch.cuda.Event()
                ready.record(ops.current_stream())
                with ops.on_stream(self._side):
                    self._side.wait_event(ready)        # everything enqueued on the main stream so far
                    if self._wg is not None:
                        self._side.wait_stream(self._wg)   # ... and the decoder weight gradients on the wgrad stream
                    for dz2, x, Wv, mv, vv, w16, works in deferred:
                        dz2.record_stream(self._side)
                        if x is not None:
                            x.record_stream(self._side)
                        if works is not None:
                            for wk in works:
                                wk.wait()               # this stream waits for the all-gather of THIS layer only
                            ops.skinny_linear_wgrad_adam_t16_(dz2, self.world, Wv, mv, vv, *self.betas_g(), 1e-8,
                                                              self.flat_g.coef_dev, gs)      # (dz2: the gathered packed chunks)
                            continue
                        ops.skinny_linear_wgrad_adam_(dz2, x, Wv, mv, vv, *self.betas_g(), 1e-8, self.flat_g.coef_dev, gs,
                                                      exact=exact_dec)
                        if w16 is not None:
                            # the image follows its weight on this stream (a coalesced 6 B/param pass; written from inside
                            # the update kernel -- 64-B partial lines from different CUs -- it doubled that kernel's time)
                            ops.cast_bf16(Wv, want_transposed=False, out=w16)
                    for lo, hi, work in pending:
                        if hi <= self._dec_start:
                            continue                    # the projection-head slice is updated on the main stream
                        g16 = None
                        if isinstance(work, _CompressedWork) and (lo % 4 == 0):
                            work.wait(widen=False)      # Adam reads the reduced bf16 bucket as it came off the wire
                            work.done = True
                            g16 = self._g16
                        elif work is not None:
                            work.wait()                 # side stream waits for THIS chunk's all-reduce only
                        # what the fused kernels did not take: [lo, hi) minus their ranges
                        a = max(lo, self._dec_start)
                        for flo, fhi in sorted(fused_ranges) + [(hi, hi)]:
                            if flo > a:
                                self.flat_g.adam(cfg["LR"], cfg["B1"], cfg["B2"], grad_scale=gs, lo=a, hi=min(flo, hi),
                                                 advance=False, max_blocks=self._side_adam_blocks, g16=g16)
                            a = max(a, fhi)
                            if a >= hi:
                                break
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    done.append(ev)

            # right after the decoder backward, i.e. beside the heads' and the temporal block's backward (measured
            # against "after the heads' launch" 6.576 | 6.621 and "beside the PointNet backward GEMMs" 6.90 ms/step)
            at = _SIDE_ADAM_AT
            if at == "pointnet":
                hook = launch_side_adam
            elif at == "heads":
                hook_heads = launch_side_adam
            else:
                launch_side_adam()
        try:
            gv = self.flat_g.grad_views
            F_hip.encoder_backward(enc, st, dlogits if supervise else None, dsup, gout=self._enc_grads,
                                   before_pointnet=hook, after_heads=hook_heads,
                                   gph=self.decoder_projection_head, d_hproj=dh,
                                   gph_gout=(gv["GPH.0.weight"], gv["GPH.0.bias"]) if dh is not None else None)
        finally:
            F_hip.set_wgrad_stream(None)
        if self._wg is not None:
            ops.current_stream().wait_stream(self._wg)       # its products feed the all-reduce / Adam below
        ev_c0 = None
        if self.time_comm and collective:
            ev_c0 = torch.cuda.Event(enable_timing=True)
            ev_c0.record()
        self._allreduce(self.flat_g.g[:self._dec_start])      # encoder + projection head
        for _, _, work in pending:
            if work is not None:
                work.wait()         # stream-side wait, no host block
        if ev_c0 is not None:
            ev_c1 = torch.cuda.Event(enable_timing=True)
            ev_c1.record()
            step_events.append((ev_c0, ev_c1))
        if zero:
            self._adam_g(0, self._dec_start, supervise, gs)
            if ev_c0 is not None:
                ev_g0 = torch.cuda.Event(enable_timing=True)
                ev_g0.record()
            for work in zero_gather:
                work.wait()                                      # next forward reads the gathered decoder
            if ev_c0 is not None:
                ev_g1 = torch.cuda.Event(enable_timing=True)
                ev_g1.record()
                step_events.append((ev_g0, ev_g1))
        elif early:
            self._adam_g(0, self._dec_start, supervise, gs)
            ev_j0 = None
            if ev_c0 is not None:
                # the join with the side stream is where the main stream pays for a LATE exchange of the gathered-operand
                # scheme (its all-gathers are waited for on the side stream only) -- and, in every scheme, for what of the
                # decoder update did not fit beside the backward: timed in all of them, so the legs compare like with like
                ev_j0 = torch.cuda.Event(enable_timing=True)
                ev_j0.record()
            ops.current_stream().wait_event(done[0])      # next forward reads the updated decoder
            if ev_j0 is not None:
                ev_j1 = torch.cuda.Event(enable_timing=True)
                ev_j1.record()
                step_events.append((ev_j0, ev_j1))
        else:
            self._advance_g(supervise)
            self._adam_g(0, self.flat_g.total, supervise, gs)

        F_hip.mark("adam+join")
        # SyncBN's statistics all-reduces (functional._sync_stats) belong to the step's exchanges too
        self.comm["collectives"] += F_hip._SYNC_BN.get("collectives", 0)
        self.comm["payload_bytes"] += F_hip._SYNC_BN.get("payload_bytes", 0)
        tot = rec_loss + loss_g + (sup_loss if supervise else 0.0)
        return {"d_loss": d_losses[0], "gp": d_losses[1], "rec_loss": rec_loss, "loss_g": loss_g,
                "sup_loss": sup_loss, "tot_loss": tot, "preds": preds, "out_labels": logits, "sup_fvs": sup_fv}

    def exposed_comm_us(self):
        """Microseconds per recorded step (``time_comm``) the main stream spent waiting for exchanges: the sum over the
        step's event pairs (gradient all-reduce window, ZeRO all-gather waits, SyncBN statistics).  Synchronise first."""
        return [sum(e0.elapsed_time(e1) for e0, e1 in pairs) * 1e3 for pairs in self.comm_events]

    # ------------------------------------------------------------------ hipGraph replay of the step
    def prefers_graph(self, B, N):
        """Whether step_graphed beats step for this shape on one GPU.  The eager step costs the host ~2.5 ms of
        enqueues (~120 launches); that only binds when the GPU needs less: measured (B=64, bf16, ms/step eager |
        graph) N=32: 2.56 | 2.34, N=64: 3.56 | 3.71, N=128: 6.51 | 6.66 -- replay wins below ~80 K points per step
        and loses 2-4 % above (profiles/r02_graph_vs_eager.txt).  Variants without autograd inside the step only; single
        process or an emulated world by default (see can_graph)."""
        return (self.can_graph() and (self._xchg is None or self._xchg.emulated or os.environ.get("PCAA_GRAPH") == "on")
                and B * self.T * N <= 80_000 and os.environ.get("PCAA_GRAPH", "auto") != "off")

    def can_graph(self):
        """Whether step_graphed can capture this trainer's step: no autograd inside it (variant 1's mean learner), and every
        exchange a stream operation -- RCCL's collectives are (round 6: the data-parallel step replays as ONE graph launch
        per rank, collectives included; at small N the eager DP step is bound by the host's ~2.5 ms of enqueues), gloo's
        are host calls.  A real multi-rank group replays only on request (PCAA_GRAPH=on / bench.py --graph on): no node
        with more than one GPU has been available to measure it on."""
        return (self.variant in ("v4", "base", "v3") and self.device.type == "cuda"
                and (self._xchg is None or self._xchg.capturable) and not self.time_comm)

    def step_graphed(self, pcs, gt, z0, alphas, supervise=True, warmup=2):
        """step() through a captured hipGraph.  The step is a fixed sequence of ~200 launches on four
        streams with no host synchronisation and no per-step host scalar (the Adam step count lives on
        the device), so it is captured once per (batch shape, supervise) and replayed: one graph launch
        per step instead of ~200 enqueues, and the latency-bound parts (temporal block, heads, critic)
        no longer wait for the host.  The first ``warmup`` calls of a shape run eagerly (lazy
        initialisation, allocator warm-up), the next one captures; every call performs exactly one real
        train step.  The returned tensors are the graph's static outputs: they are overwritten by the
        next replay (clone what must survive)."""
        if not self.can_graph():
            raise RuntimeError("PCAATrainer.step_graphed: this trainer's step cannot be captured (variant 1's autograd, a gloo "
                               "process group, or time_comm's timing events): use step()")
        if not supervise and not self._sup_split:
            # the first unsupervised step changes the launch sequence of the supervised one as well
            self._sup_split = True
            self._graphs.clear()
        key = (tuple(pcs.shape), tuple(pcs.stride()), bool(supervise))
        ent = self._graphs.setdefault(key, {"eager": 0, "graph": None})
        if ent["graph"] is None:
            if ent["eager"] < warmup:
                ent["eager"] += 1
                return self.step(pcs, gt, z0, alphas, supervise)
            static = [torch.empty_like(pcs), torch.empty_like(gt), torch.empty_like(z0), torch.empty_like(alphas)]
            for dst, src in zip(static, (pcs, gt, z0, alphas)):
                dst.copy_(src)
            steps0 = (self.flat_g.step, self.flat_d.step)
            self._refresh_w16(self.precision or F_hip.get_precision(), pcs.shape[0], False)      # not inside the capture
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(self.device)
            # capture records the launches without running them: parameters, Adam state, BatchNorm
            # running statistics and the device-side step counts are untouched until the first replay
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.step(*static, supervise)
            # the Python side of step() ran once during capture: that bump of the host step mirrors
            # stands for the replay below
            assert (self.flat_g.step, self.flat_d.step) == (steps0[0] + 1, steps0[1] + 1)
            ent.update(graph=graph, static=static, out=out)
            graph.replay()
            return out
        for dst, src in zip(ent["static"], (pcs, gt, z0, alphas)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        # the replay streams the bf16 weight images the capture saw: rebuild them (eagerly, before the replay) if the
        # weights were written from outside since
        self._refresh_w16(self.precision or F_hip.get_precision(), pcs.shape[0], False)
        self.flat_g.step += 1
        self.flat_d.step += 1
        if supervise:
            self.flat_g.sup_count.step += 1
        ent["graph"].replay()
        return ent["out"]

    @torch.no_grad()
    def evaluate_batch(self, pcs, gt):
        """Validation forward (PCAA_ablation.py:1046-1064): eval-mode encoder ->
        projection head -> decoder -> Chamfer, CE, argmax."""
        mode = self.precision or F_hip.get_precision()
        logits, sup_fv, _ = F_hip.encoder_forward(self.encoder, pcs, False, mode)
        if self.decoder is None:         # variant 3 validates cross-entropy and accuracy only (:672-692)
            ce, _, preds = ops.cross_entropy(logits, gt, want_loss=True, want_preds=True, err_flag=self._err)
            return None, ce, preds, sup_fv
        hproj = sup_fv
        if self.decoder_projection_head is not None:
            hproj = F_hip.linear_act_forward(sup_fv, self.decoder_projection_head[0], ACT_ELU)
        rec, _ = F_hip.decoder_forward(self.decoder, hproj, mode)
        B = pcs.shape[0]
        fl, _ = ops.chamfer(rec.view(B, self.C, self.T, self.N), pcs, want_grad=False)
        rec_loss = ops.total(fl, 1.0 / (B * self.T))
        ce, _, preds = ops.cross_entropy(logits, gt, want_loss=True, want_preds=True, err_flag=self._err)
        return rec_loss, ce, preds, sup_fv

    # ------------------------------------------------------------------ checkpoints (format of the reference)
    def save_checkpoints(self, folder, model_name, write=True):
        suffix = {"E": "_E", "G": "_G", "D": "_D", "GPH": "_GPH", "DPH": "_DPH", "ML": "_ML"}
        if write:
            for key, mod in self.modules().items():
                save_model(mod, os.path.join(folder, f"{model_name}{suffix[key]}.pt"))
        if self.mean_learner is not None:
            with self._sync_bn():
                cent = self.learned_centroids()
            if write:
                torch.save(cent, os.path.join(folder, "discriminator_means.pt"))
        elif self.variant == "v3" and write:
            # variant 3 re-saves the (fixed) centroids with every checkpoint (PCAA_ablation.py:738-739)
            torch.save(self.discriminator_means, os.path.join(folder, "discriminator_means.pt"))

    @torch.no_grad()
    def learned_centroids(self):
        """Variant 1's ``discriminator_means.pt`` (PCAA_ablation.py:367-375): the mean learner applied to the K
        one-hot labels -- in whatever mode it is in; the reference never leaves train mode here, so the
        BatchNorm statistics are those of the K one-hots (and the running statistics move)."""
        return self.mean_learner(torch.eye(self.K, dtype=torch.float32, device=self.device))


# ----------------------------------------------------------------------
# loops with the reference's call surface
# ----------------------------------------------------------------------
class _NullRun:
    def finish(self):
        pass


def _wandb():
    try:
        import wandb  # noqa: F401
        return wandb
    except Exception:
        return None


class _EpochDraws:
    """The loop's two host RNG draws (``z0 = np.random.normal(0, 1, (B, L))`` PCAA_ablation.py:915-925 and
    ``alphas = torch.rand(size=(B, 1))`` :944-948) for a WHOLE EPOCH, made at its start: the same calls on the same two
    global generators in the same order as the per-step draws of the reference (nothing else draws from either generator
    between the epoch's order and its validation pass), written into pinned host buffers and moved with ONE asynchronous
    copy each -- instead of two pageable, blocking ``.to(device)`` per step in front of every step's launches (round 5
    VERDICT item 2; SURVEY section 7 "host-side RNG in the loop").  Data parallel: the draws are for the global batch,
    rank 0's values are broadcast ONCE per epoch and every rank slices its rows."""

    def __init__(self, L, device, process_group=None, rank=0, world=1):
        self.L, self.dev, self.pg, self.rank, self.world = int(L), device, process_group, rank, world
        self._z_host = self._a_host = None
        self._copied = None

    def draw(self, steps, gb):
        if steps <= 0:
            return None, None
        if self._z_host is None or self._z_host.shape[0] < steps or self._z_host.shape[1] != gb:
            pin = self.dev.type == "cuda"
            self._z_host = torch.empty((steps, gb, self.L), dtype=torch.float32, pin_memory=pin)
            self._a_host = torch.empty((steps, gb, 1), dtype=torch.float32, pin_memory=pin)
        elif self._copied is not None:
            self._copied.synchronize()                   # last epoch's copies have left the pinned buffers
        z_np = self._z_host.numpy()
        for i in range(steps):
            # one call per step, as in the reference (the float64 -> float32 cast is the same rounding on either side of the copy)
            z_np[i] = np.random.normal(0.0, 1.0, (gb, self.L))
            torch.rand(size=(gb, 1), out=self._a_host[i])
        z = self._z_host[:steps].to(self.dev, non_blocking=True)
        a = self._a_host[:steps].to(self.dev, non_blocking=True)
        if self.dev.type == "cuda":
            self._copied = torch.cuda.Event()
            self._copied.record()
        if self.world > 1:
            import torch.distributed as dist
            src = dist.get_global_rank(self.pg, 0)
            dist.broadcast(z, src=src, group=self.pg)
            dist.broadcast(a, src=src, group=self.pg)
            per = gb // self.world
            z = z[:, self.rank * per:(self.rank + 1) * per]
            a = a[:, self.rank * per:(self.rank + 1) * per]
        return z, a


def _run_loop(config, variant, dataset_factory=None, log_fn=None, process_group=None, sync_bn=False, device="cuda",
              timing=None):
    """``timing``: a list that receives one dict per epoch -- wall-clock seconds of its train part (up to the point where
    the epoch's predictions are on the host, i.e. every step has finished) and of its validation pass, and the number of
    train steps (bench.py's ``loop`` leg)."""
    import time
    from .datasets import MSRadarDataset
    from .constants import SPLIT

    nmax_points = config["NMAX"]
    os.makedirs(f"models/{config['MODEL_NAME']}", exist_ok=True)
    if process_group is None or torch.distributed.get_rank(process_group) == 0:
        with open(os.path.join("models", config["MODEL_NAME"], "config.pkl"), "wb") as f:
            pickle.dump(config, f)

    # Data parallel (process_group given, one process per GPU): config["BATCH_SIZE"] is the GLOBAL batch, as in the
    # single-process reference; every rank steps on its slice of each global batch, and the host RNG draws of the
    # loop (epoch order, z0, alphas) are rank 0's, broadcast -- so the N-rank loop IS the single-process loop on
    # the global batch (exactly so with sync_bn=True; per-rank BatchNorm statistics otherwise).
    rank, world = 0, 1
    if process_group is not None:
        import torch.distributed as dist
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        if config["BATCH_SIZE"] % world:
            raise ValueError(f"BATCH_SIZE {config['BATCH_SIZE']} is not divisible by the world size {world}")
    local_cfg = dict(config)
    local_cfg["BATCH_SIZE"] = config["BATCH_SIZE"] // world
    trainer = PCAATrainer(local_cfg, variant=variant, process_group=process_group, sync_bn=sync_bn, device=device)
    dev = trainer.device
    make = dataset_factory or (lambda split: MSRadarDataset(split, subsample_factor=config["SUBSAMPLE_FACTOR"]))
    train_set, valid_set = make(SPLIT.TRAIN), make(SPLIT.VALID)
    if len(train_set) and len(valid_set):
        # packed store in HBM + device-side batch assembly; same batches (and the same consumption of
        # torch's global RNG) as the DataLoaders of the reference (batcher.py)
        from .batcher import batcher_for
        loader_train = batcher_for(train_set, config["BATCH_SIZE"], dev, shuffle=True, rank=rank, world=world,
                                   group=process_group)
        loader_valid = batcher_for(valid_set, config["BATCH_SIZE"], dev, shuffle=False, rank=rank, world=world,
                                   group=process_group)
    elif world > 1:
        raise RuntimeError("the data-parallel loop needs the device batcher (PCAA_DEVICE_BATCHER=1)")
    else:
        loader_train = torch.utils.data.DataLoader(train_set, batch_size=config["BATCH_SIZE"], drop_last=True,
                                                   shuffle=True, num_workers=0)
        loader_valid = torch.utils.data.DataLoader(valid_set, batch_size=config["BATCH_SIZE"], drop_last=True,
                                                   shuffle=False, num_workers=0)
    _ = make(SPLIT.UNSEEN) if dataset_factory is None else None

    wb = _wandb() if rank == 0 else None          # logging is rank 0's: one wandb run per job, not per rank
    run = _NullRun()
    if wb is not None and hasattr(wb, "init"):
        wb.login()
        run = wb.init(project=constants.WANDB_PROJECT, config=config, name=config["MODEL_NAME"],
                      notes=config["NOTES"], reinit=True, mode=constants.WANDB_MODE)

    if variant != "v1":
        means = trainer.sample_prior_means()
        if rank == 0:
            torch.save(means, os.path.join("models", config["MODEL_NAME"], "discriminator_means.pt"))
    trainer.finalize()

    best_valid_accuracy = 0
    history = []
    L = config["SUP_LATENT_DIM"]
    use_graph = trainer.prefers_graph(local_cfg["BATCH_SIZE"], nmax_points)
    draws = _EpochDraws(L, dev, process_group, rank, world)
    for epoch in range(config["EPOCHS"]):
        t_epoch = time.perf_counter()
        trainer.train()
        steps = []
        ys = []
        # the first batch is fetched BEFORE the epoch's draws: creating / starting the loader's iterator is what consumes
        # torch's global RNG for the epoch order (DataLoader: base seed at iter(), sampler seed at the first next())
        batches = iter(loader_train)
        first = next(batches, None)
        n_steps = len(loader_train) if first is not None else 0
        gb = first[0].shape[0] * world if first is not None else 0
        z0_all, alphas_all = draws.draw(n_steps, gb)
        for i, (pcs, gt_labels) in enumerate(itertools.chain([first] if first is not None else [], batches)):
            pcs = pcs.to(dev, non_blocking=True)
            gt_labels = gt_labels.to(dev, non_blocking=True)
            if i < n_steps and pcs.shape[0] * world == gb:
                z0, alphas = z0_all[i], alphas_all[i]
            else:
                # (a loader that yields more, or other, batches than it announced: the reference's per-step draws)
                z0, alphas = draws.draw(1, pcs.shape[0] * world)
                z0, alphas = z0[0], alphas[0]
            supervise = i % config["SUPERVISION_FREQUENCY"] == 0
            if use_graph:
                # replayed hipGraph (small shapes, where the host's enqueues bound the eager step): the outputs are
                # the graph's static tensors, so what the epoch statistics keep is copied out
                out = trainer.step_graphed(pcs, gt_labels, z0, alphas, supervise=supervise)
                out = {k: (v.clone() if torch.is_tensor(v) and k in ("rec_loss", "d_loss", "sup_loss", "tot_loss", "preds")
                           else v) for k, v in out.items()}
            else:
                out = trainer.step(pcs, gt_labels, z0, alphas, supervise=supervise)
            out["supervised"] = supervise
            steps.append(out)
            ys.append(gt_labels)
        for ld in (loader_train, loader_valid):
            if hasattr(ld, "check"):
                ld.check()               # a batch index outside the packed store (one host sync per epoch)
        trainer.check()

        def _mean(key, only_supervised=False):
            # one device->host transfer per logged quantity and epoch instead of blocking .item()s per step;
            # an empty list averages to nan, like the reference's np.mean([]) (variant 3's reconstruction loss)
            vals = [o[key] for o in steps if o[key] is not None and (o["supervised"] or not only_supervised)]
            return float(torch.stack(vals).double().mean().item()) if vals else float("nan")

        y_hats = torch.cat([o["preds"] for o in steps]).cpu().numpy()
        ys = torch.cat(ys).cpu().numpy()
        t_train = time.perf_counter()

        trainer.eval()
        v_rec, v_ce, v_hat, v_y = [], [], [], []
        for valid_pc, valid_gt in loader_valid:
            valid_pc = valid_pc.to(dev, non_blocking=True)
            valid_gt = valid_gt.to(dev, non_blocking=True)
            r, c, p, _ = trainer.evaluate_batch(valid_pc, valid_gt)
            if r is not None:
                v_rec.append(r)
            v_ce.append(c); v_hat.append(p); v_y.append(valid_gt)
        # the cross-entropy and the total loss are logged over the SUPERVISED steps only (:1005-1012)
        record = {
            "Reconstruction Loss Train": _mean("rec_loss"),
            "Reconstruction Loss Valid": float(torch.stack(v_rec).mean().item()) if v_rec else float("nan"),
            "Cross Entropy Loss Train": _mean("sup_loss", only_supervised=True),
            "Cross Entropy Loss Valid": float(torch.stack(v_ce).mean().item()) if v_ce else float("nan"),
            "Discriminator Loss": _mean("d_loss"),
            "Total Loss Train": _mean("tot_loss", only_supervised=True),
            "Train Accuracy": float(np.mean(ys == y_hats)),
            "Valid Accuracy": float((torch.cat(v_y) == torch.cat(v_hat)).float().mean().item()) if v_y else 0.0,
        }
        if world > 1:
            # every entry is a mean over equally sized shards: the global value is the mean over the ranks
            import torch.distributed as dist
            vals = torch.tensor(list(record.values()), dtype=torch.float64, device=dev)
            dist.all_reduce(vals, group=process_group)
            record = {k: float(v) / world for k, v in zip(record, vals.tolist())}
        history.append(record)
        if timing is not None:
            # (every value of ``record`` is on the host: the validation pass has finished)
            timing.append({"epoch": epoch, "train_steps": len(steps), "train_s": t_train - t_epoch,
                           "valid_s": time.perf_counter() - t_train, "valid_batches": len(v_ce)})
        if rank == 0:                    # logging is rank 0's
            if log_fn is not None:
                log_fn(record)
            elif wb is not None and hasattr(wb, "log"):
                wb.log(record)
            print(f"[Epoch {epoch}/{config['EPOCHS']}] " + " ".join(f"[{k}: {v:.4f}]" for k, v in record.items()))

        if epoch % config["CHECKPOINT_FREQUENCY"] == 0 and record["Valid Accuracy"] > best_valid_accuracy:
            best_valid_accuracy = record["Valid Accuracy"]
            # every rank takes part (variant 1 evaluates the mean learner: identical replicas, and with SyncBN a
            # collective), rank 0 writes the files
            trainer.save_checkpoints(os.path.join("models", config["MODEL_NAME"]), config["MODEL_NAME"],
                                     write=rank == 0)
    run.finish()
    return trainer, history


def train_variant4(config, wandb_mode="online", proj_head_on_discriminator=False, **kw):
    """Variant 4 = the PCAA model of the paper (PCAA_ablation.py:746)."""
    if proj_head_on_discriminator:
        raise NotImplementedError(
            "proj_head_on_discriminator=True is never used by the reference's drivers and would fail "
            "there too (Linear(64,32) applied to the 32-wide sup_fvs, PCAA_ablation.py:783-786, :934)")
    return _run_loop(config, "v4", **kw)


def train_variant1(config, wandb_mode="online", **kw):
    """Variant 1: the prior centroids come from a GaussianMeanLearner (PCAA_ablation.py:28-378); see
    PCAATrainer for what the reference executes vs. what the variant intends."""
    return _run_loop(config, "v1", **kw)


def train_CGAAE(config=None, **kw):
    """Base conditional-Gaussian AAE loop (train_AAE.py:25)."""
    return _run_loop(constants.CONFIG if config is None else config, "base", **kw)


def train_variant2(config, wandb_mode="online", **kw):
    """Variant 2 == train_CGAAE with the supervision frequency forced to 1 -- in the caller's dict, as the
    reference does (PCAA_ablation.py:381-389)."""
    config["SUPERVISION_FREQUENCY"] = 1
    return train_CGAAE(config, **kw)


def train_variant3(config, wandb_mode="online", **kw):
    """Variant 3: the conditional-Gaussian AAE without the decoder (PCAA_ablation.py:392-743): encoder without
    projection head + critic only, ``tot = loss_g (+ sup_loss)``, optimizer_G's betas are ``(B1, B1)`` (:455),
    the reconstruction losses are logged as nan (``np.mean([])``), checkpoints are ``_E.pt``, ``_D.pt`` and
    ``discriminator_means.pt``.  (The reference builds the encoder with the DEFAULT ``nmax_points`` (:404-409),
    i.e. it only works when ``config["NMAX"] == constants.NMAX``; here the encoder follows ``config["NMAX"]``.)"""
    return _run_loop(config, "v3", **kw)


def train_pointsubsampling(n_training_classes=(2, 4, 6, 8), n_points_subs=(50, 70, 90, 110, 130, 150), n_tests=5,
                           ks=(1, 2, 4, 6), model_name_base="PCAA_npts_V4_", splits_seed=0, config=None):
    """The point-subsampling study of the reference's ``train_pointsubsampling.py`` (its ``__main__``, :19-76;
    BASELINE config[3]): for every number of training classes, ``n_tests`` distinct random class subsets
    (``default_rng(splits_seed).choice``, as there); for every NMAX the splits are regenerated from the raw
    tracks with that many points per frame, variant 4 is trained and evaluated open-set for each k.
    Returns {model_name: CGAAE_inference's log}."""
    from .datasets import LABEL_DICT, generate_splits
    from .inference import CGAAE_inference
    from .utils import openness
    rng = np.random.default_rng(splits_seed)
    results = {}
    for n_tr in n_training_classes:
        chosen = []
        for i in range(n_tests):
            while True:
                classes = sorted(int(c) for c in rng.choice(len(LABEL_DICT), n_tr, replace=False))
                if classes not in chosen:
                    chosen.append(classes)
                    break
            cfg = dict(constants.CONFIG if config is None else config)
            cfg["TRAIN_CLASSES"] = classes
            cfg["Openness"] = openness(n_tr, len(LABEL_DICT))
            for n_points in n_points_subs:
                cfg["NMAX"] = n_points
                generate_splits(train_classes=classes, seed=0, nmax_points=n_points, verbose=False)
                name = f"{model_name_base}{n_points}.{n_tr}.{i + 1}"
                cfg["MODEL_NAME"] = name
                cfg["NOTES"] = f"Runs with different number of points ({n_points}.{n_tr}.{i + 1})"
                train_variant4(dict(cfg), wandb_mode="disabled", proj_head_on_discriminator=False)
                results[name] = CGAAE_inference(model_names=[name], ks=list(ks), variation="V4",
                                                generate_dataset=False)
    return results

