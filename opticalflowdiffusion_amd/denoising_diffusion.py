"""`Unet` and `ConditionalDiffusion` with the reference's interface
(algorithms/diffusion_animation/denoising_diffusion.py, "DD"), executed by libofd_hip.

`Unet(dim, channels=, out_dim=, time_in=True)` owns fp32 `nn.Parameter`s under the reference's
state-dict names (DD:272-361), so reference checkpoints load with `load_state_dict`.  Its
forward is ONE C call (`ofd_unet_forward`) that runs the hand-written HIP kernels; there is no
PyTorch fallback.  `ConditionalDiffusion` restates DD:463-993 for the configuration FlowDiffuser
builds (FD:118-127) with the elementwise steps fused into single HIP kernels, generalised to
non-square `image_size=(H, W)` and with a working DDIM path.
"""
import ctypes
import math
from collections import namedtuple

import torch
from torch import nn

from . import _lib as L
from .warp import nan_mse

ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start", "additional_out"])


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def identity(t, *args, **kwargs):
    return t


class _Node(nn.Module):
    """Parameter container that reproduces the reference's module tree in state-dict keys."""


def _registry(dim, channels, out_dim, eps_mode, no_time=0):
    """(handle, [(name, shape)]) from the C library, which owns the layer table."""
    lib = L.lib()
    cfg = L.UnetConfig(dim, channels, out_dim, eps_mode, no_time)
    h = ctypes.c_void_p()
    L.check(lib.ofd_unet_create(ctypes.byref(cfg), ctypes.byref(h)))
    names = []
    dims = (ctypes.c_int * 4)()
    for i in range(lib.ofd_unet_num_params(h)):
        nd = lib.ofd_unet_param_shape(h, i, dims)
        names.append((lib.ofd_unet_param_name(h, i).decode(), tuple(dims[k] for k in range(nd))))
    return h, names


class _UnetTrain(torch.autograd.Function):
    """Unet.forward under autograd: the HIP training forward keeps the tape inside the executor's
    workspace, backward() replays it and hands the parameter gradients back as views of the
    executor's flat gradient buffer (already all-reduced when a grad_sync is attached)."""

    @staticmethod
    def forward(ctx, unet, x, cond, t, *params):
        ctx.unet = unet
        ctx.ticket = unet._train_forward(x, cond, t)
        return unet._train_out

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.unet._backward(ctx.ticket, gout)
        return (None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[4:]))


class Unet(nn.Module):
    """DD:272-417.  Supported: dim=64, dim_mults=(1,2,4,8), no self-conditioning -- the UNet FlowDiffuser instantiates
    (FD:106-111), with time_in=True (diffusion) or time_in=False (is_diffusion=False, and FlowLearner's regression UNet)."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, time_in=True, precision="bf16"):
        super().__init__()
        if (dim != 64 or tuple(dim_mults) != (1, 2, 4, 8) or self_condition or learned_variance or learned_sinusoidal_cond
                or random_fourier_features or resnet_block_groups != 8 or init_dim not in (None, dim)):
            raise NotImplementedError("the HIP engine implements the FlowDiffuser UNet: Unet(64, channels=, out_dim=, time_in=)")
        self.channels = channels
        self.self_condition = False
        self.time_in = bool(time_in)
        self.random_or_learned_sinusoidal_cond = False
        self.out_dim = default(out_dim, channels)
        self.dim = dim
        # eps_mode 1: per-site eps of the reference under bf16 autocast (DD:107,122); 0: fp32 rule
        self.eps_mode = 1 if precision == "bf16" else 0
        self._handle, reg = _registry(dim, channels, self.out_dim, self.eps_mode, 0 if self.time_in else 1)
        self._names = [n for n, _ in reg]
        gen = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        fan_in = 1
        for name, shape in reg:
            if name.endswith(".weight") and len(shape) > 1:
                fan_in = 1
                for s_ in shape[1:]:
                    fan_in *= s_
            self._add(name, nn.Parameter(self._init(name, shape, gen, fan_in)))
        self._synced = None
        self._pflat = None
        self._poffsets = None
        self._ws = None
        self._train_ws = None
        self._gflat = None
        self._ticket = 0
        self.grad_sync = None          # parallel.BucketedAllReduce for data-parallel training (or None)

    # -- parameter tree ----------------------------------------------------------------------
    def _add(self, name, param):
        node = self
        parts = name.split(".")
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        node.register_parameter(parts[-1], param)

    @staticmethod
    def _init(name, shape, gen, fan_in):
        """PyTorch default initialisation of the corresponding reference layers (Conv2d / Linear:
        kaiming_uniform_(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weight and bias; norms: ones / zeros)."""
        if name.endswith(".g") or name.endswith("norm.weight"):
            return torch.ones(shape)
        if name.endswith("norm.bias"):
            return torch.zeros(shape)
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound

    def _param(self, name):
        node = self
        for p in name.split("."):
            node = node._modules[p] if p in node._modules else node._parameters[p]
        return node

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                L.lib().ofd_unet_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # -- execution ---------------------------------------------------------------------------
    def _sync_params(self, device):
        """The executor reads the parameters in place: every nn.Parameter is a view of one flat fp32 device
        tensor laid out like the executor's registry (bound once, zero copies per step).  Whenever a
        parameter's version changed (optimizer step, load_state_dict, manual edit) only the weight
        preparation (standardise + bf16 pack) is re-run."""
        lib = L.lib()
        params = [self._param(n) for n in self._names]
        if self._poffsets is None:
            self._poffsets = [lib.ofd_unet_param_offset(self._handle, i) for i in range(len(self._names))]
        flat = self._pflat
        intact = flat is not None and flat.device == device
        if intact:
            base = flat.data_ptr()
            intact = all(p.data_ptr() == base + 4 * off and p.dtype == torch.float32 for p, off in zip(params, self._poffsets))
        if not intact:
            flat = torch.zeros(lib.ofd_unet_param_floats(self._handle), dtype=torch.float32, device=device)
            for p, off in zip(params, self._poffsets):
                view = flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.detach())
                p.data = view
            self._pflat = flat
            L.check(lib.ofd_unet_bind_param_buffer(self._handle, L.ptr(flat), flat.numel()))
            self._synced = None
        key = tuple(p._version for p in params)
        if self._synced != key:
            L.check(lib.ofd_unet_prepare(self._handle, L.stream()))
            self._synced = key

    def _workspace(self, device, B, H, W):
        need = L.lib().ofd_unet_workspace_bytes(self._handle, B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, x, external_cond=None, time=None, x_self_cond=None, additional_out=False):
        if additional_out:
            raise ValueError("additional tgt not supported for non warp Unet")             # DD:364-365
        if self.time_in and time is None:
            raise ValueError("when Unet takes time arg, time argument must be passed in")  # DD:378-379
        if not self.time_in and time is not None:
            raise ValueError("this Unet does not take time arg")                           # DD:382-383
        L.require_gpu(x, external_cond, time)
        if x.requires_grad or (external_cond is not None and external_cond.requires_grad):
            raise L.OfdError("Unet.forward: gradients w.r.t. the inputs are not produced (the training step never needs them)")
        x = L.f32c(x)
        cond = L.f32c(external_cond) if external_cond is not None else None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _UnetTrain.apply(self, x, cond, time.to(torch.int64).contiguous() if self.time_in else None,
                                    *[self._param(n) for n in self._names])
        B, Cx, H, W = x.shape
        Cc = cond.shape[1] if cond is not None else 0
        t = time.to(torch.int64).contiguous() if self.time_in else None
        self._sync_params(x.device)
        ws = self._workspace(x.device, B, H, W)
        out = torch.empty(B, self.out_dim, H, W, dtype=torch.float32, device=x.device)
        L.check(L.lib().ofd_unet_forward(self._handle, L.ptr(x), Cx, L.ptr(cond), Cc, L.ptr(t), L.ptr(out), B, H, W,
                                         L.ptr(ws), ws.numel(), L.stream()))
        return out

    # -- training step (ofd_unet_train_forward / ofd_unet_backward) -------------------------------
    def flat_grads(self, device=None):
        """the flat fp32 gradient buffer the executor writes (parameter i at [offset_i, offset_i + numel_i))."""
        lib = L.lib()
        if self._gflat is None or (device is not None and self._gflat.device != device):
            if device is None:
                raise L.OfdError("no gradient buffer yet: run a training forward first")
            self._gflat = torch.zeros(lib.ofd_unet_param_floats(self._handle), dtype=torch.float32, device=device)
            L.check(lib.ofd_unet_bind_grad_buffer(self._handle, L.ptr(self._gflat), self._gflat.numel()))
            self._goffsets = [lib.ofd_unet_param_offset(self._handle, i) for i in range(len(self._names))]
        return self._gflat

    def _train_forward(self, x, cond, t):
        lib = L.lib()
        B, Cx, H, W = x.shape
        Cc = cond.shape[1] if cond is not None else 0
        self._sync_params(x.device)
        self.flat_grads(x.device)
        need = lib.ofd_unet_train_workspace_bytes(self._handle, B, H, W)
        if need == 0:
            raise L.OfdError(f"training workspace planning failed for B={B} H={H} W={W}: {L.last_error()}")
        if self._train_ws is None or self._train_ws.numel() < need or self._train_ws.device != x.device:
            self._train_ws = None
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        out = torch.empty(B, self.out_dim, H, W, dtype=torch.float32, device=x.device)
        L.check(lib.ofd_unet_train_forward(self._handle, L.ptr(x), Cx, L.ptr(cond), Cc, L.ptr(t), L.ptr(out), B, H, W,
                                           L.ptr(self._train_ws), self._train_ws.numel(), L.stream()))
        self._train_t = t                  # the backward re-reads the timesteps
        self._train_out = out
        self._ticket += 1
        return self._ticket

    def _backward(self, ticket, gout):
        if ticket != self._ticket:
            raise L.OfdError("Unet backward: the tape belongs to an older forward (one training forward per backward; "
                             "retain_graph / double backward are not supported)")
        lib = L.lib()
        flat = self.flat_grads()
        gout = L.f32c(gout)
        sync, err = self.grad_sync, []

        def ready(begin, end, _user):
            try:
                if sync is not None:
                    sync.on_range(flat, begin, end)
            except BaseException as e:      # must not propagate through the C frame
                err.append(e)

        cb = L.GRAD_READY(ready)
        if sync is not None:
            sync.begin(flat)
        L.check(lib.ofd_unet_backward(self._handle, L.ptr(gout), cb, None, L.stream()))
        if err:
            raise err[0]
        if sync is not None:
            sync.finish(flat)
        self._ticket += 1                   # tape consumed
        # autograd gets views of ONE copy of the flat buffer (a single 143 MB device copy): handing out views of
        # the executor's own buffer would alias a surviving `.grad` from the previous step with the incoming
        # gradient (zero_grad(set_to_none=False) / gradient accumulation: `grad += grad` would double it)
        snap = flat.clone()
        grads = []
        for i, name in enumerate(self._names):
            p = self._param(name)
            off = self._goffsets[i]
            grads.append(snap[off:off + p.numel()].view(p.shape))
        return grads

    def read_tap(self, name, shape):
        """named intermediate of the last forward as NCHW fp32 (parity tests)."""
        out = torch.empty(shape, dtype=torch.float32, device=self._ws.device)
        L.check(L.lib().ofd_unet_read_tap(self._handle, name.encode(), L.ptr(out), out.numel(), L.stream()))
        return out

    def set_debug_taps(self, enabled=True):
        """materialise every named intermediate of the inference forward for `read_tap` (off: `final_res_block`'s output is never written,
        the final 1x1 conv rides on its producer's tile)"""
        L.check(L.lib().ofd_unet_set_debug_taps(self._handle, int(enabled)))

    def set_graph(self, enabled=True):
        """replay the inference forward as one hipGraph per (shape, stream) instead of ~250 launches (launch-bound
        regimes: small images, long sampling loops); bit-identical, ignored while profiling"""
        L.check(L.lib().ofd_unet_set_graph(self._handle, int(enabled)))

    def set_split_streams(self, enabled=True, offset_blocks=-1):
        """inference forward of an even batch as two half-batches on two streams, the second `offset_blocks` blocks behind the
        first (HBM-bound kernels of one half overlap the MFMA-bound kernels of the other); bit-identical per sample.
        enabled: True / False, or None for the engine's default (on for batches of at least 2^21 pixels in all)"""
        L.check(L.lib().ofd_unet_set_split_streams(self._handle, -1 if enabled is None else int(bool(enabled)), int(offset_blocks)))

    # -- per-kernel-class device timing (HIP events on the launch stream) ----------------------
    def set_deterministic(self, enabled=True):
        """order-independent gradient accumulation in the backward (csrc/det.h: 64-bit fixed-point shadows instead of float atomics): two
        backward passes over the same inputs give bit-identical parameter gradients.  Default: the environment variable OFD_DETERMINISTIC."""
        L.check(L.lib().ofd_unet_set_deterministic(self._handle, int(enabled)))

    def deterministic_misses(self):
        """accumulations that found no fixed-point shadow and fell back to float atomics since the handle was created (0 expected)"""
        return int(L.lib().ofd_unet_deterministic_misses(self._handle))

    def set_profiling(self, enabled, dump_path=None):
        L.check(L.lib().ofd_unet_set_profiling(self._handle, int(enabled)))
        L.check(L.lib().ofd_unet_prof_dump_path(self._handle, dump_path.encode() if dump_path else None))

    def profile(self, reset=False):
        lib = L.lib()
        res = {}
        for i in range(lib.ofd_unet_prof_count(self._handle)):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
            L.check(lib.ofd_unet_prof_read(self._handle, i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
            res[lib.ofd_unet_prof_name(self._handle, i).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        if reset:
            L.check(lib.ofd_unet_prof_reset(self._handle))
        return res


# ------------------------------------------------------------------------------------ schedules
def extract(a, t, x_shape):
    """DD:422-425."""
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1, clamp_min=1e-5):
    """DD:448-461."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    alphas_cumprod = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)


def linear_beta_schedule(timesteps):
    """DD:427-434."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    """DD:436-446."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


class ConditionalDiffusion(nn.Module):
    """DD:463-993 for objective='pred_x0', noise_space='image' (what FlowDiffuser builds)."""

    def __init__(self, model, image_size, timesteps=1000, sampling_timesteps=None, objective="pred_v",
                 beta_schedule="sigmoid", schedule_fn_kwargs=dict(), ddim_sampling_eta=0.0, auto_normalize=True,
                 offset_noise_strength=0.0, min_snr_loss_weight=False, min_snr_gamma=5, conditioned=True,
                 channels=3, noise_space="image", ddim_draw_unused_noise=False):
        super().__init__()
        # ddim_draw_unused_noise (not in the reference): the reference's ddim_sample draws randn_like(img) every step even when
        # eta == 0 multiplies it by zero (DD:763); the engine skips that draw, so a SEEDED eta == 0 run consumes a different RNG
        # stream.  True restores the draw (same stream positions as the reference) for seeded comparisons.
        self.ddim_draw_unused_noise = bool(ddim_draw_unused_noise)
        if objective != "pred_x0" or noise_space != "image" or auto_normalize or offset_noise_strength != 0.0:
            raise NotImplementedError("HIP path: objective='pred_x0', noise_space='image', auto_normalize=False (FD:118-127); "
                                      "noise_space='flow' is broken in the reference itself (warp.py:181-182)")
        self.model = model
        self.channels = channels
        self.self_condition = False
        self.conditioned = conditioned
        self.noise_space = noise_space
        self.image_size = image_size                      # int (square, as the reference) or (H, W)
        self.objective = objective
        fn = {"linear": linear_beta_schedule, "cosine": cosine_beta_schedule, "sigmoid": sigmoid_beta_schedule}.get(beta_schedule)
        if fn is None:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        betas = fn(timesteps, **schedule_fn_kwargs)
        alphas = 1.0 - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = torch.cat((torch.ones(1, dtype=torch.float64), alphas_cumprod[:-1]))
        (timesteps,) = betas.shape
        self.num_timesteps = int(timesteps)
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        # optional: with return_all_timesteps keep x_T, every `trajectory_stride`-th step and the final sample instead of all T+1
        # frames (1001 x 57.7 MB = 57.8 GB at B=16, 440x1024; the reference's logging only looks at samples[:, ::50], FD:246).
        # None = the reference's behaviour (every frame).
        self.trajectory_stride = None
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta

        def reg(name, val):
            self.register_buffer(name, val.to(torch.float32))

        reg("betas", betas)
        reg("alphas_cumprod", alphas_cumprod)
        reg("alphas_cumprod_prev", alphas_cumprod_prev)
        reg("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        reg("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - alphas_cumprod))
        reg("log_one_minus_alphas_cumprod", torch.log(1.0 - alphas_cumprod))
        reg("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / alphas_cumprod))
        reg("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / alphas_cumprod - 1))
        posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
        reg("posterior_variance", posterior_variance)
        reg("posterior_log_variance_clipped", torch.log(posterior_variance.clamp(min=1e-20)))
        reg("posterior_mean_coef1", betas * torch.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod))
        reg("posterior_mean_coef2", (1.0 - alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - alphas_cumprod))
        self.offset_noise_strength = offset_noise_strength
        snr = alphas_cumprod / (1 - alphas_cumprod)
        clipped = snr.clone()
        if min_snr_loss_weight:
            clipped.clamp_(max=min_snr_gamma)
        reg("loss_weight", clipped)                        # pred_x0 (DD:575-576)
        self.normalize = identity
        self.unnormalize = identity

    @property
    def device(self):
        return self.betas.device

    def _hw(self):
        s = self.image_size
        return (s, s) if isinstance(s, int) else tuple(s)

    # -- network call ------------------------------------------------------------------------
    def model_with_condition(self, x, t, x_self_cond, external_cond=None, additional_tgt=None):
        assert self.conditioned == torch.is_tensor(external_cond)                      # DD:627
        return self.model(x, external_cond if self.conditioned else None, t, x_self_cond,
                          additional_out=additional_tgt is not None)

    def predict_noise_from_start(self, x_t, t, x0):
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - x0) / \
            extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def model_predictions(self, x, t, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False,
                          external_cond=None, additional_tgt=None):
        """DD:634-664 (pred_x0 branch)."""
        out = self.model_with_condition(x, t, x_self_cond, external_cond=external_cond, additional_tgt=additional_tgt)
        additional_out = None
        if additional_tgt is not None:
            additional_out = out[:, -1 * additional_tgt.shape[1]:]
            out = out[:, :-1 * additional_tgt.shape[1]]
        x_start = torch.clamp(out, min=-1.0, max=1.0) if clip_x_start else out
        pred_noise = self.predict_noise_from_start(x, t, x_start)
        return ModelPrediction(pred_noise, x_start, additional_out)

    def q_sample(self, x_start, t, noise=None):
        """DD:806-812 as one fused HIP kernel."""
        noise = default(noise, lambda: torch.randn_like(x_start))
        x0, nz = L.f32c(x_start), L.f32c(noise)
        out = torch.empty_like(x0)
        a, b = self.sqrt_alphas_cumprod[t].contiguous(), self.sqrt_one_minus_alphas_cumprod[t].contiguous()
        B = x0.shape[0]
        L.check(L.lib().ofd_q_sample(L.ptr(x0), L.ptr(nz), L.ptr(a), L.ptr(b), L.ptr(out), B, x0[0].numel(), L.stream()))
        return out

    # -- DDPM --------------------------------------------------------------------------------
    @torch.no_grad()
    def p_sample(self, x, t: int, x_self_cond=None, external_cond=None, additional_tgt=None, noise=None):
        """DD:676-698: network call + one fused kernel for clamp / posterior mean / noise add."""
        b = x.shape[0]
        tab = self._sampling_tables(b, x.device)           # rows of per-(T, batch) tables: no fill / gather / exp launches per step
        bt = tab["t"][t]
        out = self.model_with_condition(x, bt, x_self_cond, external_cond=external_cond, additional_tgt=additional_tgt)
        additional_out = None
        if additional_tgt is not None:
            additional_out = out[:, -1 * additional_tgt.shape[1]:]
            out = L.f32c(out[:, :-1 * additional_tgt.shape[1]])
        x = L.f32c(x)
        c1, c2, sigma = tab["c1"][t], tab["c2"][t], tab["sigma"][t]
        if t > 0:
            noise = default(noise, lambda: torch.randn_like(x))
        else:
            noise = None                                                               # DD:687
        pred = torch.empty_like(x)
        x_start = torch.empty_like(x)
        L.check(L.lib().ofd_ddpm_update(L.ptr(x), L.ptr(out), L.ptr(noise), L.ptr(c1), L.ptr(c2), L.ptr(sigma),
                                        L.ptr(pred), L.ptr(x_start), b, x[0].numel(), L.stream()))
        return pred, x_start, additional_out

    def _sampling_tables(self, batch, device):
        """per-(T, batch) views of everything a reverse step reads that does not depend on the data: timestep tensors and the
        posterior coefficients, expanded once so that a step indexes a ROW (a view, no gather launch, no allocation)"""
        # the key carries the identity AND version of every schedule buffer a row is derived from: load_state_dict of another
        # schedule, a dtype / device move or an in-place edit of a buffer rebuilds the tables instead of serving stale rows
        srcs = (self.posterior_mean_coef1, self.posterior_mean_coef2, self.posterior_log_variance_clipped,
                self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod)
        key = (batch, str(device)) + tuple((b.data_ptr(), b._version, b.dtype) for b in srcs)
        if getattr(self, "_samp_tab", None) is None or self._samp_tab[0] != key:
            T = self.num_timesteps
            rep = lambda v: v.to(device=device, dtype=torch.float32).reshape(T, 1).repeat(1, batch).contiguous()
            sigma = (0.5 * self.posterior_log_variance_clipped).exp()
            self._samp_tab = (key, dict(t=torch.arange(T, device=device, dtype=torch.long).reshape(T, 1).repeat(1, batch).contiguous(),
                                        c1=rep(self.posterior_mean_coef1), c2=rep(self.posterior_mean_coef2), sigma=rep(sigma),
                                        sr=rep(self.sqrt_recip_alphas_cumprod), srm1=rep(self.sqrt_recipm1_alphas_cumprod)))
        return self._samp_tab[1]

    @torch.no_grad()
    def p_sample_loop(self, shape, return_all_timesteps=False, external_cond=None, additional_tgt=None, verbose=False, x_T=None):
        """DD:700-729 (no per-step print / host sync).  `x_T` (optional, not in the reference): the start of the chains (DD:705).
        A step is: one UNet call, one in-place normal_ into a reused buffer, one fused update kernel writing into the other of
        two ping-pong images -- no per-step allocation, no coefficient gathers (rows of `_sampling_tables`)."""
        img = torch.randn(shape, device=self.device) if x_T is None else L.f32c(x_T)
        assert tuple(img.shape) == tuple(shape)
        if additional_tgt is not None:                                                # target='target': the general step (DD:676-698)
            imgs, additionals = [img], [None]
            for i, t in enumerate(reversed(range(0, self.num_timesteps))):
                img, _, additional_out = self.p_sample(img, t, None, external_cond=external_cond, additional_tgt=additional_tgt)
                if return_all_timesteps:
                    imgs.append(img)
                additionals.append(additional_out)
            return (img if not return_all_timesteps else torch.stack(imgs, dim=1)), additionals
        imgs = [img]
        stride = self.trajectory_stride
        tab = self._sampling_tables(shape[0], img.device)
        b, n = shape[0], img[0].numel()
        pong = [torch.empty_like(img), torch.empty_like(img)]
        noise, x_start = torch.empty_like(img), torch.empty_like(img)
        lib = L.lib()
        for i, t in enumerate(reversed(range(0, self.num_timesteps))):
            out = L.f32c(self.model_with_condition(img, tab["t"][t], None, external_cond=external_cond))
            if t > 0:
                noise.normal_()                                                       # DD:687: z = 0 at t = 0
            nxt = pong[i & 1]
            L.check(lib.ofd_ddpm_update(L.ptr(img), L.ptr(out), L.ptr(noise) if t > 0 else None, L.ptr(tab["c1"][t]), L.ptr(tab["c2"][t]),
                                        L.ptr(tab["sigma"][t]), L.ptr(nxt), L.ptr(x_start), b, n, L.stream()))
            img = nxt
            if return_all_timesteps and (stride is None or (i + 1) % stride == 0 or t == 0):
                imgs.append(img.clone())
        return img if not return_all_timesteps else torch.stack(imgs, dim=1)

    # -- DDIM --------------------------------------------------------------------------------
    @torch.no_grad()
    def ddim_sample(self, shape, return_all_timesteps=False, external_cond=None, additional_tgt=None, x_T=None):
        """DD:731-774; accepts (and ignores) additional_tgt so that sample() can reach it.  `x_T` (optional, not in
        the reference) starts the chains from a given tensor instead of a fresh draw (DD:741)."""
        batch, device, T, S, eta = shape[0], self.device, self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        time_pairs = list(zip(times[:-1], times[1:]))
        img = torch.randn(shape, device=device) if x_T is None else L.f32c(x_T)
        assert tuple(img.shape) == tuple(shape)
        imgs = [img]
        n = img[0].numel()
        stride = self.trajectory_stride
        tab = self._sampling_tables(batch, img.device)
        # the per-pair scalars of DD:757-761 for ALL pairs at once, in the reference's fp32 operation order -> one (S, 3, batch) table
        ac = self.alphas_cumprod
        tt = torch.tensor([p[0] for p in time_pairs], device=ac.device)
        tn = torch.tensor([max(p[1], 0) for p in time_pairs], device=ac.device)
        alpha, alpha_next = ac[tt], ac[tn]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        coef = torch.stack((alpha_next.sqrt(), c, sigma), dim=1).to(torch.float32).reshape(len(time_pairs), 3, 1).repeat(1, 1, batch).contiguous()
        pong = [torch.empty_like(img), torch.empty_like(img)]
        noise = torch.empty_like(img) if (eta > 0 or self.ddim_draw_unused_noise) else None
        lib = L.lib()
        for i, (time, time_next) in enumerate(time_pairs):
            out = L.f32c(self.model_with_condition(img, tab["t"][time], None, external_cond=external_cond))
            last = time_next < 0
            if noise is not None and not last:
                noise.normal_()                                                          # DD:763 (eta == 0: only with ddim_draw_unused_noise)
            nxt = pong[i & 1]
            L.check(lib.ofd_ddim_update(L.ptr(img), L.ptr(out), L.ptr(noise) if not last else None, L.ptr(tab["sr"][time]), L.ptr(tab["srm1"][time]),
                                        None if last else L.ptr(coef[i, 0]), None if last else L.ptr(coef[i, 1]), None if last else L.ptr(coef[i, 2]),
                                        int(last), L.ptr(nxt), None, batch, n, L.stream()))
            img = nxt
            if return_all_timesteps and (stride is None or (i + 1) % stride == 0 or last):
                imgs.append(img.clone())
        return img if not return_all_timesteps else torch.stack(imgs, dim=1)

    @torch.no_grad()
    def sample(self, batch_size=16, return_all_timesteps=False, external_cond=None, additional_tgt=None):
        """DD:776-784, with image_size allowed to be (H, W)."""
        H, W = self._hw()
        fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        assert external_cond is None or external_cond.shape[0] == batch_size
        return fn((batch_size, self.channels, H, W), return_all_timesteps=return_all_timesteps,
                  external_cond=external_cond, additional_tgt=additional_tgt)

    # -- training loss -----------------------------------------------------------------------
    def p_losses(self, x_start, t, noise=None, offset_noise_strength=None, external_cond=None, additional_tgt=None,
                 additional_weight=None, model_out_override=None):
        """DD:823-891."""
        noise = default(noise, lambda: torch.randn_like(x_start))
        x = self.q_sample(x_start=x_start, t=t, noise=noise)
        if model_out_override is None:
            model_out_full = self.model_with_condition(x, t, None, external_cond=external_cond, additional_tgt=additional_tgt)
            model_out = model_out_full
            if additional_tgt is not None:
                model_out = model_out_full[:, :-1 * additional_tgt.shape[1]]
        else:
            model_out, _ = model_out_override
        target = x_start                                                               # pred_x0 (DD:876-877)
        if additional_tgt is not None:                                                 # target='target' (DD:884-885)
            additional_out = model_out_full[:, -1 * additional_tgt.shape[1]:] if model_out_override is None else model_out_override[1]
            return self._loss(model_out, target, t, additional_tgt, external_cond, additional_out, additional_weight)
        if target.shape[1] == 5:                                                       # target='joint' (DD:886-887)
            return self._loss(model_out[:, :3], target[:, :3], t, target[:, 3:], external_cond, model_out[:, 3:], 0.0)
        return self._loss(model_out[:, :3], target[:, :3], t)

    def _loss(self, image_out, target, t=None, flow_tgt=None, external_cond=None, flow_out=None, additional_weight=None):
        """DD:893-983.  Level 1: NaN-masked squared error of the (warped) image.  With a flow target the
        reference adds pyramid levels 2, 4, 8, 16: the condition image splatted by the PREDICTED flow at
        1/level resolution (`self.model._warp(cond, flow_out, scale=level)`) against the target image
        splatted by zero flow at the same scale, weighted level^4; the loss is the `nanmean` of the
        concatenation of all levels = sum_L L^4 S_L / sum_L N_L.  (The flow-MSE term, the SNR weighting,
        anomaly mode and the prints are disabled / dropped as in the reference, DD:963-980.)  Every piece
        is a HIP kernel with its own backward: splat (forward, d/dflow), NaN-masked reductions."""
        from .warp import nan_sq_sum
        if flow_tgt is None:
            return nan_mse(image_out, target, reduction="mean")
        levels = [1, 2, 4, 8, 16]                                                      # DD:896
        s1, n1 = nan_sq_sum(image_out, target)
        num, den = s1, n1
        self.last_levels = [(1, s1.detach(), n1)]                                      # (level, S_L, N_L): device scalars, no host sync
        for level in levels[1:]:
            image_out_ = self.model._warp(external_cond, flow_out, scale=level)       # DD:936
            with torch.no_grad():
                image_out_tgt = self.model._warp(target, torch.zeros_like(flow_out), scale=level)   # DD:941
            s, n = nan_sq_sum(image_out_, image_out_tgt)
            num = num + s * float(level ** 4)                                          # DD:956
            den = den + n
            self.last_levels.append((level, s.detach(), n))
        return num / den.float()

    def forward(self, img, external_cond=None, *args, **kwargs):
        """DD:985-993."""
        b, c, h, w = img.shape
        H, W = self._hw()
        assert h == H and w == W, f"height and width of image must be {(H, W)}"
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        return self.p_losses(img, t, external_cond=external_cond, *args, **kwargs)
