"""TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of ConditionalDiffusion (DD:463-993).

Only the configuration FlowDiffuser builds is restated (FD:118-127): sigmoid beta
schedule, objective ``pred_x0``, ``auto_normalize=False``, ``noise_space='image'``,
``min_snr_loss_weight=True``.  Noise is always passed in so results are reproducible.
"""
import math

import torch

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight",
)


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1):
    """DD:448-461 (float64)."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def make_schedule(timesteps, min_snr_gamma=5.0):
    """DD:511-578: the 13 registered buffers, float64 math stored as float32 (DD:530)."""
    betas = sigmoid_beta_schedule(timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat((torch.ones(1, dtype=torch.float64), ac[:-1]))
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    snr = ac / (1 - ac)
    S = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "loss_weight": snr.clamp(max=min_snr_gamma),          # pred_x0: DD:575-576
    }
    return {k: v.to(torch.float32) for k, v in S.items()}


def _ex(a, t):
    """DD:422-425 extract for 4-D tensors."""
    return a.gather(-1, t).reshape(-1, 1, 1, 1)


def q_sample(S, x0, t, noise):
    """DD:806-812."""
    return _ex(S["sqrt_alphas_cumprod"], t) * x0 + _ex(S["sqrt_one_minus_alphas_cumprod"], t) * noise


def predict_noise_from_start(S, x_t, t, x0):
    """DD:595-599."""
    return (_ex(S["sqrt_recip_alphas_cumprod"], t) * x_t - x0) / _ex(S["sqrt_recipm1_alphas_cumprod"], t)


def p_sample_update(S, x_t, t_int, model_out, noise):
    """One reverse step after the network call: DD:653-656 (pred_x0), DD:670-671 clamp,
    DD:613-623 posterior, DD:686-688 update.  ``noise`` is ignored at t == 0.
    Returns (x_{t-1}, x_start)."""
    b = x_t.shape[0]
    t = torch.full((b,), t_int, dtype=torch.long)
    x0 = model_out.clamp(-1.0, 1.0)
    mean = _ex(S["posterior_mean_coef1"], t) * x0 + _ex(S["posterior_mean_coef2"], t) * x_t
    logvar = _ex(S["posterior_log_variance_clipped"], t)
    if t_int > 0:
        return mean + (0.5 * logvar).exp() * noise, x0
    return mean, x0


def ddim_times(num_timesteps, sampling_timesteps):
    """DD:737-739: reversed (time, time_next) pairs."""
    times = torch.linspace(-1, num_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_update(S, x_t, time, time_next, model_out, noise, eta=0.0):
    """DD:750-767 for pred_x0 with clip_x_start + rederive_pred_noise. Returns (x_next, x_start)."""
    b = x_t.shape[0]
    t = torch.full((b,), time, dtype=torch.long)
    x0 = model_out.clamp(-1.0, 1.0)
    pred_noise = predict_noise_from_start(S, x_t, t, x0)
    if time_next < 0:
        return x0, x0
    alpha = S["alphas_cumprod"][time]
    alpha_next = S["alphas_cumprod"][time_next]
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    return x0 * alpha_next.sqrt() + c * pred_noise + sigma * noise, x0


def nan_mse_none(pred, target):
    """WP:260-271 with reduction='none': squared error over positions where neither is NaN."""
    pred = pred.flatten()
    target = target.flatten()
    ok = ~(torch.isnan(pred) | torch.isnan(target))
    return torch.square(pred[ok] - target[ok])


def p_losses_flow(S, model_fn, x0, cond, t, noise):
    """DD:823-891 + DD:893-983 for target='flow' (no pyramid levels, SNR weighting disabled
    at DD:975-980): nanmean of the level-1 NaN-masked squared error against x_start."""
    x = q_sample(S, x0, t, noise)
    out = model_fn(x, cond, t)
    return torch.nanmean(nan_mse_none(out[:, :3], x0[:, :3]))


def pyramid_loss(image_out, target, cond, flow_out, flow_max, dim=3, levels=(1, 2, 4, 8, 16)):
    """DD:893-983 with a flow target (target='joint' / 'target'): level 1 is the NaN-masked squared error of
    the warped image; level L > 1 compares `_warp(cond, flow_out, scale=L)` with `_warp(target, 0, scale=L)`
    (FD:35-36: `_warp(image, flow, **kw) = warp(image[:, :dim], None, flow * flow_max, mode='forward', **kw)`),
    weighted L^4 (DD:956); `nanmean` over the concatenation (DD:973).  Flow-MSE term and SNR weighting are
    disabled in the reference (DD:963-980)."""
    from . import warp_ref as WR
    loss = nan_mse_none(image_out, target)
    for level in levels[1:]:
        io = WR.warp(cond[:, :dim], None, flow_out * flow_max, mode="forward", scale=level)
        it = WR.warp(target[:, :dim], None, torch.zeros_like(flow_out) * flow_max, mode="forward", scale=level)
        loss = torch.cat((loss, nan_mse_none(io, it) * level ** 4), dim=0)
    return torch.nanmean(loss)
