"""DDPM step with background smoothing (src/pipe_FRESCO.py:14-77) and classifier-free guidance (:212-214)
with the reference's signature; the elementwise math runs in two fused HIP kernels."""
import torch

from . import _lib, ops
from .warp import warp_tensor


def _dt(t):
    if t.dtype == torch.float16:
        return _lib.F16
    if t.dtype == torch.float32:
        return _lib.F32
    raise TypeError("fresco_amd.step: fp16 / fp32 latents only (got %s)" % t.dtype)


def predict_x0(sample, eps_uncond, eps_text=None, guidance_scale=1.0, alpha_prod_t=1.0):
    """x0 (and the guided eps) from x_t: formula (12) of DDIM, with CFG fused in when eps_text is given."""
    ops._need_gpu(sample, eps_uncond, eps_text)
    sample, eps_uncond = sample.contiguous(), eps_uncond.contiguous()
    eps_text = None if eps_text is None else eps_text.contiguous()
    x0 = torch.empty_like(sample)
    eps = torch.empty_like(sample) if eps_text is not None else None
    a = float(alpha_prod_t)
    rc = _lib.load().fresco_ddpm_x0(sample.data_ptr(), eps_uncond.data_ptr(), ops._ptr(eps_text), x0.data_ptr(),
                                    ops._ptr(eps), sample.numel(), float(guidance_scale), (1.0 - a) ** 0.5, a ** 0.5,
                                    _dt(sample), ops._stream())
    _lib.check(rc, "fresco_ddpm_x0")
    return x0, (eps if eps is not None else eps_uncond)


@torch.no_grad()
def step(pipe, model_output, timestep, sample, generator, repeat_noise=False, visualize_pipeline=False,
         flows=None, occs=None, saliency=None):
    """Returns (pred_prev_sample, pred_original_sample) like the reference."""
    scheduler = pipe.scheduler
    prev_timestep = scheduler.previous_timestep(timestep)
    alpha_prod_t = float(scheduler.alphas_cumprod[timestep])
    alpha_prod_t_prev = float(scheduler.alphas_cumprod[prev_timestep]) if prev_timestep >= 0 else float(scheduler.one)
    beta_prod_t = 1 - alpha_prod_t
    beta_prod_t_prev = 1 - alpha_prod_t_prev
    current_alpha_t = alpha_prod_t / alpha_prod_t_prev
    current_beta_t = 1 - current_alpha_t

    pred_original_sample, _ = predict_x0(sample, model_output, alpha_prod_t=alpha_prod_t)
    if saliency is not None and flows is not None and occs is not None:
        # background smoothing: decode, warp the previous frame's background in, encode (pipe_FRESCO.py:44-47)
        image = pipe.vae.decode(pred_original_sample / pipe.vae.config.scaling_factor).sample
        image = warp_tensor(image, flows, occs, saliency, unet_chunk_size=1)
        pred_original_sample = pipe.vae.config.scaling_factor * pipe.vae.encode(image).latent_dist.sample()

    c_x0 = (alpha_prod_t_prev ** 0.5 * current_beta_t) / beta_prod_t
    c_xt = current_alpha_t ** 0.5 * beta_prod_t_prev / beta_prod_t
    variance = max(beta_prod_t_prev / beta_prod_t * current_beta_t, 1e-20)
    noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                        dtype=model_output.dtype)
    n = sample.numel()
    period = n // sample.shape[0] if repeat_noise else n  # repeat_noise: frame 0's noise for every frame
    out = torch.empty_like(sample)
    x0c = pred_original_sample.to(sample.dtype).contiguous()
    rc = _lib.load().fresco_ddpm_prev(x0c.data_ptr(), sample.contiguous().data_ptr(), noise.data_ptr(),
                                      out.data_ptr(), n, period, c_x0, c_xt, variance ** 0.5, _dt(sample),
                                      ops._stream())
    _lib.check(rc, "fresco_ddpm_prev")
    return (out, pred_original_sample)
