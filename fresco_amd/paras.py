"""Per-batch parameter producers of the FRESCO path, with the reference's signatures
(src/diffusion_hacked.py PART III, 842-957): everything between the flow network / the one-step UNet pass
and the attention / optimisation kernels runs on the GPU through the C ABI.

  forward_backward_consistency_check   gmflow/geometry.py:75-96          fresco_flow_occlusion
  get_flow_and_interframe_paras        src/diffusion_hacked.py:904-957   + fresco_resize_bilinear, fresco_mapping_ind
  get_intraframe_paras                 src/diffusion_hacked.py:842-901   fresco_gram_target

The flow network itself (GMFlow) and the UNet / VAE / scheduler are the caller's torch modules (SURVEY.md 8:
out of scope); they are invoked exactly as the reference invokes them.
"""
import gc

import torch

from . import ops
from .mapping import cross_frame_masks, get_mapping_ind


def forward_backward_consistency_check(fwd_flow, bwd_flow, alpha=0.01, beta=0.5):
    """geometry.py:75-96: (fwd_occ, bwd_occ), float {0,1} (B,H,W)."""
    assert fwd_flow.dim() == 4 and bwd_flow.dim() == 4
    assert fwd_flow.size(1) == 2 and bwd_flow.size(1) == 2
    return ops.flow_occlusion(fwd_flow, bwd_flow, None, alpha, beta)


def interframe_paras_from_flows(images, fwd_flows, bwd_flows):
    """Lines 917-953 of get_flow_and_interframe_paras for given flows.  images (N,3,H,W) float 0..255 on the
    GPU.  Returns ([fwd_flows, bwd_flows], [fwd_occs, bwd_occs], attn_mask, interattn_paras)."""
    fwd_occs, bwd_occs = ops.flow_occlusion(fwd_flows, bwd_flows, images, 0.01, 0.5, 255 * 0.25)
    imgs_torch = images / 255.0 * 2.0 - 1.0  # numpy2tensor, utils.py:9
    attn_mask = cross_frame_masks(bwd_occs, (8.0, 16.0, 32.0))
    interattn_paras = {"fwd_mappings": [], "bwd_mappings": [], "interattn_masks": []}
    for scale in (8.0, 16.0):
        f, b, m = get_mapping_ind(bwd_flows, bwd_occs, imgs_torch, scale=scale)
        interattn_paras["fwd_mappings"].append(f)
        interattn_paras["bwd_mappings"].append(b)
        interattn_paras["interattn_masks"].append(m)
    return [fwd_flows, bwd_flows], [fwd_occs, bwd_occs], attn_mask, interattn_paras


@torch.no_grad()
def get_flow_and_interframe_paras(flow_model, imgs, visualize_pipeline=False):
    """diffusion_hacked.py:904-957.  imgs: list of HxWx3 uint8 numpy frames.  `flow_model` is called as the
    reference calls GMFlow (frame i against frame i+1 mod N, bidirectional).  `visualize_pipeline` is accepted
    for signature compatibility; the matplotlib previews (927-932) belong to the UI and are not reproduced."""
    dev = torch.device("cuda", torch.cuda.current_device())
    images = torch.stack([torch.from_numpy(img).permute(2, 0, 1).float() for img in imgs], dim=0).to(dev)
    reshuffle_list = list(range(1, len(images))) + [0]
    results_dict = flow_model(images, images[reshuffle_list], attn_splits_list=[2], corr_radius_list=[-1],
                              prop_radius_list=[-1], pred_bidir_flow=True)
    flow_pr = results_dict["flow_preds"][-1]   # (2N,2,H,W)
    fwd_flows, bwd_flows = flow_pr.chunk(2)
    return interframe_paras_from_flows(images, fwd_flows.float().contiguous(), bwd_flows.float().contiguous())


def correlation_matrices(features):
    """Lines 888-895: per decoder feature map (B,C,h,w) the Gram matrix of the L2-normalised pixel vectors,
    fp32 (B,hw,hw).  Computed in fp32 from the stored features (the reference normalises and multiplies in the
    UNet dtype and casts the result: for fp16 features ours differs by that fp16 rounding, <= 1e-3)."""
    return [ops.gram_target(t) for t in features]


@torch.no_grad()
def get_intraframe_paras(pipe, imgs, frescoProc, prompt_embeds, do_classifier_free_guidance=True, seed=0):
    """diffusion_hacked.py:842-901: one denoising pass at the last timestep with the controller storing the
    decoder hidden states (-> spatial-guided attention), Gram targets of the decoder features (-> spatial
    consistency loss).  pipe.unet.forward must return (sample, *up_samples), i.e. apply_FRESCO_opt has run."""
    from .hook import disable_FRESCO_opt

    noise_scheduler = pipe.scheduler
    timestep = noise_scheduler.timesteps[-1]
    device = pipe._execution_device
    generator = torch.Generator(device=device).manual_seed(seed)
    B, C, H, W = imgs.shape

    frescoProc.controller.disable_controller()
    disable_FRESCO_opt(pipe)
    frescoProc.controller.clear_store()
    frescoProc.controller.enable_store()

    latents = pipe.prepare_latents(B, pipe.unet.config.in_channels, H, W, prompt_embeds.dtype, device, generator,
                                   latents=None)
    latent_x0 = pipe.vae.config.scaling_factor * pipe.vae.encode(imgs.to(pipe.unet.dtype)).latent_dist.sample()
    latents = noise_scheduler.add_noise(latent_x0, latents, timestep).detach()
    latent_model_input = torch.cat([latents] * 2) if do_classifier_free_guidance else latents
    model_output = pipe.unet(latent_model_input, timestep, encoder_hidden_states=prompt_embeds,
                             cross_attention_kwargs=None, return_dict=False)
    frescoProc.controller.disable_store()
    correlation_matrix = correlation_matrices(model_output[1:])
    del model_output
    gc.collect()
    return correlation_matrix
