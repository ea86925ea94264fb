"""Forward hook that applies feature optimisation + background smoothing at the input of the UNet's
up-blocks: apply_FRESCO_opt / disable_FRESCO_opt with the reference's signatures
(src/diffusion_hacked.py:491-831).

The reference re-implements `UNet2DConditionModel.forward` (a ~300-line copy of diffusers 0.19.3) to
insert three hacks at lines 757-779 and 811-812.  Here the same behaviour is obtained without touching
the model's forward: a `forward_pre_hook` on every `unet.up_blocks[i]` rewrites that block's
`hidden_states` input, and a thin wrapper around `unet.forward` latches the timestep and appends the
collected decoder features to the tuple output -- so it works with any diffusers version that has
`up_blocks` called with `hidden_states`.

  * `up_samples`: input of up-block i (before optimisation), for i in `layers`   (773-774, 811-812)
  * if timestep in steps and i in layers: optimize_feature(...), then warp_tensor(..., 2) when a
    saliency map is given                                                        (775-779)
`timestep in steps` is evaluated once per forward on the host (one sync instead of the reference's
four tensor membership tests per UNet pass).
"""
import torch

from . import opt as _opt
from . import warp as _warp

_STATE = "_fresco_opt_state"


class _OptState:
    def __init__(self, steps, layers, flows, occs, correlation_matrix, intra_weight, iters,
                 optimize_temporal, saliency):
        self.steps = steps
        self.step_set = None
        if steps is not None and len(steps) > 0:
            self.step_set = set(int(s) for s in (steps.tolist() if torch.is_tensor(steps) else steps))
        self.layers = list(layers)
        self.flows, self.occs = flows, occs
        self.correlation_matrix = correlation_matrix
        self.intra_weight, self.iters = intra_weight, iters
        self.optimize_temporal = optimize_temporal
        self.saliency = saliency
        self.shard = None  # fresco_amd.dist.FrameShard: set on unet._fresco_opt_state for frame-parallel runs
        self.active = False
        self.up_samples = ()
        self.handles = []

    def latch(self, timestep):
        self.up_samples = ()
        self.active = False
        if self.step_set:
            t = int(timestep) if not torch.is_tensor(timestep) or timestep.numel() == 1 else None
            if t is None:
                raise ValueError("fresco_amd hook: one timestep per UNet call expected")
            self.active = t in self.step_set


def _make_block_hook(state, i):
    def hook(module, args, kwargs):
        if i not in state.layers:
            return None
        in_kwargs = "hidden_states" in kwargs
        sample = kwargs["hidden_states"] if in_kwargs else args[0]
        state.up_samples += (sample,)
        if not state.active:
            return None
        extra = {} if state.shard is None else {"shard": state.shard}
        sample = _opt.optimize_feature(sample, state.flows, state.occs, state.correlation_matrix,
                                       state.intra_weight, state.iters,
                                       optimize_temporal=state.optimize_temporal, **extra)
        if state.saliency is not None:
            sample = _warp.warp_tensor(sample, state.flows, state.occs, state.saliency, 2, **extra)
        if in_kwargs:
            kwargs = dict(kwargs)
            kwargs["hidden_states"] = sample
            return args, kwargs
        return (sample,) + tuple(args[1:]), kwargs

    return hook


def _remove(unet):
    old = getattr(unet, _STATE, None)
    if old is not None:
        for h in old.handles:
            h.remove()
        if "forward" in unet.__dict__:
            del unet.__dict__["forward"]
        delattr(unet, _STATE)


def apply_FRESCO_opt(pipe, steps=[], layers=[0, 1, 2, 3], flows=None, occs=None, correlation_matrix=[],
                     intra_weight=1e2, iters=20, optimize_temporal=True, saliency=None):
    """Apply FRESCO-based optimisation to a StableDiffusionPipeline (diffusion_hacked.py:819-825)."""
    unet = pipe.unet
    _remove(unet)
    state = _OptState(steps, layers, flows, occs, correlation_matrix, intra_weight, iters,
                      optimize_temporal, saliency)
    for i, blk in enumerate(unet.up_blocks):
        state.handles.append(blk.register_forward_pre_hook(_make_block_hook(state, i), with_kwargs=True))
    inner = type(unet).forward

    def forward(sample, timestep, *args, **kwargs):
        state.latch(timestep)
        return_dict = kwargs.get("return_dict", True)
        out = inner(unet, sample, timestep, *args, **kwargs)
        ups, state.up_samples = state.up_samples, ()
        if not return_dict:
            return tuple(out) + ups  # (sample,) + up_samples, diffusion_hacked.py:811-812
        return out

    unet.forward = forward
    setattr(unet, _STATE, state)


def disable_FRESCO_opt(pipe):
    """diffusion_hacked.py:827-831: keeps returning the decoder features, never optimises."""
    apply_FRESCO_opt(pipe)


def patch_reference(dh=None, pf=None, fu=None):
    """Zero-diff drop-in for an imported reference tree: rebinds the names the
    reference looks up at call time to this package's implementations.

        import src.diffusion_hacked as dh, src.pipe_FRESCO as pf, src.flow_utils as fu
        fresco_amd.patch_reference(dh, pf, fu)
    """
    from . import processor

    if dh is not None:
        dh.FRESCOAttnProcessor2_0 = processor.FRESCOAttnProcessor2_0
        dh.AttentionControl = processor.AttentionControl
        dh.optimize_feature = _opt.optimize_feature
        dh.warp_tensor = _warp.warp_tensor
        dh.flow_warp = _warp.flow_warp
        dh.adaptive_instance_normalization = _warp.adaptive_instance_normalization
    if pf is not None:
        # (`from . import step` would fetch the FUNCTION the package re-exports under the submodule's name)
        from .step import step as _step_fn

        pf.warp_tensor = _warp.warp_tensor
        pf.step = _step_fn  # inference() looks `step` up in its module globals (pipe_FRESCO.py:222-228)
    if fu is not None:
        from . import mapping

        fu.warp_tensor = _warp.warp_tensor
        fu.flow_warp = _warp.flow_warp
        fu.get_mapping_ind = mapping.get_mapping_ind
        fu.get_single_mapping_ind = mapping.get_single_mapping_ind
    if dh is not None:
        from . import mapping

        from . import paras

        dh.get_mapping_ind = mapping.get_mapping_ind  # looked up by get_flow_and_interframe_paras (:943)
        dh.forward_backward_consistency_check = paras.forward_backward_consistency_check  # (:917)
        # run_fresco.py:20 binds these two by `from ... import` before anyone can patch, so a caller that
        # wants them replaced imports them from fresco_amd; rebinding covers webUI-style
        # late lookups through the module
        dh.get_flow_and_interframe_paras = paras.get_flow_and_interframe_paras
        dh.get_intraframe_paras = paras.get_intraframe_paras
