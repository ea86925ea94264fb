"""fresco_amd -- MI355X-native (gfx950) implementation of FRESCO's hot path: flow-guided spatial /
cross-frame / temporal attention, bilinear feature warp + fuse, and the feature-optimisation loop,
behind the reference's own plugin surface (diffusers AttnProcessor + UNet forward hook).

Everything computes in libfresco_hip.so (hand-written HIP, C ABI in include/fresco_hip.h); importing
the package does not need a GPU, calling any operator does, and there is no CPU fallback.
"""
from ._lib import FrescoHipError, LIB_PATH
from .control import AttentionControl
from .processor import FRESCOAttnProcessor2_0, apply_FRESCO_attn
from .opt import optimize_feature
from .warp import Dilate, adaptive_instance_normalization, calc_mean_std, flow_warp, warp_tensor
from .hook import apply_FRESCO_opt, disable_FRESCO_opt, patch_reference
from .mapping import cross_frame_masks, get_mapping_ind, get_single_mapping_ind
from .step import predict_x0, step
from .paras import (correlation_matrices, forward_backward_consistency_check, get_flow_and_interframe_paras,
                    get_intraframe_paras, interframe_paras_from_flows)

__all__ = [
    "AttentionControl", "FRESCOAttnProcessor2_0", "apply_FRESCO_attn", "optimize_feature",
    "warp_tensor", "flow_warp", "adaptive_instance_normalization", "calc_mean_std", "Dilate", "apply_FRESCO_opt",
    "disable_FRESCO_opt", "patch_reference", "get_mapping_ind", "get_single_mapping_ind", "cross_frame_masks",
    "step", "predict_x0", "get_flow_and_interframe_paras", "get_intraframe_paras", "interframe_paras_from_flows",
    "forward_backward_consistency_check", "correlation_matrices",
    "FrescoHipError", "LIB_PATH",
]
