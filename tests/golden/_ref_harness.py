"""Import the UNMODIFIED reference modules from /root/reference on CPU (test infrastructure only).

Used by ``make_golden.py`` (fixture generation, build container only) and by the optional
``tests/test_oracle_vs_reference.py`` cross-check.  /root/reference does not exist on the GPU
box, so nothing that runs there may import this module.

Recipe follows SURVEY.md Appendix B: the reference hard-imports cv2 and diffusers (absent here),
so empty stub modules are registered first; only the names used at import time are provided
(src/diffusion_hacked.py:7-8, src/utils.py:4, src/flow_utils.py:3).
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "src"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns (diffusion_hacked module, flow_utils module, gmflow.geometry module, utils module)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except Exception:
            _stub("matplotlib")
            _stub("matplotlib.pyplot")
    if "diffusers" not in sys.modules:
        class UNet2DConditionOutput:  # only constructed at diffusion_hacked.py:814
            def __init__(self, sample=None):
                self.sample = sample

        class AttnProcessor2_0:  # only constructed at diffusion_hacked.py:395
            pass

        _stub("diffusers")
        _stub("diffusers.models")
        _stub("diffusers.models.unet_2d_condition", UNet2DConditionOutput=UNet2DConditionOutput)
        _stub("diffusers.models.attention_processor", AttnProcessor2_0=AttnProcessor2_0)
    cwd = os.getcwd()
    os.chdir(REF_ROOT)  # the reference appends relative paths to sys.path
    try:
        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
        import src.diffusion_hacked as dh
        import src.flow_utils as fu
        import src.utils as ut
        from gmflow import geometry as geo
    finally:
        os.chdir(cwd)
    return dh, fu, geo, ut
