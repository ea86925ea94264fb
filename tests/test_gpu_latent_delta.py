"""Row n1 of VERDICT r04 / the second half of BASELINE.json's metric: max latent delta vs the reference path over the
denoising loop of src/pipe_FRESCO.py:166-228 -- six steps of the reference's schedule (spatial + cross-frame + temporal,
cross-frame + temporal, cross-frame only) on a stand-in SD-1.5 UNet + ControlNet (tools/standin_unet.py: diffusers'
module tree and shapes, random fp16 weights), 8 frames x 512^2, from identical latents / weights / FRESCO parameters /
per-step noise: fresco_amd's processor + fresco_amd.step against the reference's own op sequence
(oracle/torch_path.processor_call + step() restated in torch ops) in the same six layers.  Feature optimisation off.

What the bar can be.  The north star's 1e-3 holds per CALL (tests/test_gpu_fullsize.py: every element of every layer call
within 1e-3 of the fp32 oracle; bench.py `torch_gpu_baseline.max_abs_delta` 6e-5 against the reference's op sequence).
Over a LOOP it is not a property an fp16 path can have on this network: the reference's op sequence against ITSELF with
the six layers evaluated in fp32 (its own fp16 rounding noise: ~6e-5 per layer call) already sits 1.3e-2 apart after ONE
step and 5-6e-2 after six (measured, MI355X) -- classifier-free guidance multiplies an eps difference by up to 14
(eu + 7.5 (et - eu)), the step at t = 951 by 0.4, and the random-weight stand-in decoder by ~50.  So the test states the
fact that IS attainable and that a drop-in needs: after every step our latents are no further from the reference's than the
reference's own fp16 noise (x 1.5, + 1e-3), for fp16 latents (the pipeline's dtype) and with the scheduler arithmetic in
fp32 on both sides; and ours vs the fp32-layer reference is no worse than the fp16 reference vs it."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_latent_delta_over_six_denoising_steps():
    import bench_full_step as B
    h = B.Harness(8, 512, "cuda")
    r = B.measure_latent_delta(h)
    f32, f16 = r["fp32_latents"], r["fp16_latents"]
    print("latent delta, fp32 scheduler arithmetic: per step %s | reference's own fp16 noise %s"
          % (f32["max_abs_delta_per_step"], f32["reference_own_fp16_noise_per_step"]))
    print("latent delta, fp16 latents: per step %s | reference's own fp16 noise %s | |latent| max %.2f, %.3f %% of the "
          "elements differ after the last step" % (f16["max_abs_delta_per_step"], f16["reference_own_fp16_noise_per_step"],
                                                  f16["latent_abs_max"], 100 * f16["differing_elements_last_step"]))
    for tag, r_ in (("fp32", f32), ("fp16", f16)):
        for step, (d, n, d32) in enumerate(zip(r_["max_abs_delta_per_step"], r_["reference_own_fp16_noise_per_step"],
                                               r_["ours_vs_reference_with_fp32_layers_per_step"])):
            assert d <= 1.5 * n + 1e-3, (tag, step, d, n)
            assert d32 <= 1.5 * n + 1e-3, (tag, step, d32, n)
    assert f32["mean_abs_delta_last_step"] < 0.25 * f32["max_abs_delta"]
