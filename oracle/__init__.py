"""CPU oracle for the ppdiffusers Stable-Diffusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``paddlemix_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported CPU baseline.

What it is: a torch-CPU fp32 (optionally fp64) restatement of the reference's
UNet2DConditionModel / SD3 MMDiT forward and of the three schedulers on the
path.  PaddlePaddle is not installable in this environment (``import paddle``
fails, no network; pinned upstream: paddlepaddle-gpu==3.0.0b1,
/root/reference/build_paddle_env.sh:28-42), so the reference itself cannot be
executed; every function cites the reference file:line it follows.

Parity pin status:
  * pinned by the reference's own RNG-free known-answer tests:
    ``get_timestep_embedding`` (tests/models/test_layers_utils.py:90-115),
    DDIM / Euler full-loop sums (tests/schedulers/test_scheduler_ddim.py:121-190,
    test_scheduler_euler.py:84-163), activation fixed points
    (tests/models/test_activations.py:24-62);
  * whole-UNet / whole-MMDiT level with real weights: PARITY UNPINNED in this
    container (the reference's expected slices need Paddle RNG, real weights or
    HF-hosted fixtures).  Whole-model parity is defined as device-vs-this-oracle
    on identical synthetic weights and inputs.
"""
