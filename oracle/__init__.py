"""CPU oracle for the ppdiffusers Stable-Diffusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``paddlemix_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported CPU baseline.

What it is: a torch-CPU fp32 (optionally fp64) restatement of the reference's
UNet2DConditionModel / ControlNet / DiT / SD3 MMDiT / AutoencoderKL / CLIP / T5
forwards and of the six schedulers on the path; every function cites the
reference file:line it follows.  PaddlePaddle is not installable in this
environment (``import paddle`` fails, no network; pinned upstream:
paddlepaddle-gpu==3.0.0b1, /root/reference/build_paddle_env.sh:28-42), so the
reference cannot run on Paddle here -- its Python runs over
``oracle/paddle_shim.py`` instead (next paragraph).

Parity pin status:
  * pinned by the reference's own RNG-free known-answer tests:
    ``get_timestep_embedding`` (tests/models/test_layers_utils.py:90-115),
    DDIM / Euler / PNDM / DPM-Solver full-loop sums (tests/schedulers/*), activation
    fixed points (tests/models/test_activations.py:24-62)  -- tests/test_oracle_pins.py;
  * pinned against THE REFERENCE'S OWN CODE (round 3): the reference's model and scheduler
    files -- models/{unet_2d_condition, unet_2d_blocks, resnet, transformer_2d, attention,
    attention_processor, embeddings, normalization, lora, controlnet, transformer_sd3,
    autoencoder_kl, vae}.py, schedulers/scheduling_{ddim, euler_discrete,
    flow_match_euler_discrete, pndm, dpmsolver_multistep, lcm}.py,
    transformers/{clip, t5}/modeling.py -- are loaded unmodified from /root/reference and
    executed on CPU over ``oracle/paddle_shim.py`` (a torch-fp32 stand-in for the ``paddle``
    package) by ``oracle/reference_runner.py``, on this oracle's parameters and inputs.
    Every restated forward agrees with the reference's to fp32 rounding (73 cases, worst
    relative difference 4.7e-6 -- a 5-step SDXL pipeline loop; 1.3e-6 for single forwards --
    most bit-identical), eleven pipeline __call__ loops and the SDXL / SD3 encode_prompt (pipelines/stable_diffusion*/pipeline_*.py,
    pipelines/controlnet/pipeline_controlnet.py: text2img, img2img, inpaint, ControlNet, LCM)
    included; the reference's outputs are committed
    under tests/golden/reference_modules/ (scripts/make_reference_golden.py) and
    tests/test_reference_modules.py holds the oracle to them everywhere and to a live run
    where /root/reference exists.  This pins the restatement's STRUCTURE -- parameter names
    and shapes, wiring, axes, scales, epsilons, the order of scheduler updates;
    At the real architectures (SD-1.5, SDXL, SD3-medium parameter sets) the stored per-step
    predictions of tests/golden/parity/*.npz are reproduced bit for bit by the reference's
    model classes (scripts/check_parity_fixtures_against_reference.py);
  * what remains unpinned: Paddle's own kernels (the array library under the reference's
    code is torch's here) and real checkpoints (none exist in this environment).
"""
