"""UNet configs shared by the tests (shapes follow the reference's own test / model configs)."""

# kernel-compatible shrink of the reference's tiny test config (ppdiffusers/tests/models/test_models_unet_2d_condition.py:181-194:
# block_out (32,64), cross 32, head_dim 8) -- channels doubled so every GEMM K is a multiple of 8 groups of 32.
TINY = dict(block_out_channels=(64, 128), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), cross_attention_dim=64, attention_head_dim=8,
            layers_per_block=2, norm_num_groups=32, sample_size=32)

# SDXL structure in miniature (mirrors ppdiffusers/tests/pipelines/stable_diffusion_xl/test_stable_diffusion_xl.py:66-83):
# DownBlock + 2 CrossAttnDown, transformer_layers (1,2,2), text_time add-embedding, linear projections, head_dim 32/64
MINI_XL = dict(block_out_channels=(64, 128, 256), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
               up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
               transformer_layers_per_block=(1, 2, 2), attention_head_dim=(2, 4, 4), cross_attention_dim=128,
               use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=32,
               projection_class_embeddings_input_dim=32 * 6 + 64, layers_per_block=2, sample_size=32)

# SD-1.5 (ppdiffusers/examples/stable_diffusion/sd/unet_config.json)
SD15 = dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768, attention_head_dim=8,
            layers_per_block=2, sample_size=64)

# SDXL base UNet (public model config; structure per SURVEY.md 8a "X")
SDXL = dict(block_out_channels=(320, 640, 1280), down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
            transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
            use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=256,
            projection_class_embeddings_input_dim=2816, layers_per_block=2, sample_size=128)

# SD3 MMDiT in miniature (the reference's own tiny test config, ppdiffusers/tests/models/test_models_transformer_sd3.py:60-73,
# widened so that every GEMM K is a multiple of 8 and head_dim is 32)
MINI_SD3 = dict(sample_size=32, patch_size=2, in_channels=4, num_layers=3, attention_head_dim=32, num_attention_heads=4,
                caption_projection_dim=128, joint_attention_dim=64, pooled_projection_dim=64, out_channels=4,
                pos_embed_max_size=96)
# SD3-medium (public config: 24 layers, 24 heads x 64, joint dim 4096, pooled 2048, 16 latent channels, pos-embed 192)
SD3_MEDIUM = dict(sample_size=128, patch_size=2, in_channels=16, num_layers=24, attention_head_dim=64,
                  num_attention_heads=24, caption_projection_dim=1536, joint_attention_dim=4096,
                  pooled_projection_dim=2048, out_channels=16, pos_embed_max_size=192)
# AutoencoderKL decoder in miniature (same structure as the SD VAE: 3 levels here, 32 groups -> channels multiples of 32)
MINI_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64, 64), layers_per_block=1,
                norm_num_groups=32, scaling_factor=0.18215, use_post_quant_conv=True)
# SD-1.5 / SDXL VAE (public config: 128-256-512-512, 2 layers per block; SDXL scaling_factor 0.13025)
SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
              layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, use_post_quant_conv=True)
# CLIP text encoder in miniature (reference tiny config shape: ppdiffusers/tests/pipelines/stable_diffusion/test_stable_diffusion.py:137-150,
# widened so head_dim = 32) and the two SD encoders (public configs)
MINI_CLIP = dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                 max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=32, eos_token_id=2)
CLIP_L = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
              max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768, eos_token_id=2)
CLIP_BIGG = dict(vocab_size=49408, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                 max_position_embeddings=77, hidden_act="gelu", projection_dim=1280, eos_token_id=2)
# T5 v1.1 encoder in miniature and the XXL geometry SD3 uses (public config: d_model 4096, 64 heads x 64, d_ff 10240, 24 layers)
MINI_T5 = dict(vocab_size=500, d_model=64, d_kv=16, d_ff=128, num_layers=3, num_heads=4)
T5_XXL = dict(vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64)

# configuration variants the reference's own model tests exercise (ppdiffusers/tests/models/test_models_unet_2d_condition.py:
# :250 attention_head_dim tuple, :268 use_linear_projection, :286 cross_attention_dim tuple; :868 the 9-channel inpainting
# UNet; :924 the SD-2 layout = linear projections + per-level head counts) on the tiny geometry
UNET_VARIANTS = {
    "inpaint-9ch": dict(TINY, in_channels=9),
    "cross-dim-tuple": dict(TINY, cross_attention_dim=(64, 64)),
    "head-dim-tuple": dict(TINY, attention_head_dim=(8, 16)),
    "layers-per-block-tuple": dict(TINY, layers_per_block=(1, 2)),
    "sd2-layout": dict(TINY, use_linear_projection=True, attention_head_dim=(4, 8)),
    "upcast-attention": dict(TINY, upcast_attention=True),
    "center-input": dict(TINY, center_input_sample=True),
    "sin-first-shifted": dict(TINY, flip_sin_to_cos=False, freq_shift=1),
}

# class-conditional DiT in miniature (the reference's tiny test config, ppdiffusers/tests/pipelines/dit/test_dit.py:46-60,
# widened to head_dim 32 / K % 8) and DiT-XL/2 (public config: 28 layers, 16 heads x 72, 1000 classes, learned sigma)
MINI_DIT = dict(sample_size=16, num_layers=3, patch_size=2, attention_head_dim=32, num_attention_heads=4, in_channels=4,
                out_channels=8, num_embeds_ada_norm=10)
DIT_XL2 = dict(sample_size=32, num_layers=28, patch_size=2, attention_head_dim=72, num_attention_heads=16, in_channels=4,
               out_channels=8, num_embeds_ada_norm=1000)

# CLIP vision towers (CLIPVisionModelWithProjection): a miniature with the awkward patch width of the real ones (3 * 14 * 14 = 588
# input columns, not a multiple of 8) and OpenCLIP ViT-H/14, the image encoder the IP-Adapter checkpoints ship with
MINI_CLIP_VISION = dict(hidden_size=64, intermediate_size=128, projection_dim=48, num_hidden_layers=2, num_attention_heads=2,
                        num_channels=3, image_size=56, patch_size=14, hidden_act="quick_gelu")
CLIP_VIT_H14 = dict(hidden_size=1280, intermediate_size=5120, projection_dim=1024, num_hidden_layers=32, num_attention_heads=16,
                    num_channels=3, image_size=224, patch_size=14, hidden_act="gelu")
