"""Named parity cases shared by the golden generator and the tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `C1` is BASELINE.json configs[0]
verbatim; the others are reduced-size instances of C2/C4 (pixel DDPM),
C3a (AutoencoderKL) and C3b (latent UNet, one attention head of full width) sized so
the reference finishes in seconds on CPU.
"""
from __future__ import annotations

SEED = 42


def _iso(n, levels, first=1, rest=2):
    return [[first] * n] + [[rest] * n] * (levels - 1)


UNET_CASES = {
    # BASELINE configs[0]: 2D DDPM 64x64, tiny U-Net (2 levels, 32 base ch)
    "unet_c1": dict(
        kwargs=dict(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=2, num_channels=(32, 64),
                    attention_levels=(False, True), num_head_channels=(0, 64), norm_num_groups=32,
                    strides=_iso(2, 2), kernel_sizes=[[3, 3]] * 2, paddings=[[1, 1]] * 2),
        shape=(2, 1, 64, 64), timesteps=(17, 903)),
    # C2/C4-shaped (pixel-space 3D DDPM), reduced: 3 levels, attention at the coarsest only
    "unet3d": dict(
        kwargs=dict(spatial_dims=3, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(32, 64, 64),
                    attention_levels=(False, False, True), num_head_channels=(0, 0, 32), norm_num_groups=32,
                    strides=_iso(3, 3), kernel_sizes=[[3] * 3] * 3, paddings=[[1] * 3] * 3),
        shape=(2, 1, 16, 16, 16), timesteps=(0, 999)),
    # C3b-shaped (latent UNet): 8 latent channels, attention on two levels with ONE head of full width
    "unet_ldm": dict(
        kwargs=dict(spatial_dims=3, in_channels=8, out_channels=8, num_res_blocks=1, num_channels=(32, 64, 96),
                    attention_levels=(False, True, True), num_head_channels=(0, 64, 96), norm_num_groups=32,
                    strides=_iso(3, 3), kernel_sizes=[[3] * 3] * 3, paddings=[[1] * 3] * 3),
        shape=(2, 8, 8, 8, 8), timesteps=(250, 750)),
    # class-conditional variant (UNet:1837-1839, 1975-1980)
    "unet2d_class": dict(
        kwargs=dict(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(16, 32),
                    attention_levels=(False, True), num_head_channels=(0, 16), norm_num_groups=8, num_class_embeds=4,
                    strides=_iso(2, 2), kernel_sizes=[[3, 3]] * 2, paddings=[[1, 1]] * 2),
        shape=(2, 1, 16, 16), timesteps=(3, 500), class_labels=(1, 3)),
    # resblock_updown=True (avg-pool / nearest resnet resamplers, UNet:640-644, 679-687)
    "unet2d_updown": dict(
        kwargs=dict(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(16, 32),
                    attention_levels=(False, False), num_head_channels=(0, 16), norm_num_groups=8, resblock_updown=True,
                    strides=[[1, 1], [2, 2]], kernel_sizes=[[3, 3], [2, 2]], paddings=[[1, 1], [0, 0]]),
        shape=(2, 1, 16, 16), timesteps=(10, 20)),
    # with_conditioning=True: SpatialTransformer (self-attention + cross-attention on a context + GEGLU feed-forward) in place of
    # AttentionBlock at the attention levels (UNet:72-342); 2 transformer layers, context of 5 tokens x 16 channels
    "unet2d_xattn": dict(
        kwargs=dict(spatial_dims=2, in_channels=1, out_channels=1, num_res_blocks=1, num_channels=(32, 64),
                    attention_levels=(False, True), num_head_channels=(0, 32), norm_num_groups=8, with_conditioning=True,
                    transformer_num_layers=2, cross_attention_dim=16,
                    strides=_iso(2, 2), kernel_sizes=[[3, 3]] * 2, paddings=[[1, 1]] * 2),
        shape=(2, 1, 16, 16), timesteps=(5, 600), context=(2, 5, 16)),
    # ---- the BASELINE configs on their EXACT model kwargs, at sizes the reference finishes in seconds on CPU ----
    # C2 / C4: pixel-space 3D DDPM net, (32,64,128,256), 2 res-blocks, attention at the coarsest level with heads of 64
    "unet_c4": dict(
        kwargs=dict(spatial_dims=3, in_channels=1, out_channels=1, num_res_blocks=2, num_channels=(32, 64, 128, 256),
                    attention_levels=(False, False, False, True), num_head_channels=(0, 0, 0, 64), norm_num_groups=32,
                    strides=_iso(3, 4), kernel_sizes=[[3] * 3] * 4, paddings=[[1] * 3] * 4),
        shape=(2, 1, 32, 32, 32), timesteps=(100, 800)),
    # the same net on non-power-of-two, anisotropic extents (every level ragged against the 4x8x8 conv tiles: 24x40x48 -> 3x5x6)
    "unet_c4_np2": dict(
        kwargs=dict(spatial_dims=3, in_channels=1, out_channels=1, num_res_blocks=2, num_channels=(32, 64, 128, 256),
                    attention_levels=(False, False, False, True), num_head_channels=(0, 0, 0, 64), norm_num_groups=32,
                    strides=_iso(3, 4), kernel_sizes=[[3] * 3] * 4, paddings=[[1] * 3] * 4),
        shape=(1, 1, 24, 40, 48), timesteps=(611,)),
    # C3b: the ONLY latent UNet the reference's planner emits (CFG:876-902): [256,512,768], ONE head of 512 / 768
    "unet_c3b": dict(
        kwargs=dict(spatial_dims=3, in_channels=8, out_channels=8, num_res_blocks=2, num_channels=(256, 512, 768),
                    attention_levels=(False, True, True), num_head_channels=(0, 512, 768), norm_num_groups=32,
                    strides=_iso(3, 3), kernel_sizes=[[3] * 3] * 3, paddings=[[1] * 3] * 3),
        shape=(1, 8, 8, 8, 8), timesteps=(333,)),
    # C5: C3b net with one label channel concatenated to the latents (in_channels = 8 + 1), non-power-of-two latent extent
    "unet_c5": dict(
        kwargs=dict(spatial_dims=3, in_channels=9, out_channels=8, num_res_blocks=2, num_channels=(256, 512, 768),
                    attention_levels=(False, True, True), num_head_channels=(0, 512, 768), norm_num_groups=32,
                    strides=_iso(3, 3), kernel_sizes=[[3] * 3] * 3, paddings=[[1] * 3] * 3),
        shape=(1, 9, 12, 12, 12), timesteps=(42,)),
}

_C3A_DOWN = [[[1] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3]]

AEKL_CASES = {
    # C3a: exactly what CFG:821-862 emits for a 128^3 single-channel dataset, run on a 32^3 patch
    "aekl_c3a": dict(
        kwargs=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, num_res_blocks=2,
                    num_channels=[32, 64, 128], attention_levels=[False, False, False], norm_num_groups=16,
                    with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False,
                    downsample_parameters=_C3A_DOWN, upsample_parameters=list(reversed(_C3A_DOWN))[:-1]),
        shape=(1, 1, 32, 32, 32)),
    # attention everywhere it can appear (level attention + both non-local blocks), anisotropic last level
    "aekl_attn": dict(
        kwargs=dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, num_res_blocks=1,
                    num_channels=[16, 32], attention_levels=[False, True], norm_num_groups=8,
                    with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True,
                    downsample_parameters=[[[1] * 3, [3] * 3, [1] * 3], [[2, 2, 1], [3, 3, 1], [1, 1, 0]]],
                    upsample_parameters=[[[2, 2, 1], [3, 3, 1], [1, 1, 0]]]),
        shape=(2, 1, 16, 16, 4)),
}

# Optimiser goldens: 3 steps, clip 1.0 (T-LDM:121,175-177 AdamW; T-AE:470 / T-DDPM:383 Adam)
STEP_LR = 1e-3
STEP_COUNT = 3
STEP_CASES = {"unet3d": "AdamW", "unet_c1": "Adam"}
KL_WEIGHT = 1e-7  # CFG:995-1026 (3-D)
