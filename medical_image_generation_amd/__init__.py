"""medical_image_generation_amd -- MI355X-native hot path of VKostoulas/Medical_Image_Generation.

Hand-written HIP kernels (gfx950) behind a C ABI (include/medimgen_hip.h, libmedimgen_hip.so),
exposed to PyTorch as torch.autograd.Function ops and as drop-in `DiffusionModelUNet` /
`AutoencoderKL` modules with the reference's constructor signatures and state_dict names.
"""
__version__ = "0.1.0"
