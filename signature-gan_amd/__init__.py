"""MI355X-native engine for the signature GAN's G+D train step and generation path.

Hand-written HIP kernels (gfx950) behind a C ABI (``include/siggan.h``), called through ctypes with
PyTorch-ROCm tensors used for device memory and streams only.  ``engine.Engine`` is the host-side
context wrapper; ``generator_vanilla_gan`` / ``discriminator_vanilla_gan`` / ``vanilla_gan_model`` /
``train_vanilla_gan_signatures`` mirror the reference's modules for this path.
"""
__version__ = "0.1.0"
