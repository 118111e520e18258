"""covomix_amd - MI355X-native (gfx950) inference path for CoVoMix's mel-generation hot loop.

Scope: the VoMix/VoSingle flow-matching vector field + fixed-grid ODE
sampler and the HiFi-GAN generator, as hand-written HIP kernels behind a C ABI
(include/covomix_hip.h), with a Python host layer that mirrors the reference's
`CoVoMixModel.synthesis_sample` / `Generator` interface.
"""
__version__ = "0.1.0"
