"""MI355X-native DDPM sampling hot path for MDM (GuyTevet/motion-diffusion-model).

Host side mirrors the reference's three seams -- `MDM.forward(x, timesteps, y)`,
`ClassifierFreeSampleModel`, `SpacedDiffusion.p_sample_loop / ddim_sample_loop` -- on top of the C ABI
of `csrc/libmdm_hip.so` (include/mdm_hip.h).  Import as `mdm_amd` (see /mdm_amd.py).
"""
__version__ = "0.1.0"
