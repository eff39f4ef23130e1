"""dynibar_amd: MI355X-native (gfx950) implementation of DynIBaR's per-ray renderer.

Drop-in for the reference's ibrnet.sample_ray / projection / render_ray / render_image
import surface.  All per-ray arithmetic runs in hand-written HIP kernels
behind the C-ABI declared in include/dynibar_hip.h; there is no CPU or eager-PyTorch
fallback: importing the compute entry points without the built library raises.
"""
__version__ = '0.1.0'
