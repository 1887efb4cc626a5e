"""icsg3d_amd -- MI355X-native engine for the ICSG3D voxel hot path (U-Net + Cond-DFC-VAE).

Layout mirrors the reference packages so `from icsg3d_amd.unet.unet import AtomUnet` and
`from icsg3d_amd.vae.lattice_vae import LatticeDFCVAE` replace `from unet.unet import AtomUnet` /
`from vae.lattice_vae import LatticeDFCVAE`.  All compute is in
libicsg3d_hip.so (icsg3d_amd/csrc, C ABI in include/icsg3d.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
