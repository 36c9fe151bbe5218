"""vit_ae_plus_plus_amd — MI355X (gfx950) native training path for the 3-D ViT-AE++ masked
autoencoder of chinmay5/vit_ae_plus_plus.

Layout
  csrc/ + libvitae_hip.so   hand-written HIP kernels behind the C ABI of include/vitae_hip.h
  _abi.py                   ctypes binding generated from that header (no fallback: raises)
  engine.py                 arenas, workspace and the forward/backward/optimiser launch lists
  model/, utils/            mirror of the reference's Python surface
  ddp.py                    bucketed RCCL gradient all-reduce for one-process-per-GPU data parallel
  dropin.py                 registers this package's model/ and utils/ as top-level ``model`` /
                            ``utils`` so the reference's scripts import them unchanged
"""
__version__ = '0.1.0'
