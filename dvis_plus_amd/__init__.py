"""dvis_plus_amd — MI355X (gfx950) native hot path of DVIS++ behind the reference's op / module surfaces.

Layout:
  csrc/                    hand-written HIP kernels + the C ABI (include/dvis_hip.h), built by ``build.py``
  native.py                ctypes binding of that C ABI (fails loudly; no fallback)
  functions.py             MSDeformAttnFunction + the other op front-ends (tensor checks, pointer plumbing, DVIS_STRICT)
  pixel_decoder.py         MSDeformAttn, deformable encoder, MSDeformAttnPixelDecoder       (mask2former/.../pixel_decoder)
  transformer_decoder.py   masked-attention decoders (_dvisPlus, _dvis, _minvis, image)     (…/transformer_decoder)
  tracker.py, refiner.py   ReferringTracker_noiser, TemporalRefiner                         (dvis_Plus/tracker.py, refiner.py)
  meta_architecture.py     MaskFormerHead, MinVIS, DVIS_Plus_online / _offline, MaskFormer  (dvis_Plus/meta_architecture.py)
  postprocess.py           inference_video_{vis,vps,vss} on the device
  backbone.py, vit_adapter.py   R50 front-end (fused epilogues), DINOv2 ViT + ViT-Adapter
  d2.py, registry.py, config.py detectron2's construction protocol (configurable / from_config / builders / install()),
                                registries under the reference's names, yacs-style config loader
  clip_shard.py, graphs.py frame sharding over the GPUs of a node (RCCL), hipGraph capture of the tracker / refiner
"""
__version__ = "0.2.0"
