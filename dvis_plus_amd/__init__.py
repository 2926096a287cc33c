"""dvis_plus_amd — MI355X (gfx950) native hot path of DVIS++ behind the reference's op / module surfaces.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/dvis_hip.h), built by ``build.py``
  native.py        ctypes binding of that C ABI (fails loudly; no fallback)
  functions.py     MSDeformAttnFunction + the other op front-ends (tensor checks, pointer plumbing)
  modules.py ...   host-side mirrors of the reference's modules (same ctor args / state_dict keys)
"""
__version__ = "0.1.0"
