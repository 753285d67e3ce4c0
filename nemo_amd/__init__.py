"""nemo_amd - MI355X-native implementation of nemo's map-filtering + detection hot path.

Drop-in surface (reference module in brackets): :mod:`nemo_amd.filters` [nemo.filters], :mod:`nemo_amd.photometry`
[nemo.photometry], :mod:`nemo_amd.pipelines` [nemo.pipelines], :mod:`nemo_amd.startUp` [nemo.startUp].
All map-sized arithmetic is done by libnemofilt (hand-written gfx950 HIP kernels behind the C ABI in
include/nemofilt.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
