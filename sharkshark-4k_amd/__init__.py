"""MI355X-native drop-in for sharkshark-4k's ``src/upscale`` per-frame super-resolution path.

Python host code (this package) mirrors the reference's service interface
(``BaseService`` / ``BaseUpscalerService`` / ``FsrcnnUpscalerService``) and calls a C-ABI HIP
library (``csrc/`` -> ``libss4k_hip.so``, declared in ``include/ss4k.h``) for all arithmetic.
There is no CPU fallback: constructing a context without the library or a GPU raises.
"""
__version__ = "0.1.0"
