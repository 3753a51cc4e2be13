"""ucod_dpl_amd -- MI355X (gfx950) native hot path of UCOD-DPL.

Layout: ``csrc/`` HIP kernels + C ABI (include/ucod_dpl.h), ``native.py`` ctypes binding,
``ops.py`` functional wrappers, ``vit_engine.py`` frozen backbone, and the host-side mirror of the
reference's module interface (``models/``, ``data/``, ``engine/``, ``configs/``).
"""
__version__ = "0.1.0"
