"""pintron_amd -- MI355X-native accelerator for PIntron's est-fact stage.

The product is the C-ABI library ``pintron_amd/lib/libpintron_gpu.so`` (hand-written HIP for gfx950,
see ``include/pintron_gpu.h``) and the C host program built on it.  This Python package is only
plumbing for tests, ``bench.py`` and multi-GPU launches: a ctypes binding of that ABI (``capi``),
sessions of the host program and the EST-sharded driver (``estfact``), the workload generator
(``synth``).  There is no CPU fallback:
:func:`pintron_amd.capi.lib` fails loudly when the library has not been built.
"""
__all__ = ["capi", "estfact", "synth"]
