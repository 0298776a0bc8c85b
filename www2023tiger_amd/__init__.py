"""www2023tiger_amd - MI355X (gfx950) engine for the TIGER event-batch hot path.

Host side mirrors the reference's Python interface (tiger.data.graph.Graph,
tiger.data.data_loader.GraphCollator, tiger.model.*) over the C ABI of
csrc/libtiger_hip.so (include/tiger_hip.h).  Importing the package loads the
library; there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)

__version__ = '0.1.0'
