"""hugectr_amd -- MI355X-native sparse-embedding hot path behind HugeCTR's interfaces.

Importing this package loads libhugectr_amd.so (HIP, gfx950); it fails loudly when the library
is missing.  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib  # noqa: F401  (loads the HIP library or raises)
from ._lib import HugeCTRAmdError  # noqa: F401
from .embedding import OptParams, SparseEmbeddingHash, backward_reorder, forward_reorder  # noqa: F401
from .layers import (InteractionLayer, MultiCrossLayer, interaction,  # noqa: F401
                     interaction_gather, interaction_indexed)

__version__ = "0.1.0"
from .embedding_collection import (EmbeddingCollection, EmbeddingCollectionConfig,  # noqa: F401,E402
                                   EmbeddingTableConfig)
