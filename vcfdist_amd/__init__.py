"""vcfdist_amd -- MI355X-native precision/recall alignment path of vcfdist.

Host-side mirror of the reference interface for this path (precision_recall_wrapper,
src/dist.cpp:1731-1904) over the C ABI in include/vcfdist_pr.h."""
from . import _abi
from ._abi import Batch, Results, Variants, default_config

__all__ = ["_abi", "Batch", "Results", "Variants", "default_config"]
