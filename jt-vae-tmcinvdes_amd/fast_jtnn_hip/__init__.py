"""MI355X-native encoder message passing for JT-VAE: drop-in replacements for
``fast_jtnn``'s ``MPN``, ``JTNNEncoder``/``GraphGRU`` and ``JTMPN`` (same names, as
re-exported by fast_jtnn/__init__.py:3-8) on hand-written HIP kernels behind a C ABI.

Importing this package loads ``libjtvae_hip.so``; it fails if the library has not
been built - there is no CPU implementation in the product path.
"""
from . import _capi  # noqa: F401  (loads the shared library or raises)
from .jtmpn import JTMPN
from .jtnn_dec import JTNNDecoder
from .jtnn_enc import GraphGRU, JTNNEncoder
from .jtnn_vae import JTNNVAE, JTpropVAE
from .mpn import MPN
from .nnutils import create_var, gather_sum, index_select_ND

__all__ = ["MPN", "JTNNEncoder", "GraphGRU", "JTMPN", "JTNNDecoder", "JTNNVAE", "JTpropVAE", "create_var",
           "gather_sum", "index_select_ND"]
