"""Candidate-subgraph message passing network on the HIP kernels.

Drop-in for ``fast_jtnn.jtmpn.JTMPN`` (fast_jtnn/jtmpn.py:71-183): same
constructor, parameters (``W_i.weight [H,44]``, ``W_h.weight [H,H]``,
``W_o.{weight [H,38+H],bias}``) and ``forward(fatoms, fbonds, agraph, bgraph,
scope, tree_message) -> [n_candidates, H]``.

The reference concatenates ``[tree_message ; graph_message]`` three times per call
(jtmpn.py:93,99) only to index into it; here the gather kernel reads from the two
buffers directly (index < len(tree_message) selects the tree part).  Gradients
flow back into ``tree_message`` (and from there into the tree GRU) through the
transposed graph.  Row 0 of ``tree_message`` is the zero padding message
(jtmpn.py:83 "tree_message[0] == vec(0)"); padding slots are skipped rather than
added.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import graphprep
from .mpn import _GraphMPNFunction
from .synthetic import JTMPN_ATOM_FDIM, JTMPN_BOND_FDIM, JTMPN_MAX_NB

ATOM_FDIM = JTMPN_ATOM_FDIM  # 38, fast_jtnn/jtmpn.py:37
BOND_FDIM = JTMPN_BOND_FDIM  # 6,  fast_jtnn/jtmpn.py:38
MAX_NB = JTMPN_MAX_NB  # 15, fast_jtnn/jtmpn.py:39


class JTMPN(nn.Module):
    def __init__(self, hidden_size, depth):
        super(JTMPN, self).__init__()
        self.hidden_size = hidden_size
        self.depth = depth
        self.W_i = nn.Linear(ATOM_FDIM + BOND_FDIM, hidden_size, bias=False)
        self.W_h = nn.Linear(hidden_size, hidden_size, bias=False)
        self.W_o = nn.Linear(ATOM_FDIM + hidden_size, hidden_size)

    def forward(self, fatoms, fbonds, agraph, bgraph, scope, tree_message):
        device = self.W_h.weight.device
        if device.type != "cuda":
            raise RuntimeError("fast_jtnn_hip.JTMPN runs on an MI355X only (module is on %s); there is no "
                               "CPU path" % device)
        if tree_message.dim() != 2 or tree_message.shape[1] != self.hidden_size:
            raise ValueError("tree_message must be [n_messages, %d]" % self.hidden_size)
        if tree_message.device != device:
            raise ValueError("tree_message is on %s, module on %s" % (tree_message.device, device))
        n_tree = int(tree_message.shape[0])
        key = bgraph if isinstance(bgraph, torch.Tensor) else None
        build = lambda: graphprep.prepare_mpn((fatoms, fbonds, agraph, bgraph, scope), device, n_tree,
                                                torch.is_grad_enabled())  # noqa: E731
        prep = build() if key is None else graphprep.cached_prep(key, ("jtmpn", str(device), n_tree), build)
        if prep.bond_in_fdim != self.W_i.in_features or prep.atom_fdim + self.hidden_size != self.W_o.in_features:
            raise ValueError("feature widths %d/%d do not match W_i/W_o" % (prep.bond_in_fdim, prep.atom_fdim))
        with torch.cuda.device(device):
            return _GraphMPNFunction.apply(self.W_i.weight, self.W_h.weight, self.W_o.weight, self.W_o.bias,
                                           tree_message, prep, self.depth, torch.is_grad_enabled())

    @staticmethod
    def tensorize(cand_batch, mess_dict):
        """Same holder as fast_jtnn/jtmpn.py:113-183, from abstract candidate graphs
        (objects with ``atom_feats``, ``bonds``, ``bond_feats``, ``atom_nid``, ``all_nodes``)."""
        from .tensorize import jtmpn_tensorize

        return jtmpn_tensorize(cand_batch, mess_dict)
