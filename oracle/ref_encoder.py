"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement, in plain fp32 PyTorch ops, of the encoder message-passing path
of the reference, in the reference's own op order (materialised [N,K,H] gathers,
per-neighbour U_r product, concatenated Linear inputs, Python readout loops).
Gradients come from torch autograd, as in the reference.

Follows:
  fast_jtnn/nnutils.py:13-18      index_select_ND
  fast_jtnn/mpn.py:72-108         MPN.__init__ / forward
  fast_jtnn/jtnn_enc.py:9-42      JTNNEncoder.__init__ / forward
  fast_jtnn/jtnn_enc.py:96-128    GraphGRU
  fast_jtnn/jtmpn.py:72-111       JTMPN.__init__ / forward
  fast_jtnn/jtnn_vae.py:56-65     encode_latent (latent heads)
  fast_jtnn/jtnn_vae.py:105-114   assm scoring (bilinear candidate score)

The only deliberate differences: ``create_var`` (nnutils.py:6-10) moves tensors to
CUDA unconditionally, here everything stays on the CPU; and the two tensors the
reference creates itself (the zero start state and the row-0 mask) take the
module's dtype, so that ``.double()`` gives an fp64 run of the same op sequence
(used by the tests to tell kernel error from ReLU-kink noise).

Pinned by tests/golden/*.npz, which hold outputs of the reference itself
(imported from its source tree when the fixtures were generated).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def index_select_ND(source, dim, index):
    """nnutils.py:13-18."""
    final_size = index.size() + source.size()[1:]
    return source.index_select(dim, index.view(-1)).view(final_size)


class OracleMPN(nn.Module):
    def __init__(self, hidden_size, depth, atom_fdim=39, bond_fdim=12):
        super().__init__()
        self.hidden_size, self.depth = hidden_size, depth
        self.W_i = nn.Linear(atom_fdim + bond_fdim, hidden_size, bias=False)  # mpn.py:77
        self.W_h = nn.Linear(hidden_size, hidden_size, bias=False)  # mpn.py:78
        self.W_o = nn.Linear(atom_fdim + hidden_size, hidden_size)  # mpn.py:79

    def forward(self, fatoms, fbonds, agraph, bgraph, scope):
        binput = self.W_i(fbonds)  # mpn.py:87
        message = F.relu(binput)
        for _ in range(self.depth - 1):  # mpn.py:90-94
            nei = index_select_ND(message, 0, bgraph).sum(dim=1)
            nei = self.W_h(nei)
            message = F.relu(binput + nei)
        nei = index_select_ND(message, 0, agraph).sum(dim=1)  # mpn.py:96-99
        atom_hiddens = F.relu(self.W_o(torch.cat([fatoms, nei], dim=1)))
        vecs = [atom_hiddens[st:st + le].mean(dim=0) for st, le in scope]  # mpn.py:103-105
        return torch.stack(vecs, dim=0)


class OracleGraphGRU(nn.Module):
    def __init__(self, input_size, hidden_size, depth):
        super().__init__()
        self.hidden_size, self.input_size, self.depth = hidden_size, input_size, depth
        self.W_z = nn.Linear(input_size + hidden_size, hidden_size)  # jtnn_enc.py:102-105
        self.W_r = nn.Linear(input_size, hidden_size, bias=False)
        self.U_r = nn.Linear(hidden_size, hidden_size)
        self.W_h = nn.Linear(input_size + hidden_size, hidden_size)

    def forward(self, h, x, mess_graph):
        mask = torch.ones(h.size(0), 1, dtype=h.dtype)  # dtype follows the module so it also runs in fp64
        mask[0] = 0  # jtnn_enc.py:108-109: row 0 is the padding message
        for _ in range(self.depth):  # jtnn_enc.py:111-126
            h_nei = index_select_ND(h, 0, mess_graph)
            sum_h = h_nei.sum(dim=1)
            z = torch.sigmoid(self.W_z(torch.cat([x, sum_h], dim=1)))
            r_1 = self.W_r(x).view(-1, 1, self.hidden_size)
            r_2 = self.U_r(h_nei)
            r = torch.sigmoid(r_1 + r_2)
            sum_gated_h = (r * h_nei).sum(dim=1)
            pre_h = torch.tanh(self.W_h(torch.cat([x, sum_gated_h], dim=1)))
            h = (1.0 - z) * sum_h + z * pre_h
            h = h * mask
        return h


class OracleJTNNEncoder(nn.Module):
    def __init__(self, hidden_size, depth, embedding):
        super().__init__()
        self.hidden_size, self.depth = hidden_size, depth
        self.embedding = embedding
        self.outputNN = nn.Sequential(nn.Linear(2 * hidden_size, hidden_size), nn.ReLU())
        self.GRU = OracleGraphGRU(hidden_size, hidden_size, depth=depth)

    def forward(self, fnode, fmess, node_graph, mess_graph, scope):
        messages = torch.zeros(mess_graph.size(0), self.hidden_size,
                               dtype=self.embedding.weight.dtype)  # jtnn_enc.py:25
        fnode = self.embedding(fnode)  # jtnn_enc.py:27-29
        fmess = index_select_ND(fnode, 0, fmess)
        messages = self.GRU(messages, fmess, mess_graph)
        mess_nei = index_select_ND(messages, 0, node_graph)  # jtnn_enc.py:31-33
        node_vecs = self.outputNN(torch.cat([fnode, mess_nei.sum(dim=1)], dim=-1))
        tree_vecs = torch.stack([node_vecs[st] for st, _ in scope], dim=0)  # jtnn_enc.py:36-41
        return tree_vecs, messages


class OracleJTMPN(nn.Module):
    def __init__(self, hidden_size, depth, atom_fdim=38, bond_fdim=6):
        super().__init__()
        self.hidden_size, self.depth = hidden_size, depth
        self.W_i = nn.Linear(atom_fdim + bond_fdim, hidden_size, bias=False)  # jtmpn.py:77-79
        self.W_h = nn.Linear(hidden_size, hidden_size, bias=False)
        self.W_o = nn.Linear(atom_fdim + hidden_size, hidden_size)

    def forward(self, fatoms, fbonds, agraph, bgraph, scope, tree_message):
        binput = self.W_i(fbonds)  # jtmpn.py:89-90
        graph_message = F.relu(binput)
        for _ in range(self.depth - 1):  # jtmpn.py:92-97
            message = torch.cat([tree_message, graph_message], dim=0)
            nei = index_select_ND(message, 0, bgraph).sum(dim=1)
            nei = self.W_h(nei)
            graph_message = F.relu(binput + nei)
        message = torch.cat([tree_message, graph_message], dim=0)  # jtmpn.py:99-103
        nei = index_select_ND(message, 0, agraph).sum(dim=1)
        atom_hiddens = F.relu(self.W_o(torch.cat([fatoms, nei], dim=1)))
        vecs = [atom_hiddens.narrow(0, st, le).sum(dim=0) / le for st, le in scope]  # jtmpn.py:106-108
        return torch.stack(vecs, dim=0)


class OracleEncoder(nn.Module):
    """The encoder half of JTNNVAE with the reference's attribute names, so one
    state dict fits the reference, this oracle and the HIP modules alike
    (fast_jtnn/jtnn_vae.py:27-43)."""

    def __init__(self, vocab_size, hidden_size, latent_size, depthT, depthG):
        super().__init__()
        self.hidden_size = hidden_size
        self.latent_size = latent_size = latent_size // 2
        self.jtnn = OracleJTNNEncoder(hidden_size, depthT, nn.Embedding(vocab_size, hidden_size))
        self.jtmpn = OracleJTMPN(hidden_size, depthG)
        self.mpn = OracleMPN(hidden_size, depthG)
        self.A_assm = nn.Linear(latent_size, hidden_size, bias=False)
        self.T_mean = nn.Linear(hidden_size, latent_size)
        self.T_var = nn.Linear(hidden_size, latent_size)
        self.G_mean = nn.Linear(hidden_size, latent_size)
        self.G_var = nn.Linear(hidden_size, latent_size)

    def encode(self, jtenc_holder, mpn_holder):  # jtnn_vae.py:45-48
        tree_vecs, tree_mess = self.jtnn(*jtenc_holder)
        mol_vecs = self.mpn(*mpn_holder)
        return tree_vecs, tree_mess, mol_vecs

    def encode_latent(self, jtenc_holder, mpn_holder):  # jtnn_vae.py:56-65
        tree_vecs, _ = self.jtnn(*jtenc_holder)
        mol_vecs = self.mpn(*mpn_holder)
        tree_mean = self.T_mean(tree_vecs)
        mol_mean = self.G_mean(mol_vecs)
        tree_var = -torch.abs(self.T_var(tree_vecs))
        mol_var = -torch.abs(self.G_var(mol_vecs))
        return torch.cat([tree_mean, mol_mean], dim=1), torch.cat([tree_var, mol_var], dim=1)
