"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement, in plain fp32 PyTorch ops and in the reference's own op order
(host loop over the time steps of the depth-first trace, dictionary of hidden
rows, stacks padded to 15 neighbours, per-node assembly loss loop), of the
training-side callers of the encoder path:

  fast_jtnn/nnutils.py:56-71        GRU (decoder form: bias on W_r, none on U_r)
  fast_jtnn/jtnn_dec.py:14-53       JTNNDecoder.__init__ / aggregate
  fast_jtnn/jtnn_dec.py:55-193      JTNNDecoder.forward (teacher forcing)
  fast_jtnn/jtnn_dec.py:299-305     dfs
  fast_jtnn/jtnn_vae.py:67-78       rsample (noise may be injected)
  fast_jtnn/jtnn_vae.py:85-103      JTNNVAE.forward
  fast_jtnn/jtnn_vae.py:105-138     JTNNVAE.assm
  fast_jtnn/jtprop_vae.py:62-93     property / denticity / isomer heads
  fast_jtnn/jtprop_vae.py:361-441   JTpropVAE.forward

Deliberate differences: nothing moves to CUDA (``create_var``), and ``rsample`` /
``forward`` accept the noise as an argument so that a run can be repeated with
the noise the reference drew.

Pinned by tests/golden/*.npz (keys ``fwd_*``, ``fg*``, ``prop_*``, ``pg*``), which
hold what the reference itself returned on the same batches and noise.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_encoder import OracleEncoder

MAX_NB = 15  # jtnn_dec.py:9


def GRU(x, h_nei, W_z, W_r, U_r, W_h):
    """nnutils.py:56-71."""
    hidden_size = x.size()[-1]
    sum_h = h_nei.sum(dim=1)
    z = torch.sigmoid(W_z(torch.cat([x, sum_h], dim=1)))
    r_1 = W_r(x).view(-1, 1, hidden_size)
    r_2 = U_r(h_nei)
    r = torch.sigmoid(r_1 + r_2)
    sum_gated_h = (r * h_nei).sum(dim=1)
    pre_h = torch.tanh(W_h(torch.cat([x, sum_gated_h], dim=1)))
    return (1.0 - z) * sum_h + z * pre_h


def dfs(stack, x, fa_idx):
    """jtnn_dec.py:299-305."""
    for y in x.neighbors:
        if y.idx == fa_idx:
            continue
        stack.append((x, y, 1))
        dfs(stack, y, x.idx)
        stack.append((y, x, 0))


class OracleDecoder(nn.Module):
    def __init__(self, vocab_size, hidden_size, latent_size, embedding):
        super().__init__()
        self.hidden_size, self.vocab_size = hidden_size, vocab_size
        self.embedding = embedding
        self.W_z = nn.Linear(2 * hidden_size, hidden_size)  # jtnn_dec.py:22-25
        self.U_r = nn.Linear(hidden_size, hidden_size, bias=False)
        self.W_r = nn.Linear(hidden_size, hidden_size)
        self.W_h = nn.Linear(2 * hidden_size, hidden_size)
        self.W = nn.Linear(hidden_size + latent_size, hidden_size)  # jtnn_dec.py:28
        self.U = nn.Linear(hidden_size + latent_size, hidden_size)  # jtnn_dec.py:31-32
        self.U_i = nn.Linear(2 * hidden_size, hidden_size)
        self.W_o = nn.Linear(hidden_size, vocab_size)  # jtnn_dec.py:35-36
        self.U_o = nn.Linear(hidden_size, 1)
        self.pred_loss = nn.CrossEntropyLoss(reduction="sum")  # size_average=False, jtnn_dec.py:39-40
        self.stop_loss = nn.BCEWithLogitsLoss(reduction="sum")

    def aggregate(self, hiddens, contexts, x_tree_vecs, mode):  # jtnn_dec.py:42-53
        V, V_o = (self.W, self.W_o) if mode == "word" else (self.U, self.U_o)
        tree_contexts = x_tree_vecs.index_select(0, contexts)
        return V_o(F.relu(V(torch.cat([hiddens, tree_contexts], dim=-1))))

    def forward(self, mol_batch, x_tree_vecs):  # jtnn_dec.py:55-193
        dt = x_tree_vecs.dtype
        pred_hiddens, pred_contexts, pred_targets = [], [], []
        stop_hiddens, stop_contexts, stop_targets = [], [], []
        traces = []
        for mol_tree in mol_batch:
            s = []
            dfs(s, mol_tree.nodes[0], -1)
            traces.append(s)
            for node in mol_tree.nodes:
                node.neighbors = []

        batch_size = len(mol_batch)
        pred_hiddens.append(torch.zeros(batch_size, self.hidden_size, dtype=dt))
        pred_targets.extend([mol_tree.nodes[0].wid for mol_tree in mol_batch])
        pred_contexts.append(torch.LongTensor(list(range(batch_size))))

        max_iter = max([len(tr) for tr in traces])
        padding = torch.zeros(self.hidden_size, dtype=dt)
        h = {}
        for t in range(max_iter):
            prop_list, batch_list = [], []
            for i, plist in enumerate(traces):
                if t < len(plist):
                    prop_list.append(plist[t])
                    batch_list.append(i)
            cur_x, cur_h_nei, cur_o_nei = [], [], []
            for node_x, real_y, _ in prop_list:
                cur_nei = [h[(node_y.idx, node_x.idx)] for node_y in node_x.neighbors if node_y.idx != real_y.idx]
                cur_h_nei.extend(cur_nei)
                cur_h_nei.extend([padding] * (MAX_NB - len(cur_nei)))
                cur_nei = [h[(node_y.idx, node_x.idx)] for node_y in node_x.neighbors]
                cur_o_nei.extend(cur_nei)
                cur_o_nei.extend([padding] * (MAX_NB - len(cur_nei)))
                cur_x.append(node_x.wid)
            cur_x = self.embedding(torch.LongTensor(cur_x))
            cur_h_nei = torch.stack(cur_h_nei, dim=0).view(-1, MAX_NB, self.hidden_size)
            new_h = GRU(cur_x, cur_h_nei, self.W_z, self.W_r, self.U_r, self.W_h)
            cur_o = torch.stack(cur_o_nei, dim=0).view(-1, MAX_NB, self.hidden_size).sum(dim=1)

            pred_target, pred_list, stop_target = [], [], []
            for i, (node_x, node_y, direction) in enumerate(prop_list):
                h[(node_x.idx, node_y.idx)] = new_h[i]
                node_y.neighbors.append(node_x)
                if direction == 1:
                    pred_target.append(node_y.wid)
                    pred_list.append(i)
                stop_target.append(direction)

            stop_hiddens.append(torch.cat([cur_x, cur_o], dim=1))
            stop_contexts.append(torch.LongTensor(batch_list))
            stop_targets.extend(stop_target)
            if len(pred_list) > 0:
                pred_contexts.append(torch.LongTensor([batch_list[i] for i in pred_list]))
                pred_hiddens.append(new_h.index_select(0, torch.LongTensor(pred_list)))
                pred_targets.extend(pred_target)

        cur_x, cur_o_nei = [], []  # last stop at the root, jtnn_dec.py:149-167
        for mol_tree in mol_batch:
            node_x = mol_tree.nodes[0]
            cur_x.append(node_x.wid)
            cur_nei = [h[(node_y.idx, node_x.idx)] for node_y in node_x.neighbors]
            cur_o_nei.extend(cur_nei)
            cur_o_nei.extend([padding] * (MAX_NB - len(cur_nei)))
        cur_x = self.embedding(torch.LongTensor(cur_x))
        cur_o = torch.stack(cur_o_nei, dim=0).view(-1, MAX_NB, self.hidden_size).sum(dim=1)
        stop_hiddens.append(torch.cat([cur_x, cur_o], dim=1))
        stop_contexts.append(torch.LongTensor(list(range(batch_size))))
        stop_targets.extend([0] * len(mol_batch))

        pred_contexts = torch.cat(pred_contexts, dim=0)  # jtnn_dec.py:169-178
        pred_hiddens = torch.cat(pred_hiddens, dim=0)
        pred_scores = self.aggregate(pred_hiddens, pred_contexts, x_tree_vecs, "word")
        pred_targets = torch.LongTensor(pred_targets)
        pred_loss = self.pred_loss(pred_scores, pred_targets) / len(mol_batch)
        _, preds = torch.max(pred_scores, dim=1)
        pred_acc = torch.eq(preds, pred_targets).float().sum() / pred_targets.nelement()

        stop_contexts = torch.cat(stop_contexts, dim=0)  # jtnn_dec.py:180-191
        stop_hiddens = F.relu(self.U_i(torch.cat(stop_hiddens, dim=0)))
        stop_scores = self.aggregate(stop_hiddens, stop_contexts, x_tree_vecs, "stop").squeeze(-1)
        stop_targets = torch.tensor(stop_targets, dtype=dt)
        stop_loss = self.stop_loss(stop_scores, stop_targets) / len(mol_batch)
        stops = torch.ge(stop_scores, 0).to(dt)
        stop_acc = torch.eq(stops, stop_targets).float().sum() / stop_targets.nelement()
        return pred_loss, stop_loss, pred_acc.item(), stop_acc.item()


class OracleJTNNVAE(OracleEncoder):
    """fast_jtnn/jtnn_vae.py:18-138 with the reference's attribute names (one state dict fits all)."""

    def __init__(self, vocab_size, hidden_size, latent_size, depthT, depthG):
        super().__init__(vocab_size, hidden_size, latent_size, depthT, depthG)
        self.decoder = OracleDecoder(vocab_size, hidden_size, self.latent_size,
                                     nn.Embedding(vocab_size, hidden_size))
        self.assm_loss = nn.CrossEntropyLoss(reduction="sum")

    def rsample(self, z_vecs, W_mean, W_var, epsilon=None):  # jtnn_vae.py:67-78
        batch_size = z_vecs.size(0)
        z_mean = W_mean(z_vecs)
        z_log_var = -torch.abs(W_var(z_vecs))
        kl_loss = -0.5 * torch.sum(1.0 + z_log_var - z_mean * z_mean - torch.exp(z_log_var)) / batch_size
        if epsilon is None:
            epsilon = torch.randn_like(z_mean)
        return z_mean + torch.exp(z_log_var / 2) * epsilon, kl_loss

    def assm(self, mol_batch, jtmpn_holder, x_mol_vecs, x_tree_mess):  # jtnn_vae.py:105-138
        jtmpn_holder, batch_idx = jtmpn_holder
        fatoms, fbonds, agraph, bgraph, scope = jtmpn_holder
        cand_vecs = self.jtmpn(fatoms, fbonds, agraph, bgraph, scope, x_tree_mess)
        x_mol_vecs = self.A_assm(x_mol_vecs.index_select(0, batch_idx))
        scores = torch.bmm(x_mol_vecs.unsqueeze(1), cand_vecs.unsqueeze(-1)).squeeze()
        cnt, tot, acc = 0, 0, 0
        all_loss = []
        for mol_tree in mol_batch:
            comp_nodes = [node for node in mol_tree.nodes if len(node.cands) > 1 and not node.is_leaf]
            cnt += len(comp_nodes)
            for node in comp_nodes:
                label = node.cands.index(node.label)
                ncand = len(node.cands)
                cur_score = scores.narrow(0, tot, ncand)
                tot += ncand
                if cur_score.data[label] >= cur_score.max().item():
                    acc += 1
                all_loss.append(self.assm_loss(cur_score.view(1, -1), torch.LongTensor([label])))
        return sum(all_loss) / len(mol_batch), acc * 1.0 / cnt

    def forward_terms(self, x_batch, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder, epsilons):
        x_tree_vecs, x_tree_mess, x_mol_vecs = self.encode(x_jtenc_holder, x_mpn_holder)
        e_t, e_g = epsilons if epsilons is not None else (None, None)
        z_tree_vecs, tree_kl = self.rsample(x_tree_vecs, self.T_mean, self.T_var, e_t)
        z_mol_vecs, mol_kl = self.rsample(x_mol_vecs, self.G_mean, self.G_var, e_g)
        word_loss, topo_loss, word_acc, topo_acc = self.decoder(x_batch, z_tree_vecs)
        assm_loss, assm_acc = self.assm(x_batch, x_jtmpn_holder, z_mol_vecs, x_tree_mess)
        return dict(z_tree=z_tree_vecs, z_mol=z_mol_vecs, tree_kl=tree_kl, mol_kl=mol_kl, word_loss=word_loss,
                    topo_loss=topo_loss, word_acc=word_acc, topo_acc=topo_acc, assm_loss=assm_loss,
                    assm_acc=assm_acc)

    def forward(self, x_batch, beta, epsilons=None):  # jtnn_vae.py:85-103
        x_batch, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder = x_batch
        t = self.forward_terms(x_batch, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder, epsilons)
        kl_div = t["tree_kl"] + t["mol_kl"]
        self.last_terms = t
        return (t["word_loss"] + t["topo_loss"] + t["assm_loss"] + beta * kl_div, kl_div.item(), t["word_acc"],
                t["topo_acc"], t["assm_acc"])


class OracleJTpropVAE(OracleJTNNVAE):
    """fast_jtnn/jtprop_vae.py:27-96, 361-441."""

    def __init__(self, vocab_size, hidden_size, latent_size, depthT, depthG, train_mode=()):
        super().__init__(vocab_size, hidden_size, latent_size, depthT, depthG)
        self.propNN = nn.Sequential(nn.Linear(self.latent_size * 2, hidden_size), nn.Tanh(),
                                    nn.Linear(hidden_size, 2))
        self.train_mode = list(train_mode)
        for term in self.train_mode:
            if term == "denticity":
                self.denticityNN = nn.Sequential(nn.Linear(self.latent_size * 2, hidden_size), nn.Sigmoid(),
                                                 nn.Linear(hidden_size, 1))
                self.denticityNN_loss = nn.BCEWithLogitsLoss()
            elif term == "isomer":
                self.isomerNN = nn.Sequential(nn.Linear(self.latent_size * 2, hidden_size), nn.Sigmoid(),
                                              nn.Linear(hidden_size, 3))
                self.isomerNN_loss = nn.CrossEntropyLoss()
            else:
                raise ValueError("train_mode is not a valid value")
        self.prop_loss = nn.MSELoss()

    def forward(self, x_batch, beta, epsilons=None):
        x_batch, x_prop_holder, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder = x_batch
        length_p = len(x_prop_holder[0])
        x_prop = x_prop_holder[:, 0:2]
        if length_p == 2:
            pass
        elif length_p == 3:
            x_dent = x_prop_holder[:, 2]
        elif length_p == 4:
            x_dent = x_prop_holder[:, 2]
            x_isomer = x_prop_holder[:, 3]
        else:
            raise ValueError("The correct number of properties could not be unpacked")
        t = self.forward_terms(x_batch, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder, epsilons)
        kl_div = t["tree_kl"] + t["mol_kl"]
        all_vec = torch.cat([t["z_tree"], t["z_mol"]], dim=1)
        prop_loss = self.prop_loss(self.propNN(all_vec).squeeze(), x_prop[:, 0:2].to(all_vec.dtype))
        for term in self.train_mode:
            if term == "denticity":
                # FloatTensor in the reference; the module's dtype here so that the same code also runs in fp64
                dent_loss = self.denticityNN_loss(self.denticityNN(all_vec).squeeze(), x_dent.to(all_vec.dtype))
            elif term == "isomer":
                isomer_loss = self.isomerNN_loss(self.isomerNN(all_vec).squeeze(), x_isomer.type(torch.LongTensor))
        standard_loss = t["word_loss"] + t["topo_loss"] + t["assm_loss"] + beta * kl_div + prop_loss
        log_metrics = {"kl_div": kl_div.item(), "word_acc": t["word_acc"] * 100, "topo_acc": t["topo_acc"] * 100,
                       "assm_acc": t["assm_acc"] * 100, "prop_loss": prop_loss.item() * 100}
        for term in self.train_mode:
            if term == "denticity":
                standard_loss = standard_loss + dent_loss
                log_metrics["dent_loss"] = dent_loss.item() * 100
            elif term == "isomer":
                standard_loss = standard_loss + isomer_loss
                log_metrics["isomer_loss"] = isomer_loss.item() * 100
        return standard_loss, log_metrics
