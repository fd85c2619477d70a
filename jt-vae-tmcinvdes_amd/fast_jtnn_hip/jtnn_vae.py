"""Model wrappers that own the three encoder modules under the reference's
attribute names, so a reference state dict loads unchanged.

Mirrors the encoder-side surface of ``fast_jtnn.jtnn_vae.JTNNVAE``
(fast_jtnn/jtnn_vae.py:18-138) and ``fast_jtnn.jtprop_vae.JTpropVAE``
(fast_jtnn/jtprop_vae.py:27-136, 361-476): ``encode``, ``encode_latent``,
``rsample``, ``assm`` and ``forward``, with the teacher-forced tree decoder of
``jtnn_dec.py`` under ``decoder`` as in the reference.  ``decode`` (greedy
assembly of a molecule, RDKit-driven, batch of one) stays with the reference.
``rsample`` and ``forward`` accept the noise as an optional argument so that a
step can be repeated exactly (tests, N-rank == 1-rank checks).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _capi as K
from . import graphprep
from .jtmpn import JTMPN
from .jtnn_dec import JTNNDecoder
from .jtnn_enc import JTNNEncoder
from .mpn import MPN, as_row_buffer, dense_view, pack_weights


class _SegmentCE(torch.autograd.Function):
    """scores [NC], seg_ptr int32 [S+1], label_pos int32 [S] -> (loss sum, hits): one kernel forward,
    one multiply backward."""

    @staticmethod
    def forward(ctx, scores, seg_ptr, label_pos):
        from . import _capi as K
        S = label_pos.shape[0]
        sc = scores.detach().contiguous().float()
        loss_seg = torch.empty((S,), dtype=torch.float32, device=sc.device)
        hit_seg = torch.empty((S,), dtype=torch.int32, device=sc.device)
        dsc = torch.empty_like(sc)
        with torch.cuda.device(sc.device):
            K.segment_ce(sc.data_ptr(), seg_ptr.data_ptr(), label_pos.data_ptr(), S, loss_seg.data_ptr(),
                         hit_seg.data_ptr(), dsc.data_ptr(), n_scores=sc.shape[0])
        ctx.save_for_backward(dsc)
        ctx.mark_non_differentiable(hit_seg)
        return loss_seg.sum(), hit_seg

    @staticmethod
    def backward(ctx, dloss, _):
        (dsc,) = ctx.saved_tensors
        return dsc * dloss, None, None


def segment_cross_entropy(scores: torch.Tensor, seg_len: torch.Tensor, label_pos: torch.Tensor, seg_ptr=None):
    """Sum over segments of CrossEntropy(scores[segment], label) and the number of segments
    whose label scores highest - the loop of fast_jtnn/jtnn_vae.py:116-135 as one kernel, without
    one launch and one host sync per tree node.

    scores [NC]; seg_len [S] candidates per node (consecutive); label_pos [S] global index of the label;
    seg_ptr: int32 [S + 1] on the device if the caller has it already (seg_len is then not read).
    """
    if scores.device.type != "cuda":
        raise RuntimeError("segment_cross_entropy runs on the GPU only (got %s)" % scores.device)
    if seg_ptr is None:
        seg_ptr = torch.zeros(seg_len.shape[0] + 1, dtype=torch.int32, device=scores.device)
        seg_ptr[1:] = torch.cumsum(seg_len.to(scores.device), 0)
    loss, hits = _SegmentCE.apply(scores, seg_ptr, label_pos.to(device=scores.device, dtype=torch.int32))
    return loss, hits.sum()


class _RowDot(torch.autograd.Function):
    """score[c] = <q[owner[c]], cand[c]> (the bmm of fast_jtnn/jtnn_vae.py:114 without repeating the
    projected molecule vector once per candidate); owner's candidates are consecutive (own_ptr)."""

    @staticmethod
    def forward(ctx, q, cand, owner, own_ptr):
        H = q.shape[1]
        ld = K.round_ld(H)
        with torch.cuda.device(q.device):
            qp, cp = as_row_buffer(q, H, ld), as_row_buffer(cand, H, ld)
            score = torch.empty((cand.shape[0],), dtype=torch.float32, device=q.device)
            K.row_dot(score.data_ptr(), qp.data_ptr(), owner.data_ptr(), cp.data_ptr(), cand.shape[0], H, ld,
                      n_owner=q.shape[0])
        ctx.keep = (qp, cp, owner, own_ptr, H, ld)
        return score

    @staticmethod
    def backward(ctx, dscore):
        qp, cp, owner, own_ptr, H, ld = ctx.keep
        nc, nb = cp.shape[0], qp.shape[0]
        ds = dscore.contiguous().float()
        with torch.cuda.device(ds.device):
            dq, dc = K.rows(nb, ld, ds.device), K.rows(nc, ld, ds.device)
            K.row_dot_bwd(ds.data_ptr(), qp.data_ptr(), owner.data_ptr(), cp.data_ptr(), own_ptr.data_ptr(), nc, nb,
                          dq.data_ptr(), dc.data_ptr(), H, ld)
        return dense_view(dq, H), dense_view(dc, H), None, None


class _LatentHeads(torch.autograd.Function):
    """x [B, H] -> (W_m x + b_m, -|W_v x + b_v|): the mean and log-variance heads of one encoder
    (fast_jtnn/jtnn_vae.py:59-64, 69-70).  Forward: two products with the bias in their epilogue; backward: two
    dgrad products summed into one buffer and two weight-gradient products that carry the bias gradient.  The
    ``nn.Linear`` calls they replace cost ~0.1 ms of host time each through the BLAS library (twelve calls per
    encoder and step, forward and backward), which a step of a few hundred molecules cannot hide."""

    @staticmethod
    def forward(ctx, x, Wm, bm, Wv, bv):
        dev = x.device
        B, H = x.shape
        L = Wm.shape[0]
        ld, lp = K.round_ld(H), K.round_ld(L)
        with torch.cuda.device(dev):
            X = as_row_buffer(x, H, ld)
            Wm_p, Wv_p = pack_weights([(Wm, 0, H, False), (Wv, 0, H, False)])
            Y = torch.empty((2, B, lp), dtype=torch.float32, device=dev)
            for i, (W_p, b) in enumerate(((Wm_p, bm), (Wv_p, bv))):
                K.gemm_nt(X.data_ptr(), ld, W_p.data_ptr(), ld, Y[i].data_ptr(), lp, B, L, H,
                          bias=None if b is None else b.detach().data_ptr())
            pre = Y[1, :, :L]
            log_var = torch.abs(pre).neg_()
        if any(ctx.needs_input_grad):
            ctx.keep = (X, pre, Wm, Wv, H, L, ld, lp, bm is not None, bv is not None)
        ctx.set_materialize_grads(False)
        return Y[0, :, :L], log_var

    @staticmethod
    def backward(ctx, dmean, dlog_var):
        X, pre, Wm, Wv, H, L, ld, lp, has_bm, has_bv = ctx.keep
        dev = X.device
        B = X.shape[0]
        with torch.cuda.device(dev):
            dm = None if dmean is None else as_row_buffer(dmean, L, lp)
            # d(-|p|)/dp = -sign(p) (0 at p = 0, as torch.abs differentiates)
            dv = None if dlog_var is None else as_row_buffer(torch.sign(pre).neg_().mul_(dlog_var), L, lp)
            dX = None
            if ctx.needs_input_grad[0] and (dm is not None or dv is not None):
                WmT, WvT = pack_weights([(Wm, 0, H, True), (Wv, 0, H, True)])
                dX = K.rows(B, ld, dev)
                first = True
                for d, WT in ((dm, WmT), (dv, WvT)):
                    if d is None:
                        continue
                    K.gemm_nt(d.data_ptr(), lp, WT.data_ptr(), lp, dX.data_ptr(), ld, B, H, L,
                              add=None if first else dX.data_ptr(), ld_add=ld)
                    first = False
                dX = dense_view(dX, H)
            grads = []
            for d, W, has_b, need_w, need_b in ((dm, Wm, has_bm, ctx.needs_input_grad[1], ctx.needs_input_grad[2]),
                                                (dv, Wv, has_bv, ctx.needs_input_grad[3], ctx.needs_input_grad[4])):
                if d is None or not (need_w or (has_b and need_b)):
                    grads += [None, None]
                    continue
                dW = torch.empty((L, H), dtype=torch.float32, device=dev)
                db = torch.empty((L,), dtype=torch.float32, device=dev) if has_b else None
                K.gemm_tn(d.data_ptr(), lp, X.data_ptr(), ld, dW.data_ptr(), H, B, L, H, False, dev,
                          dbias=None if db is None else db.data_ptr())
                grads += [dW, db]
        return (dX, *grads)


class JTNNVAE(nn.Module):
    def __init__(self, vocab, hidden_size, latent_size, depthT, depthG, decoder=None):
        """``decoder``: None builds the in-package teacher-forced decoder (as the reference does,
        jtnn_vae.py:30-32); False leaves it out (encoder-only use); a module is taken as it is."""
        super(JTNNVAE, self).__init__()
        self.vocab = vocab
        self.hidden_size = hidden_size
        self.latent_size = latent_size = latent_size // 2  # tree and graph halves (jtnn_vae.py:23-25)

        self.jtnn = JTNNEncoder(hidden_size, depthT, nn.Embedding(vocab.size(), hidden_size))
        if decoder is None:
            decoder = JTNNDecoder(vocab, hidden_size, latent_size, nn.Embedding(vocab.size(), hidden_size))
        if decoder is not False:
            self.decoder = decoder
        self.jtmpn = JTMPN(hidden_size, depthG)
        self.mpn = MPN(hidden_size, depthG)

        self.A_assm = nn.Linear(latent_size, hidden_size, bias=False)
        self.T_mean = nn.Linear(hidden_size, latent_size)
        self.T_var = nn.Linear(hidden_size, latent_size)
        self.G_mean = nn.Linear(hidden_size, latent_size)
        self.G_var = nn.Linear(hidden_size, latent_size)

    # The tree encoder and the graph encoder share nothing until their outputs meet, and the tail
    # iterations of the tree GRU are short launches that leave most of the GPU idle: on a GPU the
    # two can run on two side streams, so that the graph encoder's large GEMMs fill those gaps (autograd
    # replays each backward on the stream of its forward, so the backward passes overlap as well).
    # The launches and their data are those of the one-stream order; only the tiling of the tree encoder's short
    # products follows the stream layout (JTNNEncoder.crowded), so results agree to rounding by default and bit for bit
    # with that choice pinned.  Off by default: the reference issues
    # everything on one stream, and per-kernel timings are only meaningful without co-running kernels.
    overlap_streams = False
    # Batches below this many tree messages (about 150 molecules) stay on one stream even with overlap_streams set:
    # the forks and joins cost ~0.4 ms of host time per step, and a step that small is bound by the host that issues it
    # (128 molecules: 1.86 ms on one stream, 2.13 ms on two; 256 molecules: 2.50 against 2.27).
    overlap_min_messages = 6000

    def _side_streams(self, device):
        cache = self.__dict__.setdefault("_stream_cache", {})
        key = (device.type, device.index)
        if key not in cache:
            cache[key] = tuple(torch.cuda.Stream(device=device) for _ in range(3))
        return cache[key]

    def encode(self, jtenc_holder, mpn_holder):  # jtnn_vae.py:45-48
        device = self.mpn.W_h.weight.device
        if not (self.overlap_streams and device.type == "cuda"
                and int(jtenc_holder[3].shape[0]) >= self.overlap_min_messages):
            self.jtnn.side_streams = ()
            tree_vecs, tree_mess = self.jtnn(*jtenc_holder)
            mol_vecs = self.mpn(*mpn_holder)
            return tree_vecs, tree_mess, mol_vecs
        s_tree, s_graph, s_aux = self._side_streams(device)
        # the tree encoder's backward pass ends in groups of short launches that can run side by side (a second
        # and third helper stream were measured and add nothing to the first)
        self.jtnn.side_streams = (s_aux,)
        cur = torch.cuda.current_stream(device)
        s_tree.wait_stream(cur)
        s_graph.wait_stream(cur)
        with torch.cuda.stream(s_tree):
            tree_vecs, tree_mess = self.jtnn(*jtenc_holder)
        with torch.cuda.stream(s_graph):
            mol_vecs = self.mpn(*mpn_holder)
        cur.wait_stream(s_tree)
        cur.wait_stream(s_graph)
        for t in (tree_vecs, tree_mess, mol_vecs):  # allocated on a side stream, consumed on the caller's
            t.record_stream(cur)
        return tree_vecs, tree_mess, mol_vecs

    def encode_latent(self, jtenc_holder, mpn_holder):  # jtnn_vae.py:56-65
        tree_vecs, _, mol_vecs = self.encode(jtenc_holder, mpn_holder)
        tree_mean, tree_var = self.latent_heads(tree_vecs, self.T_mean, self.T_var)
        mol_mean, mol_var = self.latent_heads(mol_vecs, self.G_mean, self.G_var)
        return torch.cat([tree_mean, mol_mean], dim=1), torch.cat([tree_var, mol_var], dim=1)

    @staticmethod
    def latent_heads(z_vecs, W_mean, W_var):
        """(W_mean z, -|W_var z|) of fast_jtnn/jtnn_vae.py:59-64 / 69-70 on the package's own products."""
        if z_vecs.device.type != "cuda":
            raise RuntimeError("fast_jtnn_hip latent heads run on an MI355X only (input is on %s); there is no CPU "
                               "path" % z_vecs.device)
        return _LatentHeads.apply(z_vecs, W_mean.weight, W_mean.bias, W_var.weight, W_var.bias)

    def encode_vectors(self, jtenc_holder, mpn_holder):
        """[B, 2H] = cat(tree_vecs, mol_vecs): what bulk latent generation stores
        (``encode_from_smiles``, jtnn_vae.py:50-54, after its RDKit front end)."""
        tree_vecs, _, mol_vecs = self.encode(jtenc_holder, mpn_holder)
        return torch.cat([tree_vecs, mol_vecs], dim=-1)

    def rsample(self, z_vecs, W_mean, W_var, epsilon=None):  # jtnn_vae.py:67-78
        batch_size = z_vecs.size(0)
        z_mean, z_log_var = self.latent_heads(z_vecs, W_mean, W_var)
        kl_loss = -0.5 * torch.sum(1.0 + z_log_var - z_mean * z_mean - torch.exp(z_log_var)) / batch_size
        if epsilon is None:
            epsilon = torch.randn_like(z_mean)
        z_vecs = z_mean + torch.exp(z_log_var / 2) * epsilon
        return z_vecs, kl_loss

    def assm_scores(self, jtmpn_holder, x_mol_vecs, x_tree_mess):  # jtnn_vae.py:105-114
        jtmpn_holder, batch_idx = jtmpn_holder
        fatoms, fbonds, agraph, bgraph, scope = jtmpn_holder
        dev = x_mol_vecs.device
        cand_vecs = self.jtmpn(fatoms, fbonds, agraph, bgraph, scope, x_tree_mess)
        nb = x_mol_vecs.shape[0]

        def owners():
            bi = graphprep._to_np(batch_idx).astype("int64")
            if bi.size and (bi.min() < 0 or bi.max() >= nb):
                raise IndexError("batch_idx outside [0, %d)" % nb)
            if bi.size > 1 and (bi[1:] < bi[:-1]).any():
                return None  # candidates of a molecule not consecutive: the layout the kernel needs is absent
            import numpy as np
            ptr = np.zeros(nb + 1, dtype=np.int64)
            np.cumsum(np.bincount(bi, minlength=nb), out=ptr[1:])
            with graphprep.packed_uploads(dev):
                return graphprep._dev_i32(bi, dev), graphprep._dev_i32(ptr, dev)

        own = graphprep.cached_prep(batch_idx, ("assm_owner", str(dev), nb), owners) \
            if isinstance(batch_idx, torch.Tensor) else owners()
        if own is None:
            q = self.A_assm(x_mol_vecs.index_select(0, batch_idx.to(dev)))
            return (q * cand_vecs).sum(dim=1)
        # the bilinear form once per molecule, looked up per candidate inside the kernel
        return _RowDot.apply(self.A_assm(x_mol_vecs), cand_vecs, own[0], own[1])

    def assm(self, mol_batch, jtmpn_holder, x_mol_vecs, x_tree_mess, sync=True):  # jtnn_vae.py:105-138
        """``sync=False``: the accuracy comes back as a 0-d device tensor, not a float (no host wait here)."""
        scores = self.assm_scores(jtmpn_holder, x_mol_vecs, x_tree_mess)
        dev = scores.device

        def segments():
            seg_len, label_pos, tot = [], [], 0
            for mol_tree in mol_batch:
                for node in mol_tree.nodes:
                    if len(node.cands) > 1 and not node.is_leaf:
                        label_pos.append(tot + node.cands.index(node.label))
                        seg_len.append(len(node.cands))
                        tot += len(node.cands)
            import numpy as np
            ptr = np.zeros(len(seg_len) + 1, dtype=np.int64)
            np.cumsum(seg_len, out=ptr[1:])
            with graphprep.packed_uploads(dev):
                t = (graphprep._dev_i32(ptr, dev), graphprep._dev_i32(np.asarray(label_pos + [0], dtype=np.int64), dev))
            return t[0], t[1][:len(seg_len)], len(seg_len), tot

        key = jtmpn_holder[1]  # one batch_idx tensor per batch: its segments do not change between steps
        seg_ptr, label_pos, cnt, total = graphprep.cached_prep(key, ("assm_segments", str(dev)), segments) \
            if isinstance(key, torch.Tensor) else segments()
        if total != scores.shape[0]:  # host numbers: no device round trip
            raise ValueError("candidate segments cover %d scores, the holder has %d" % (total, scores.shape[0]))
        loss, acc = segment_cross_entropy(scores, None, label_pos, seg_ptr=seg_ptr)
        if cnt == 0:
            raise ZeroDivisionError("no tree node of the batch has more than one candidate")  # jtnn_vae.py:138
        acc = acc.float() / cnt
        return loss / len(mol_batch), (acc.item() if sync else acc)

    def _decode_and_assemble(self, x_batch, x_jtenc_holder, x_jtmpn_holder, z_tree_vecs, z_mol_vecs, x_tree_mess):
        """Decoder and assembly terms with their accuracies as 0-d device tensors: nothing here waits for the GPU."""
        if isinstance(getattr(self, "decoder", None), JTNNDecoder):
            word_loss, topo_loss, word_acc, topo_acc = self.decoder(x_batch, z_tree_vecs, x_jtenc_holder, sync=False)
        elif hasattr(self, "decoder"):
            word_loss, topo_loss, word_acc, topo_acc = self.decoder(x_batch, z_tree_vecs)
        else:
            word_loss = topo_loss = z_tree_vecs.sum() * 0.0
            word_acc = topo_acc = 0.0
        assm_loss, assm_acc = self.assm(x_batch, x_jtmpn_holder, z_mol_vecs, x_tree_mess, sync=False)
        return word_loss, topo_loss, assm_loss, [word_acc, topo_acc, assm_acc]

    @staticmethod
    def _fetch(scalars):
        """Python floats of a list of 0-d device tensors / numbers, in ONE device-to-host copy."""
        dev = [v for v in scalars if isinstance(v, torch.Tensor)]
        got = iter(torch.stack([v.detach().float() for v in dev]).tolist()) if dev else iter(())
        return [next(got) if isinstance(v, torch.Tensor) else float(v) for v in scalars]

    def forward(self, x_batch, beta, epsilons=None):  # jtnn_vae.py:85-103
        x_batch, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder = x_batch
        x_tree_vecs, x_tree_mess, x_mol_vecs = self.encode(x_jtenc_holder, x_mpn_holder)
        e_t, e_g = epsilons if epsilons is not None else (None, None)
        z_tree_vecs, tree_kl = self.rsample(x_tree_vecs, self.T_mean, self.T_var, e_t)
        z_mol_vecs, mol_kl = self.rsample(x_mol_vecs, self.G_mean, self.G_var, e_g)
        kl_div = tree_kl + mol_kl
        word_loss, topo_loss, assm_loss, accs = self._decode_and_assemble(x_batch, x_jtenc_holder, x_jtmpn_holder,
                                                                          z_tree_vecs, z_mol_vecs, x_tree_mess)
        # the reference reads kl_div.item() and three accuracies one after the other (jtnn_vae.py:97-103 and the
        # .item() calls inside decoder and assm); here they cross to the host together
        kl, word_acc, topo_acc, assm_acc = self._fetch([kl_div] + accs)
        return word_loss + topo_loss + assm_loss + beta * kl_div, kl, word_acc, topo_acc, assm_acc


class JTpropVAE(JTNNVAE):
    """Conditional variant (fast_jtnn/jtprop_vae.py:27-96, 361-441): property regressor on the sampled
    latent, optional denticity (binary) and isomer (3-class) heads selected by ``train_mode``."""

    def __init__(self, vocab, hidden_size, latent_size, depthT, depthG, train_mode=(), decoder=None):
        super(JTpropVAE, self).__init__(vocab, hidden_size, latent_size, depthT, depthG, decoder=decoder)
        self.propNN = nn.Sequential(nn.Linear(self.latent_size * 2, self.hidden_size), nn.Tanh(),
                                    nn.Linear(self.hidden_size, 2))
        self.train_mode = list(train_mode)
        for term in self.train_mode:
            if term == "denticity":
                self.denticityNN = nn.Sequential(nn.Linear(self.latent_size * 2, self.hidden_size), nn.Sigmoid(),
                                                 nn.Linear(self.hidden_size, 1))
                self.denticityNN_loss = nn.BCEWithLogitsLoss()
            elif term == "isomer":
                self.isomerNN = nn.Sequential(nn.Linear(self.latent_size * 2, self.hidden_size), nn.Sigmoid(),
                                              nn.Linear(self.hidden_size, 3))
                self.isomerNN_loss = nn.CrossEntropyLoss()
            else:
                raise ValueError("train_mode is not a valid value")
        self.prop_loss = nn.MSELoss()

    def forward(self, x_batch, beta, epsilons=None):  # jtprop_vae.py:361-441
        x_batch, x_prop_holder, x_jtenc_holder, x_mpn_holder, x_jtmpn_holder = x_batch
        length_p = len(x_prop_holder[0])
        if length_p not in (2, 3, 4):
            raise ValueError("The correct number of properties could not be unpacked")
        x_tree_vecs, x_tree_mess, x_mol_vecs = self.encode(x_jtenc_holder, x_mpn_holder)
        dev = x_tree_vecs.device
        props = torch.as_tensor(x_prop_holder).to(dev, non_blocking=True)
        e_t, e_g = epsilons if epsilons is not None else (None, None)
        z_tree_vecs, tree_kl = self.rsample(x_tree_vecs, self.T_mean, self.T_var, e_t)
        z_mol_vecs, mol_kl = self.rsample(x_mol_vecs, self.G_mean, self.G_var, e_g)
        kl_div = tree_kl + mol_kl
        word_loss, topo_loss, assm_loss, accs = self._decode_and_assemble(x_batch, x_jtenc_holder, x_jtmpn_holder,
                                                                          z_tree_vecs, z_mol_vecs, x_tree_mess)
        all_vec = torch.cat([z_tree_vecs, z_mol_vecs], dim=1)
        prop_loss = self.prop_loss(self.propNN(all_vec).squeeze(), props[:, 0:2].float())
        loss = word_loss + topo_loss + assm_loss + beta * kl_div + prop_loss
        extra = {}
        for term in self.train_mode:
            if term == "denticity":  # third column (jtprop_vae.py:373-378, 400-409)
                extra["dent_loss"] = self.denticityNN_loss(self.denticityNN(all_vec).squeeze(), props[:, 2].float())
            elif term == "isomer":  # fourth column (jtprop_vae.py:376-380, 410-419)
                extra["isomer_loss"] = self.isomerNN_loss(self.isomerNN(all_vec).squeeze(), props[:, 3].long())
        for v in extra.values():
            loss = loss + v
        # every logged scalar in one device-to-host copy
        vals = self._fetch([kl_div, prop_loss] + list(extra.values()) + accs)
        n_x = len(extra)
        word_acc, topo_acc, assm_acc = vals[2 + n_x:]
        log_metrics = {"kl_div": vals[0], "word_acc": word_acc * 100, "topo_acc": topo_acc * 100,
                       "assm_acc": assm_acc * 100, "prop_loss": vals[1] * 100}
        for name, v in zip(extra.keys(), vals[2:2 + n_x]):
            log_metrics[name] = v * 100
        return loss, log_metrics
