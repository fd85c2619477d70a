"""Teacher-forced tree decoder: ``JTNNDecoder.forward`` of the reference
(fast_jtnn/jtnn_dec.py:55-193, GRU: fast_jtnn/nnutils.py:56-71) with the same
constructor, parameter names (``embedding``, ``W_z``, ``U_r``, ``W_r``, ``W_h``,
``W``, ``U``, ``U_i``, ``W_o``, ``U_o``) and return value
``(word_loss, topo_loss, word_acc, topo_acc)``.

The reference walks the depth-first traces of the batch one time step at a time
on the host: per step it looks hidden rows up in a Python dictionary, stacks them
padded to 15 neighbours, runs the GRU and stores the new rows back.  Restated for
the GPU, the time loop disappears:

* along a trace every directed tree edge (x -> y) is created exactly once, from
  the messages that reached x earlier, so the hidden rows of a whole batch are the
  messages of ONE dependency graph without cycles.  The host builds that graph
  once (``decoder_plan``), and the tree-GRU kernels compute each message once,
  level by level (the longest dependency chain is the tree depth, not the trace
  length), forward and backward;
* the stop-prediction inputs (sum over all messages that have reached x so far)
  are one neighbour-sum kernel over all steps of all molecules;
* word and stop heads run as a few batch-wide dense products instead of being
  assembled step by step.

Only ``forward`` (training) is provided: ``decode`` is a chemistry-driven host
loop that needs RDKit and stays with the reference.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import graphprep
from .jtnn_enc import _MessageGRUFunction
from .linear import linear, table_linear
from .nnutils import csr_gather_sum

MAX_NB = 15  # fast_jtnn/jtnn_dec.py:9


def dfs_trace(root):
    """[(x, y, direction)] in the order of the reference's recursive ``dfs`` (jtnn_dec.py:299-305):
    (x, y, 1) on the way down to child y, (y, x, 0) on the way back; iterative, so deep trees are fine."""
    out = []
    stack = [(root, -1, iter(root.neighbors))]
    while stack:
        x, fa_idx, it = stack[-1]
        for y in it:
            if y.idx == fa_idx:
                continue
            out.append((x, y, 1))
            stack.append((y, x.idx, iter(y.neighbors)))
            break
        else:
            stack.pop()
            if stack:
                out.append((x, stack[-1][0], 0))
    return out


@dataclass
class DecoderPlan:
    """Index form of the teacher-forced traces of one batch (host arrays)."""

    n_trees: int
    n_mess: int  # directed edges + the padding message 0
    mess_word: np.ndarray  # [n_mess] word of the node a message leaves from
    dep_rp: np.ndarray  # CSR: messages a message is computed from (neighbours so far minus the target)
    dep_col: np.ndarray
    stop_rp: np.ndarray  # CSR per stop row: all messages that have reached the node so far
    stop_col: np.ndarray
    stop_word: np.ndarray  # [n_stop] word of the current node
    stop_ctx: np.ndarray  # [n_stop] molecule index
    stop_target: np.ndarray  # [n_stop] 1 = go down, 0 = go back (and the final stop at every root)
    pred_mess: np.ndarray  # [n_pred - n_trees] message whose hidden row predicts the child's word
    pred_ctx: np.ndarray  # [n_pred]
    pred_target: np.ndarray  # [n_pred] word ids; the first n_trees rows predict the roots from a zero row


def decoder_plan(mol_batch, mess_dict=None) -> DecoderPlan:
    """Follows fast_jtnn/jtnn_dec.py:59-167: rows are ordered time step by time step, molecules in batch
    order inside a step (the order in which the reference concatenates them), the final stop at every
    root last.  ``node.neighbors`` is left as it is (the reference clears and rewires it).

    Plain-Python statement of the plan, from the tree objects; messages are numbered in trace order, or
    with ``mess_dict`` ((x.idx, y.idx) -> id, as JTNNEncoder.tensorize returns it) in the encoder's
    numbering, which is what ``decoder_plan_from_holder`` produces natively from the tables alone."""
    B = len(mol_batch)
    traces = [dfs_trace(t.nodes[0]) for t in mol_batch]
    n_mess = 1 + (len(mess_dict) if mess_dict is not None else sum(len(tr) for tr in traces))
    mess_word = [0] * n_mess
    dep_lists = [[] for _ in range(n_mess)]
    rows = []  # (t, i, message id, stop list, x word, direction, y word)
    final_lists = []
    made = 0
    for i, trace in enumerate(traces):
        reached = {}  # node idx -> [(message id, sender idx)] in creation order
        for t, (x, y, d) in enumerate(trace):
            made += 1
            mid = made if mess_dict is None else mess_dict[(x.idx, y.idx)]
            inc = reached.get(x.idx, ())
            mess_word[mid] = x.wid
            dep_lists[mid] = [m for m, w in inc if w != y.idx]
            rows.append((t, i, mid, [m for m, _ in inc], x.wid, d, y.wid))
            reached.setdefault(y.idx, []).append((mid, x.idx))
        final_lists.append([m for m, _ in reached.get(mol_batch[i].nodes[0].idx, ())])
    rows.sort(key=lambda r: (r[0], r[1]))

    def csr(lists):
        rp = np.zeros(len(lists) + 1, dtype=np.int32)
        np.cumsum([len(l) for l in lists], out=rp[1:])
        col = np.fromiter((m for l in lists for m in l), dtype=np.int32, count=int(rp[-1]))
        return rp, col

    for lst in dep_lists:
        if len(lst) > MAX_NB:
            raise IndexError("a tree node has more than %d neighbours" % MAX_NB)
    dep_rp, dep_col = csr(dep_lists)
    stop_rp, stop_col = csr([r[3] for r in rows] + final_lists)
    roots = [t.nodes[0] for t in mol_batch]
    down = [r for r in rows if r[5] == 1]
    return DecoderPlan(
        n_trees=B, n_mess=len(mess_word), mess_word=np.asarray(mess_word, dtype=np.int64),
        dep_rp=dep_rp, dep_col=dep_col, stop_rp=stop_rp, stop_col=stop_col,
        stop_word=np.asarray([r[4] for r in rows] + [n.wid for n in roots], dtype=np.int64),
        stop_ctx=np.asarray([r[1] for r in rows] + list(range(B)), dtype=np.int64),
        stop_target=np.asarray([r[5] for r in rows] + [0] * B, dtype=np.float32),
        pred_mess=np.asarray([r[2] for r in down], dtype=np.int64),
        pred_ctx=np.asarray(list(range(B)) + [r[1] for r in down], dtype=np.int64),
        pred_target=np.asarray([n.wid for n in roots] + [r[6] for r in down], dtype=np.int64))


def decoder_plan_from_holder(jtenc_holder) -> DecoderPlan:
    """The same plan from the junction-tree holder (fnode, fmess, node_graph, mess_graph, scope) that
    ``JTNNEncoder.tensorize`` built for the batch: the tables hold every tree's nodes, neighbour order and
    root, so no tree object is touched and the walk runs in the native library."""
    fnode, fmess, node_graph, _, scope = jtenc_holder
    wid = graphprep._to_np(fnode).astype(np.int32)
    src = graphprep._to_np(fmess).astype(np.int32)
    ng = graphprep._to_np(node_graph)
    n_mess = src.shape[0]
    dst = np.zeros(n_mess, dtype=np.int32)
    nz = ng != 0
    dst[ng[nz]] = np.repeat(np.arange(ng.shape[0], dtype=np.int32), nz.sum(axis=1))  # message m arrives at node y
    roots = np.asarray([st for st, _ in scope], dtype=np.int32)
    return decoder_plan_from_arrays(wid, src, dst, roots)


def decoder_plan_from_arrays(wid, src, dst, roots) -> DecoderPlan:
    """The plan from the batch's message arrays: word of every node, source and target node of every directed
    tree edge (entry 0 = padding message), first node of every tree."""
    from . import _capi as K

    wid, src = np.ascontiguousarray(wid, dtype=np.int32), np.ascontiguousarray(src, dtype=np.int32)
    n_mess = src.shape[0]
    a = K.prep_decoder_plan(src, dst, wid, roots)
    mess_word = wid[src].astype(np.int64)
    return DecoderPlan(n_trees=len(roots), n_mess=n_mess, mess_word=mess_word, dep_rp=a["dep_rp"],
                       dep_col=a["dep_col"], stop_rp=a["stop_rp"], stop_col=a["stop_col"],
                       stop_word=a["stop_word"].astype(np.int64), stop_ctx=a["stop_ctx"].astype(np.int64),
                       stop_target=a["stop_target"], pred_mess=a["pred_mess"].astype(np.int64),
                       pred_ctx=a["pred_ctx"].astype(np.int64), pred_target=a["pred_target"].astype(np.int64))


class JTNNDecoder(nn.Module):
    def __init__(self, vocab, hidden_size, latent_size, embedding):
        super(JTNNDecoder, self).__init__()
        self.hidden_size = hidden_size
        self.vocab_size = vocab.size()
        self.vocab = vocab
        self.embedding = embedding
        # GRU weights (bias on W_r, none on U_r: jtnn_dec.py:22-25)
        self.W_z = nn.Linear(2 * hidden_size, hidden_size)
        self.U_r = nn.Linear(hidden_size, hidden_size, bias=False)
        self.W_r = nn.Linear(hidden_size, hidden_size)
        self.W_h = nn.Linear(2 * hidden_size, hidden_size)
        self.W = nn.Linear(hidden_size + latent_size, hidden_size)  # word prediction
        self.U = nn.Linear(hidden_size + latent_size, hidden_size)  # stop prediction
        self.U_i = nn.Linear(2 * hidden_size, hidden_size)
        self.W_o = nn.Linear(hidden_size, self.vocab_size)
        self.U_o = nn.Linear(hidden_size, 1)

    def aggregate(self, hiddens, contexts, x_tree_vecs, mode, contexts32=None):  # jtnn_dec.py:42-53
        """``contexts32``: ``contexts`` as int32 on the device, if the caller keeps it."""
        if mode == "word":
            V, V_o = self.W, self.W_o
        elif mode == "stop":
            V, V_o = self.U, self.U_o
        else:
            raise ValueError("aggregate mode is wrong")
        if contexts32 is None:
            contexts32 = contexts.to(device=hiddens.device, dtype=torch.int32)
        # V(cat([hiddens, x_tree_vecs[contexts]])) without the concatenation: the tree vectors go through their 28
        # columns of V once per tree and are added per row in the epilogue of the product over `hiddens`
        hid = table_linear(hiddens, x_tree_vecs, contexts32, V, h_first=True, relu=True)
        return linear(hid, V_o) if mode == "word" else V_o(hid)  # U_o has one output column: a matrix-vector product

    def messages(self, prep):
        """Hidden rows of all directed edges of the batch's traces, [n_mess, H] (row 0 = padding)."""
        return _MessageGRUFunction.apply(self.embedding.weight, self.W_z.weight, self.W_z.bias, self.W_r.weight,
                                         self.W_r.bias, self.U_r.weight, None, self.W_h.weight, self.W_h.bias,
                                         prep, torch.is_grad_enabled())

    def forward(self, mol_batch, x_tree_vecs, jtenc_holder=None, sync=True):
        """``jtenc_holder``: the junction-tree holder of the same batch, if the caller has it (JTNNVAE.forward
        does): the trace tables then come from it natively and are kept with it; without it they are built
        from the tree objects.  ``sync=False`` returns the two accuracies as 0-d device tensors instead of
        Python floats (the caller fetches them together with its other scalars in one copy)."""
        dev = self.W_z.weight.device
        if dev.type != "cuda":
            raise RuntimeError("fast_jtnn_hip.JTNNDecoder runs on an MI355X only (module is on %s); there is no "
                               "CPU path" % dev)
        if jtenc_holder is None:
            plan = decoder_plan(mol_batch)
        else:
            key = jtenc_holder[3]
            build = lambda: decoder_plan_from_holder(jtenc_holder)  # noqa: E731
            plan = graphprep.cached_prep(key, ("decoder_plan",), build) if isinstance(key, torch.Tensor) else build()
            if plan.n_trees != len(mol_batch):
                raise ValueError("jtenc_holder describes %d trees, the batch has %d" % (plan.n_trees, len(mol_batch)))
        B, H = plan.n_trees, self.hidden_size
        with torch.cuda.device(dev):
            cache = plan.__dict__.setdefault("_device", {})
            if str(dev) not in cache:  # a plan kept with its holder keeps its device tables too
                with graphprep.packed_uploads(dev):
                    # device views are filled when the block ends (one packed copy): nothing may read them inside it
                    idx32 = [graphprep._dev_i32(a, dev) for a in (plan.pred_mess, plan.pred_ctx, plan.pred_target,
                                                                  plan.stop_word, plan.stop_ctx)]
                    stop_csr = graphprep.upload_csr(plan.stop_rp, plan.stop_col, dev)
                    st_rp, st_col = graphprep.transpose_csr(plan.stop_rp, plan.stop_col, plan.n_mess)
                    stop_csr_t = graphprep.upload_csr(st_rp, st_col, dev)
                prep = None
                if plan.n_mess > 1:
                    prep = graphprep.prepare_message_dag(plan.mess_word, plan.dep_rp, plan.dep_col, self.vocab_size,
                                                         dev)
                cache[str(dev)] = ([t.long() for t in idx32], stop_csr, stop_csr_t,
                                   torch.from_numpy(np.ascontiguousarray(plan.stop_target)).to(dev), prep,
                                   (idx32[1], idx32[3], idx32[4]))
            (pred_mess, pred_ctx, pred_target, stop_word, stop_ctx), stop_csr, stop_csr_t, stop_target, prep, \
                (pred_ctx32, stop_word32, stop_ctx32) = cache[str(dev)]
            if prep is not None:
                mess = self.messages(prep)
            else:  # single-node trees only: nothing is ever passed
                mess = x_tree_vecs.new_zeros((1, H))

            # word prediction: the root from a zero row, every child from the message that reaches it
            pred_hiddens = torch.cat([x_tree_vecs.new_zeros((B, H)), mess.index_select(0, pred_mess)], dim=0)
            pred_scores = self.aggregate(pred_hiddens, pred_ctx, x_tree_vecs, "word", pred_ctx32)
            pred_loss = F.cross_entropy(pred_scores, pred_target, reduction="sum") / B
            pred_acc = (pred_scores.argmax(dim=1) == pred_target).float().sum() / pred_target.nelement()

            # stop prediction: clique embedding and the sum of everything that has reached the node
            cur_o = csr_gather_sum(mess, stop_csr, stop_csr_t)
            # U_i(cat([embedding[stop_word], cur_o])): the embedding half as a per-word table       (jtnn_dec.py:160-163)
            stop_hiddens = table_linear(cur_o, self.embedding.weight, stop_word32, self.U_i, h_first=False, relu=True)
            stop_scores = self.aggregate(stop_hiddens, stop_ctx, x_tree_vecs, "stop", stop_ctx32).squeeze(-1)
            stop_loss = F.binary_cross_entropy_with_logits(stop_scores, stop_target, reduction="sum") / B
            stop_acc = ((stop_scores >= 0).float() == stop_target).float().sum() / stop_target.nelement()
            if not sync:
                return pred_loss, stop_loss, pred_acc, stop_acc
            accs = torch.stack([pred_acc, stop_acc]).tolist()  # one host sync for both
        return pred_loss, stop_loss, accs[0], accs[1]
