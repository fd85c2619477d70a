"""Shared helpers of the test-suite: golden fixtures, oracle models, tolerances."""
import glob
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")

# north_star tolerance: fp32 latent means / log-vars and losses within 1e-4 relative.
# "relative" is taken against the largest magnitude of the compared tensor, so exact zeros
# (ReLU outputs, the padding message) do not blow the ratio up.
RTOL = 1e-4


def rel_err(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max() / scale)


def fro_err(a, b) -> float:
    """||a - b||_F / ||b||_F: insensitive to the handful of ReLU units whose pre-activation
    is within rounding of zero and therefore switch differently on two platforms."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "case*.npz")))


class Golden:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.z = z
        self.name = os.path.basename(path)
        (self.seed, self.B, self.H, self.L, self.dT, self.dG, self.V) = (int(v) for v in z["meta"])
        self.bias_scale = float(z["bias_scale"])
        self.mean_atoms = float(z["mean_atoms"])
        self.sha = str(z["weights_sha256"])
        self.tree_shape = str(z["tree_shape"]) if "tree_shape" in z.files else "random"
        self.cand_hub = bool(z["cand_hub"]) if "cand_hub" in z.files else False

    def scope(self, key):
        return [(int(a), int(b)) for a, b in self.z[key]]

    def jtenc_holder(self):
        z = self.z
        return (torch.from_numpy(z["fnode"].astype(np.int64)), torch.from_numpy(z["fmess"].astype(np.int64)),
                torch.from_numpy(z["node_graph"].astype(np.int64)),
                torch.from_numpy(z["mess_graph"].astype(np.int64)), self.scope("jt_scope"))

    def mpn_holder(self):
        z = self.z
        return (torch.from_numpy(z["fatoms"].astype(np.float32)), torch.from_numpy(z["fbonds"].astype(np.float32)),
                torch.from_numpy(z["agraph"].astype(np.int64)), torch.from_numpy(z["bgraph"].astype(np.int64)),
                self.scope("mpn_scope"))

    def jtmpn_holder(self):
        z = self.z
        return (torch.from_numpy(z["c_fatoms"].astype(np.float32)),
                torch.from_numpy(z["c_fbonds"].astype(np.float32)),
                torch.from_numpy(z["c_agraph"].astype(np.int64)), torch.from_numpy(z["c_bgraph"].astype(np.int64)),
                self.scope("c_scope"))

    def batch_idx(self):
        return torch.from_numpy(self.z["batch_idx"].astype(np.int64))

    def mess_dict(self):
        return {(int(a), int(b)): int(c) for a, b, c in self.z["mess_dict"]}

    def trees(self):
        from fast_jtnn_hip import synthetic

        return synthetic.make_batch(self.seed, self.B, vocab_size=self.V, mean_atoms=self.mean_atoms,
                                    with_cands=True, tree_shape=self.tree_shape, cand_hub=self.cand_hub)


def projection(shape, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal(tuple(shape)).astype(np.float32))


def golden_loss(model, g, tree_vecs, mol_vecs, cand_vecs):
    """The scalar of tests/golden/make_golden.py."""
    dev = tree_vecs.device
    mean = torch.cat([model.T_mean(tree_vecs), model.G_mean(mol_vecs)], dim=1)
    logvar = torch.cat([-torch.abs(model.T_var(tree_vecs)), -torch.abs(model.G_var(mol_vecs))], dim=1)
    return ((mean * projection(mean.shape, g.seed + 101).to(dev)).sum()
            + (logvar * projection(logvar.shape, g.seed + 102).to(dev)).sum()
            + (cand_vecs * projection(cand_vecs.shape, g.seed + 103).to(dev)).sum()
            / max(1, cand_vecs.shape[0]) * g.B)


def make_oracle(g):
    from fast_jtnn_hip import synthetic
    from oracle.ref_encoder import OracleEncoder

    m = OracleEncoder(g.V, g.H, g.L, g.dT, g.dG)
    params = [(n, p) for n, p in m.named_parameters()]
    synthetic.fill_params(params, seed=g.seed, bias_scale=g.bias_scale)
    return m


def make_hip_model(g, device):
    from fast_jtnn_hip import JTNNVAE, synthetic

    m = JTNNVAE(synthetic.FakeVocab(g.V), g.H, g.L, g.dT, g.dG)
    synthetic.fill_params(list(m.named_parameters()), seed=g.seed, bias_scale=g.bias_scale)
    return m.to(device)


def make_oracle_vae(g, prop_modes=None):
    """Oracle of the whole training-side model (decoder, assembly loss, property heads)."""
    from fast_jtnn_hip import synthetic
    from oracle.ref_vae import OracleJTNNVAE, OracleJTpropVAE

    if prop_modes is None:
        m = OracleJTNNVAE(g.V, g.H, g.L, g.dT, g.dG)
    else:
        m = OracleJTpropVAE(g.V, g.H, g.L, g.dT, g.dG, prop_modes)
    synthetic.fill_params(list(m.named_parameters()), seed=g.seed, bias_scale=g.bias_scale)
    return m


def golden_x_batch(g, device=None):
    """(trees, jtenc_holder, mpn_holder, (jtmpn_holder, batch_idx)) of a golden case, on fresh trees
    (the decoder rewires node.neighbors, so every forward call needs its own)."""
    return (g.trees(), g.jtenc_holder(), g.mpn_holder(), (g.jtmpn_holder(), g.batch_idx()))


def fp64_training_grads(g, prop_modes=None):
    """Gradients of the golden training step from the oracle's op sequence run in fp64 (same batch, same
    recorded noise): the yardstick that tells an implementation's rounding from the reference's own.
    The candidate-graph ReLUs sit behind 20 recurrent fp32 iterations; a unit whose pre-activation is
    within fp32 rounding of zero switches differently in fp32 and in exact arithmetic, which moves single
    gradient entries of the reference's own fp32 result by up to a few 1e-3."""
    z = g.z
    m = make_oracle_vae(g, prop_modes).double()
    trees, jt, mp, (cj, bi) = golden_x_batch(g)
    mp = (mp[0].double(), mp[1].double()) + tuple(mp[2:])
    cj = (cj[0].double(), cj[1].double()) + tuple(cj[2:])
    tag = "fwd" if prop_modes is None else "prop"
    eps = (torch.from_numpy(z[tag + "_eps_tree"]).double(), torch.from_numpy(z[tag + "_eps_mol"]).double())
    if prop_modes is None:
        loss = m((trees, jt, mp, (cj, bi)), float(z["fwd_beta"]), epsilons=eps)[0]
    else:
        loss = m((trees, torch.from_numpy(z["prop_table"]), jt, mp, (cj, bi)), float(z["fwd_beta"]), epsilons=eps)[0]
    loss.backward()
    return {n: p.grad.detach().numpy() for n, p in m.named_parameters() if p.grad is not None}


def check_grad_stats(z, tag, named_params, tol_norm, tol_head, exact=None):
    """Per-parameter gradient norm and the first 256 gradient values against the golden's.  With ``exact``
    (fp64 gradients of the same step): an entry may differ from the golden by as much as the golden itself
    differs from the fp64 result (the reference's own rounding), and must match the fp64 result to 1e-4."""
    for name, p in named_params:
        key = "%snorm/%s" % (tag, name)
        if key not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        ref = float(z[key])
        got = float(p.grad.double().norm())
        assert abs(got - ref) <= tol_norm * max(ref, 1e-6), (name, got, ref)
        head_ref = z["%shead/%s" % (tag, name)]
        if np.abs(head_ref).max() > 1e-8:
            head = p.grad.detach().reshape(-1)[:256].cpu().numpy()
            slack = 0.0
            if exact is not None:
                ex = exact[name].reshape(-1)[:256]
                slack = 1.5 * rel_err(ex, head_ref)
                assert rel_err(head, ex) < 1e-4, (name, rel_err(head, ex))
            assert rel_err(head, head_ref) < max(tol_head, slack), (name, rel_err(head, head_ref), slack)
