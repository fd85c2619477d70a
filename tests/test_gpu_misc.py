"""GPU: paths that no benchmark configuration drives - the assembly score when a molecule's candidates are not
consecutive, and the packed-weight cache when weights change in ways autograd does not count."""
import numpy as np
import pytest
import torch

from common import rel_err

pytestmark = pytest.mark.gpu


def test_assembly_scores_with_shuffled_candidates():
    """batch_idx of the reference is non-decreasing (fast_jtnn/datautils.py:150-161) and the row-dot kernel
    relies on it; any other order takes the index_select route.  Both must give the scores of the reference's
    formula (fast_jtnn/jtnn_vae.py:112-114), forward and backward."""
    from fast_jtnn_hip import JTNNVAE, synthetic
    from fast_jtnn_hip import tensorize as T

    V, B = 47, 10
    trees = synthetic.make_batch(29, B, vocab_size=V, mean_atoms=18.0, with_cands=True)
    jt, mess_dict = T.jtenc_tensorize(trees)
    mp = T.mpn_tensorize([t.mol for t in trees])
    cands, bi = synthetic.collect_candidates(trees)
    perm = np.random.RandomState(0).permutation(len(cands))
    m = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3, decoder=False)
    synthetic.fill_params(list(m.named_parameters()), 4, 0.05)
    m = m.cuda()
    proj = torch.from_numpy(np.random.RandomState(1).standard_normal(len(cands)).astype(np.float32)).cuda()

    def run(order):
        m.zero_grad(set_to_none=True)
        cj = T.jtmpn_tensorize([cands[i] for i in order], mess_dict)
        _, mess, mol_vecs = m.encode(jt, mp)
        scores = m.assm_scores((cj, torch.from_numpy(bi[order])), m.G_mean(mol_vecs), mess)
        (scores * proj[torch.from_numpy(order).cuda()]).sum().backward()
        return scores.detach().cpu().numpy(), {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()
                                               if p.grad is not None}

    ident = np.arange(len(cands))
    s0, g0 = run(ident)
    s1, g1 = run(perm)
    assert (bi[perm][1:] < bi[perm][:-1]).any()  # really out of order: the kernel route is not taken
    assert rel_err(s1, s0[perm]) < 1e-5
    assert g0.keys() == g1.keys() and "A_assm.weight" in g0 and "jtnn.GRU.W_z.weight" in g0
    for k in g0:
        assert np.linalg.norm(g0[k] - g1[k]) <= 1e-4 * max(np.linalg.norm(g0[k]), 1e-30), k


def test_packed_weights_follow_the_parameters():
    """The packed GEMM operands are cached per weight: an in-place update of the parameter, a device round trip
    and load_state_dict are seen; a write through .data needs invalidate_packed_weights()."""
    from fast_jtnn_hip import MPN, synthetic
    from fast_jtnn_hip import tensorize as T
    from fast_jtnn_hip.mpn import invalidate_packed_weights

    trees = synthetic.make_batch(3, 6, vocab_size=11, mean_atoms=14.0)
    holder = T.mpn_tensorize([t.mol for t in trees])
    m = MPN(450, 3)
    synthetic.fill_params(list(m.named_parameters()), 1, 0.05)
    m = m.cuda()
    with torch.no_grad():
        base = m(*holder).clone()
        m.W_h.weight.mul_(1.5)  # counted by autograd's version counter
        changed = m(*holder).clone()
        assert not torch.equal(base, changed)
        m.W_h.weight.data.div_(1.5)  # through .data: not counted, same storage
        invalidate_packed_weights()
        back = m(*holder).clone()
        assert rel_err(back.cpu().numpy(), base.cpu().numpy()) < 1e-6
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.W_o.weight.mul_(0.0)
        assert not torch.equal(m(*holder), back)
        m.load_state_dict(sd)  # copies in place on the parameters: seen
        assert torch.equal(m(*holder), back)
        m = m.cpu().cuda()  # new storage, same Parameter objects
        assert torch.equal(m(*holder), back)


def test_onehot_feature_layers_agree_with_the_dense_products():
    """mpn.ONEHOT routes W_i / the atom columns of W_o through the one-hot row kernels: same results as the
    dense products up to fp32 summation order, forward and backward, for MPN and JTMPN."""
    import fast_jtnn_hip.mpn as mpn_mod
    from fast_jtnn_hip import JTNNVAE, synthetic
    from fast_jtnn_hip import tensorize as T

    V, B = 31, 12
    trees = synthetic.make_batch(5, B, vocab_size=V, mean_atoms=22.0, with_cands=True)
    m = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3, decoder=False)
    synthetic.fill_params(list(m.named_parameters()), 6, 0.05)
    m = m.cuda()
    out = []
    old = mpn_mod.ONEHOT
    try:
        for flag in (False, True):
            mpn_mod.ONEHOT = flag
            _, jt, mp, (cj, bi) = T.tensorize(trees, assm=True)  # fresh holders: nothing cached from the other route
            m.zero_grad(set_to_none=True)
            tv, ms, mv = m.encode(jt, mp)
            cv = m.jtmpn(*cj, ms)
            (mv.sum() + (cv * cv).sum()).backward()
            out.append((mv.detach().cpu().numpy(), cv.detach().cpu().numpy(),
                        {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        mpn_mod.ONEHOT = old
    (mv0, cv0, g0), (mv1, cv1, g1) = out
    assert rel_err(mv1, mv0) < 1e-5 and rel_err(cv1, cv0) < 1e-5
    for k in ("mpn.W_i.weight", "mpn.W_o.weight", "jtmpn.W_i.weight", "jtmpn.W_o.weight", "mpn.W_h.weight"):
        assert np.linalg.norm(g1[k] - g0[k]) <= 2e-5 * np.linalg.norm(g0[k]), k
