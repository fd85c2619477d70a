"""GPU: holders prepared from flat storage (no tables) drive the modules to the same bits as the reference-layout
tables of the same batch, forward and backward; the whole training step too."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(V, cls="vae", seed=3):
    from fast_jtnn_hip import JTNNVAE, JTpropVAE, synthetic

    if cls == "prop":
        m = JTpropVAE(synthetic.FakeVocab(V), 450, 56, 20, 3, ["denticity", "isomer"])
    else:
        m = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3)
    synthetic.fill_params(list(m.named_parameters()), seed=seed, bias_scale=0.05)
    return m.cuda()


def _grads(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def test_encoder_on_flat_holders_is_bitwise_the_table_route():
    from fast_jtnn_hip import flat, flatprep, synthetic
    from fast_jtnn_hip import tensorize as T

    V = 131
    trees = synthetic.make_batch(77, 96, vocab_size=V, mean_atoms=30.0)
    shard = flat.shard_from_trees(trees)
    m = _models(V)
    jt, _ = T.jtenc_tensorize(trees)
    mp = T.mpn_tensorize([t.mol for t in trees])
    res = []
    for holders in ((jt, mp), flatprep.encoder_batch(m, shard, None, train=True, all_messages=True)):
        m.zero_grad(set_to_none=True)
        tv, ms, mv = m.encode(*holders)
        (tv.sum() + (ms * ms).sum() + mv.sum()).backward()
        res.append((tv.detach().clone(), ms.detach().clone(), mv.detach().clone(), _grads(m)))
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    assert res[0][3].keys() == res[1][3].keys() and len(res[0][3]) > 10
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k
    # inference holders (no transposes, no backward plan) and a selection of the shard
    ids = np.asarray([5, 80, 3, 41, 41, 0])
    sub = [copy.deepcopy(trees[i]) for i in ids]  # one copy per position: 41 occurs twice
    synthetic.set_batch_nodeID(sub)
    with torch.no_grad():
        want = m.encode_vectors(T.jtenc_tensorize(sub)[0], T.mpn_tensorize([t.mol for t in sub]))
        got = m.encode_vectors(*flatprep.encoder_batch(m, shard, ids, train=False))
    assert torch.equal(want, got)
    with pytest.raises(Exception):  # prepared for inference: no transposed tables to run a backward pass on
        tv, ms, mv = m.encode(*flatprep.encoder_batch(m, shard, ids, train=False))
        (tv.sum() + mv.sum()).backward()


@pytest.mark.parametrize("cls", ["vae", "prop"])
def test_training_step_on_flat_batch_is_bitwise_the_table_route(cls):
    from fast_jtnn_hip import flat, flatprep, synthetic
    from fast_jtnn_hip import tensorize as T

    V, B = 89, 48
    trees = synthetic.make_batch(91, B, vocab_size=V, mean_atoms=24.0, with_cands=True)
    shard = flat.shard_from_trees(trees, with_cands=True)
    m = _models(V, cls)
    g = torch.Generator().manual_seed(5)
    eps = (torch.randn((B, 28), generator=g).cuda(), torch.randn((B, 28), generator=g).cuda())
    rs = np.random.RandomState(1)
    props = np.stack([rs.standard_normal(B), rs.standard_normal(B), rs.randint(0, 2, B), rs.randint(0, 3, B)],
                     axis=1).astype(np.float32)
    _, jt, mp, cj = T.tensorize(trees, assm=True)
    if cls == "prop":
        table_batch = (trees, torch.from_numpy(props), jt, mp, cj)
        flat_batch = flatprep.training_batch(m, shard, None, props=torch.from_numpy(props))
    else:
        table_batch = (trees, jt, mp, cj)
        flat_batch = flatprep.training_batch(m, shard, None)
    out = []
    torch.use_deterministic_algorithms(True, warn_only=True)  # torch's index_add (decoder heads) without atomics
    try:
        for xb in (table_batch, table_batch, flat_batch):
            m.zero_grad(set_to_none=True)
            r = m(xb, 0.3, eps)
            r[0].backward()
            out.append((r, _grads(m)))
    finally:
        torch.use_deterministic_algorithms(False)
    (ra, ga), (ra2, ga2), (rb, gb) = out
    assert torch.equal(ra[0], rb[0])
    assert ra[1:] == rb[1:]  # kl / accuracies (or the metrics dict)
    assert ga.keys() == gb.keys()
    for k in ga:
        if torch.equal(ga[k], ga2[k]):  # the table route repeats itself to the bit: so must the flat route
            assert torch.equal(ga[k], gb[k]), k
        else:  # a torch op upstream of this gradient sums in run-dependent order
            d = float((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-30))
            assert d < 1e-5, (k, d)


def test_flat_batches_through_the_prefetcher_and_bulk_encoding(tmp_path):
    from fast_jtnn_hip import bulk, flat, flatprep, synthetic
    from fast_jtnn_hip.prefetch import Prefetcher

    V = 73
    trees = synthetic.make_batch(13, 200, vocab_size=V, mean_atoms=20.0)
    shard = flat.shard_from_trees(trees)
    m = _models(V)
    ids = [np.arange(i, min(i + 64, 200)) for i in range(0, 200, 64)]
    with torch.no_grad():
        want = [m.encode_vectors(*flatprep.encoder_batch(m, shard, sel, train=False)).clone() for sel in ids]
        with Prefetcher(ids, lambda sel: flatprep.encoder_batch(m, shard, sel, train=False), "cuda:0", depth=2) as pf:
            got = [m.encode_vectors(*h).clone() for h in pf]
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(want, got))
    path = bulk.encode_bulk(m, 200, None, 64, str(tmp_path / "lat"), shard=shard)
    rows = np.load(path)
    assert np.array_equal(rows, torch.cat(want).cpu().numpy())
