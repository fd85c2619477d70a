"""Bulk latent generation: shard files, text export (CPU) and the streamed encoder (GPU)."""
import numpy as np
import pytest
import torch

from fast_jtnn_hip import bulk


def test_shards_round_trip_and_text_export(tmp_path):
    rs = np.random.RandomState(0)
    full = rs.standard_normal((10, 7)).astype(np.float32)
    prefix = str(tmp_path / "latent")
    from fast_jtnn_hip.parallel import shard_bounds
    for r in range(3):
        lo, hi = shard_bounds(10, 3, r)
        np.save(bulk.shard_path(prefix, r, 3), full[lo:hi])
    assert np.array_equal(bulk.load_shards(prefix, 3), full)
    txt = str(tmp_path / "latent_features.txt")
    assert bulk.shards_to_text(prefix, 3, txt, rows_per_chunk=3) == 10
    ref = str(tmp_path / "ref.txt")
    np.savetxt(ref, full)  # what fast_bo/gen_latent.py:99 writes
    assert open(txt).read() == open(ref).read()
    assert np.array_equal(np.loadtxt(txt).astype(np.float32), full)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["vectors", "mean"])
def test_encode_bulk_equals_direct_encoding(tmp_path, what):
    from fast_jtnn_hip import JTNNVAE, synthetic, tensorize as T

    n, bs, world = 70, 16, 2  # ragged: last batch of each rank is short
    trees = synthetic.make_batch(9, n, vocab_size=50, mean_atoms=20.0)
    model = JTNNVAE(synthetic.FakeVocab(50), 450, 56, 20, 3)
    synthetic.fill_params(list(model.named_parameters()), seed=1, bias_scale=0.05)
    model = model.cuda()

    def make_holders(lo, hi):
        part = trees[lo:hi]
        synthetic.set_batch_nodeID(part)
        jt, _ = T.jtenc_tensorize(part)
        return jt, T.mpn_tensorize([t.mol for t in part])

    prefix = str(tmp_path / "z")
    for r in range(world):  # the ranks of a bulk job share nothing: run them one after the other
        bulk.encode_bulk(model, n, make_holders, bs, prefix, what=what, rank=r, world=world)
    got = bulk.load_shards(prefix, world)
    with torch.no_grad():
        jt, mp = make_holders(0, n)
        want = model.encode_vectors(jt, mp) if what == "vectors" else model.encode_latent(jt, mp)[0]
    want = want.cpu().numpy()
    assert got.shape == want.shape == (n, 900 if what == "vectors" else 56)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    assert model.training  # mode restored


@pytest.mark.gpu
def test_prefetched_holders_give_bitwise_the_same_step():
    """Batches prepared and uploaded by the background thread on its copy stream, one batch ahead of the
    stepping thread, against the same batches prepared inside the step: outputs and gradients bit for bit."""
    from fast_jtnn_hip import JTNNVAE, synthetic, tensorize as T
    from fast_jtnn_hip.prefetch import Prefetcher, warm_training_batch

    V = 60
    model = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3)
    synthetic.fill_params(list(model.named_parameters()), seed=2, bias_scale=0.05)
    model = model.cuda()

    def batches():
        for i in range(5):
            trees = synthetic.make_batch(40 + i, 48, vocab_size=V, mean_atoms=18.0, with_cands=True)
            jt, md = T.jtenc_tensorize(trees)
            cands, bi = synthetic.collect_candidates(trees)
            yield (trees, jt, T.mpn_tensorize([t.mol for t in trees]), (T.jtmpn_tensorize(cands, md), torch.from_numpy(bi)))

    def run(source):
        out = []
        g = torch.Generator(device="cpu").manual_seed(3)
        for xb in source:
            eps = [torch.randn((48, 28), generator=g).cuda() for _ in range(2)]
            model.zero_grad()
            loss = model(xb, 0.1, epsilons=eps)[0]
            loss.backward()
            out.append((loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
        return out

    direct = run(batches())
    with Prefetcher(batches(), lambda b: warm_training_batch(model, b), "cuda:0", depth=2) as pf:
        ahead = run(pf)
    assert len(direct) == len(ahead) == 5
    again = run(batches())
    for (la, ga), (lb, gb), (lc, gc) in zip(direct, ahead, again):
        assert torch.equal(la, lb)
        for name in ga:
            # the HIP path sums in a fixed order; the decoder's heads go through torch's index_select /
            # embedding backward, whose atomic adds make two direct runs differ in the last bits as well, and
            # everything upstream of the sampled tree vectors inherits that: bitwise where two direct runs are
            if torch.equal(ga[name], gc[name]):
                assert torch.equal(ga[name], gb[name]), name
            else:
                assert torch.allclose(ga[name], gb[name], rtol=1e-4, atol=1e-7), name
        assert torch.equal(ga["mpn.W_h.weight"], gc["mpn.W_h.weight"])  # the graph encoder sees no atomics at all

    def failing():
        yield next(batches())
        raise ValueError("loader broke")
    pf = Prefetcher(failing(), lambda b: warm_training_batch(model, b), "cuda:0")
    with pytest.raises(ValueError, match="loader broke"):
        for _ in pf:
            pass


@pytest.mark.gpu
@pytest.mark.parametrize("train", [True, False])
def test_prefetching_a_batch_with_a_deep_tree_prepares_both_of_its_parts(train):
    """A chain of 30 clusters among ordinary trees is split off the batch (it is deeper than the 20 iterations).
    The background thread prepares the split and both parts under the keys the encoder looks them up by, so the
    stepping thread finds everything ready - holders prepared for inference carry no backward tables - and the
    results are those of the direct call, bit for bit."""
    from fast_jtnn_hip import JTNNVAE, graphprep, synthetic, tensorize as T
    from fast_jtnn_hip.prefetch import Prefetcher, warm_encoder_holders

    V = 40
    model = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3)
    synthetic.fill_params(list(model.named_parameters()), seed=5, bias_scale=0.05)
    model = model.cuda()

    def batches():
        for i in range(3):
            trees = synthetic.make_batch(60 + i, 32, vocab_size=V, mean_atoms=20.0)
            chain = [synthetic.TreeNode(k + 1, (7 * k + i) % V) for k in range(30)]
            for a, b in zip(chain[:-1], chain[1:]):
                a.neighbors.append(b); b.neighbors.append(a)
            trees.insert(11, synthetic.Tree(chain, trees[0].mol))
            synthetic.set_batch_nodeID(trees)
            yield T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])

    def run(source, check_ready=False):
        out = []
        for jt, mp in source:
            if check_ready:
                cache = getattr(jt[3], graphprep._CACHE_ATTR)
                split = cache[("deep_split", 20)]
                assert split is not None and len(split) == 2
                for sub, _, _ in split:
                    prep = getattr(sub[3], graphprep._CACHE_ATTR)[("tree", "cuda:0", V)]
                    assert (prep.mcsr_t is not None) == train and len(prep.schedules) == 1
            if train:
                model.zero_grad()
                mean, logvar = model.encode_latent(jt, mp)
                (mean.sum() - logvar.sum()).backward()
                out.append([mean.detach().clone(), logvar.detach().clone()] +
                           [p.grad.clone() for p in model.parameters() if p.grad is not None])
            else:
                with torch.no_grad():
                    out.append([model.encode_vectors(jt, mp).clone()])
        return out

    direct = run(batches())
    with Prefetcher(batches(), lambda b: warm_encoder_holders(model, b[0], b[1], train=train), "cuda:0", depth=2) as pf:
        ahead = run(pf, check_ready=True)
    assert len(direct) == len(ahead) == 3
    for a, b in zip(direct, ahead):
        assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_bulk_encoding_at_job_scale_matches_oracle_on_a_sample(tmp_path):
    """The latent-generation job's shape: batches of 4096 graphs, vocabulary of 1251 entries, streamed through
    encode_bulk with prefetch; a 64-molecule sub-sample of the written rows against the CPU oracle (1e-4), the
    rest through finiteness and through equality with an unprefetched second pass."""
    from fast_jtnn_hip import JTNNVAE, synthetic, tensorize as T
    from oracle.ref_encoder import OracleEncoder

    V, n, bs = 1251, 8192 + 300, 4096  # two full batches and a short one
    trees = synthetic.make_batch(77, n, vocab_size=V, mean_atoms=30.0)
    model = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3, decoder=False)
    synthetic.fill_params(list(model.named_parameters()), seed=4, bias_scale=0.05)
    ora = OracleEncoder(V, 450, 56, 20, 3)
    synthetic.fill_params(list(ora.named_parameters()), seed=4, bias_scale=0.05)
    model = model.cuda()

    def make_holders(lo, hi):
        part = trees[lo:hi]
        synthetic.set_batch_nodeID(part)
        return T.jtenc_tensorize(part)[0], T.mpn_tensorize([t.mol for t in part])

    a = bulk.encode_bulk(model, n, make_holders, bs, str(tmp_path / "a"), what="vectors", prefetch=2)
    b = bulk.encode_bulk(model, n, make_holders, bs, str(tmp_path / "b"), what="vectors", prefetch=0)
    got, again = np.load(a), np.load(b)
    assert got.shape == (n, 900) and np.isfinite(got).all()
    assert np.array_equal(got, again)
    pick = np.random.RandomState(0).choice(n, size=64, replace=False)
    sample = [trees[i] for i in pick]
    synthetic.set_batch_nodeID(sample)
    with torch.no_grad():
        tv, _, mv = ora.encode(T.jtenc_tensorize(sample)[0], T.mpn_tensorize([t.mol for t in sample]))
    want = torch.cat([tv, mv], dim=1).numpy()
    assert np.abs(got[pick] - want).max() <= 1e-4 * np.abs(want).max()
