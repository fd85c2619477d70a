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
