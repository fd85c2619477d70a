"""CPU, 2 processes over gloo: the data-parallel layer (shard by molecule, one flat gradient bucket,
mean all-reduce) gives the same gradients as one process on the concatenated batch.

The HIP modules have no CPU path, so the model here is the CPU oracle; what is under test is
fast_jtnn_hip.parallel, which is model-agnostic.
"""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from fast_jtnn_hip import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(trees):
    from fast_jtnn_hip import synthetic
    from fast_jtnn_hip import tensorize as T

    synthetic.set_batch_nodeID(trees)
    jt, _ = T.jtenc_tensorize(trees)
    mp_ = T.mpn_tensorize([t.mol for t in trees])
    return jt, mp_


def _model():
    from fast_jtnn_hip import synthetic
    from oracle.ref_encoder import OracleEncoder

    m = OracleEncoder(30, 32, 16, 3, 2)
    synthetic.fill_params(list(m.named_parameters()), seed=7, bias_scale=0.05)
    for n, p in m.named_parameters():
        if n.startswith("jtmpn.") or n.startswith("A_assm."):
            p.requires_grad_(False)
    return m


def _loss(m, jt, mp_):
    mean, logvar = m.encode_latent(jt, mp_)
    return (mean.sum() + (logvar * logvar).sum()) / mean.shape[0]  # normalised by the local batch


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, _, w = parallel.init_process_group("gloo")
    assert (r, w) == (rank, world)
    from fast_jtnn_hip import synthetic

    trees = synthetic.make_batch(3, 8, vocab_size=30, mean_atoms=10.0)
    lo, hi = parallel.shard_bounds(len(trees), world, rank)
    jt, mp_ = _build(trees[lo:hi])
    m = _model()
    bucket = parallel.FlatGradBucket([p for p in m.parameters() if p.requires_grad])
    for _ in range(2):  # second pass checks that zero() really drops the first pass's gradients
        bucket.zero()
        _loss(m, jt, mp_).backward()
        bucket.all_reduce_mean()
    assert parallel.all_ranks_ok(True) is True
    assert parallel.all_ranks_ok(rank != 1) is False  # one failing rank skips the step everywhere
    np.save(os.path.join(out_dir, "flat%d.npy" % rank), bucket.flat.numpy())
    torch.distributed.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "flat0.npy")
    b = np.load(tmp_path / "flat1.npy")
    assert np.array_equal(a, b)  # every rank holds the same reduced gradient
    from fast_jtnn_hip import synthetic

    trees = synthetic.make_batch(3, 8, vocab_size=30, mean_atoms=10.0)
    jt, mp_ = _build(trees)
    m = _model()
    bucket = parallel.FlatGradBucket([p for p in m.parameters() if p.requires_grad])
    bucket.zero()
    _loss(m, jt, mp_).backward()
    bucket.pack()  # a single rank has no collective and leaves the gradients where autograd put them
    ref = bucket.flat.numpy()
    assert np.abs(a - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_shard_helpers():
    assert [parallel.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert parallel.shard_bounds(3, 8, 7) == (3, 3)
    parts = parallel.shard_by_cost([5, 1, 1, 1, 4, 4], 2)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([5, 1, 1, 1, 4, 4][i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 1
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def test_bucket_packs_whatever_gradients_the_parameters_hold():
    lin = torch.nn.Linear(3, 2)
    extra = torch.nn.Parameter(torch.zeros(4))  # takes no part in the loss: its slot must read zero
    b = parallel.FlatGradBucket(list(lin.parameters()) + [extra])
    b.flat.fill_(7.0)
    b.zero()
    assert all(p.grad is None for p in b.params)
    lin(torch.ones(1, 3)).sum().backward()
    w_grad = lin.weight.grad.clone()
    b.pack()
    assert torch.equal(b.flat[:6].view(2, 3), w_grad) and torch.equal(b.flat[8:], torch.zeros(4))
    assert lin.weight.grad.data_ptr() == b.flat.data_ptr()  # .grad now lives in the flat buffer
    b.pack()  # packing twice is harmless
    assert torch.equal(b.flat[:6].view(2, 3), w_grad)
    lin.zero_grad(set_to_none=True)  # a caller's own reset is as good as zero()
    lin(torch.ones(1, 3)).sum().backward()
    b.pack()
    assert torch.equal(b.flat[:6].view(2, 3), w_grad)
