"""GPU: the RCCL leg of the training step on the one GPU a test box has.  A process group of world size 1 over
the `nccl` backend (RCCL on ROCm) loads the library, creates a communicator and runs the collective as a kernel
on the launch stream: the flat gradient bucket is packed, reduced and viewed back into `.grad` exactly as on N
GPUs, only the sum has one term."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_world_of_one():
    if dist.is_initialized():
        pytest.skip("a process group exists already")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_trainer_step_through_rccl_matches_the_step_without_a_process_group(rccl_world_of_one, monkeypatch):
    from fast_jtnn_hip import JTNNVAE, synthetic, trainer
    from fast_jtnn_hip import tensorize as T

    V, B = 61, 24
    trees = synthetic.make_batch(17, B, vocab_size=V, mean_atoms=20.0, with_cands=True)
    x_batch = T.tensorize(trees, assm=True)
    g = torch.Generator().manual_seed(3)
    eps = (torch.randn((B, 28), generator=g).cuda(), torch.randn((B, 28), generator=g).cuda())

    def run(force_collective):
        m = JTNNVAE(synthetic.FakeVocab(V), 450, 56, 20, 3)
        synthetic.fill_params(list(m.named_parameters()), seed=2, bias_scale=0.02)
        m = m.cuda()
        tr = trainer.Trainer(m, trainer.Schedule(beta=0.1, print_iter=10 ** 9, save_iter=10 ** 9), log=lambda *a: None)
        tr.force_collective = force_collective  # world size 1 would short-cut the bucket: take the N-rank path
        seen = []
        if force_collective:
            orig = dist.all_reduce

            def spy(t, *a, **k):
                seen.append((dist.get_backend(), int(t.numel())))
                return orig(t, *a, **k)
            monkeypatch.setattr(dist, "all_reduce", spy)
        res = tr.step(x_batch, eps)
        assert res.ok
        if force_collective:
            monkeypatch.undo()
            assert ("nccl", tr.bucket.flat.numel()) in seen  # the bucket went through RCCL in one piece
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(tr.bucket.params, tr.bucket.views))
        torch.cuda.synchronize()
        return res.loss, {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}

    torch.use_deterministic_algorithms(True, warn_only=True)  # torch's index_add in the decoder heads: no atomics
    try:
        loss_a, g_a = run(False)
        loss_b, g_b = run(True)
    finally:
        torch.use_deterministic_algorithms(False)
    assert loss_a == loss_b
    assert g_a.keys() == g_b.keys()
    for k in g_a:  # the same (clipped) gradient: the bucket only changes where it lives
        d = float(np.linalg.norm(g_a[k] - g_b[k]) / max(np.linalg.norm(g_a[k]), 1e-30))
        assert d <= 1e-6, (k, d)


def test_all_rank_flag_and_barrier_on_rccl(rccl_world_of_one):
    from fast_jtnn_hip import parallel

    dev = torch.device("cuda", 0)
    assert parallel.all_ranks_ok(True, dev, torch.tensor(True, device=dev))
    assert not parallel.all_ranks_ok(True, dev, torch.tensor(False, device=dev))
    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    assert float(t.sum()) == float(1 << 20)
