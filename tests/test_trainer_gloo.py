"""CPU, gloo: the training step of fast_jtnn_hip.trainer (bucket, all-rank skip flag, all-reduce before
the clip, Adam) on the conditional model - two ranks on the halves of a batch take the same optimizer steps
as one process on the whole batch, with the rsample noise injected.  The HIP modules have no CPU path, so
the model is the CPU oracle of JTpropVAE; the trainer is model-agnostic."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

B, V, H, L = 6, 40, 24, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _x_batch(lo, hi):
    from fast_jtnn_hip import synthetic, tensorize as T

    trees = synthetic.make_batch(11, B, vocab_size=V, mean_atoms=9.0, with_cands=True)[lo:hi]
    synthetic.set_batch_nodeID(trees)
    jt, mess_dict = T.jtenc_tensorize(trees)
    mp_ = T.mpn_tensorize([t.mol for t in trees])
    cands, bi = synthetic.collect_candidates(trees)
    rs = np.random.RandomState(3)
    props = np.stack([rs.standard_normal(B), rs.standard_normal(B), rs.randint(0, 2, B), rs.randint(0, 3, B)],
                     axis=1).astype(np.float32)[lo:hi]
    return (trees, torch.from_numpy(props), jt, mp_, (T.jtmpn_tensorize(cands, mess_dict), torch.from_numpy(bi)))


def _eps(step, lo, hi):
    rs = np.random.RandomState(100 + step)
    return [torch.from_numpy(rs.standard_normal((B, L // 2)).astype(np.float32)[lo:hi]) for _ in range(2)]


def _run(lo, hi, save_dir=None):
    from fast_jtnn_hip import synthetic, trainer
    from oracle.ref_vae import OracleJTpropVAE

    torch.set_num_threads(2)
    m = OracleJTpropVAE(V, H, L, 3, 2, ["denticity", "isomer"])
    synthetic.fill_params(list(m.named_parameters()), seed=5, bias_scale=0.05)
    tr = trainer.Trainer(m, trainer.Schedule(beta=0.1, save_iter=2, print_iter=1), save_dir=save_dir,
                         conditional=True, log=lambda *a: None)
    for step in range(2):
        res = tr.step(_x_batch(lo, hi), epsilons=_eps(step, lo, hi))  # fresh trees per step: the oracle's
        assert res.ok and np.isfinite(res.loss)                        # decoder rewires node.neighbors
    return tr, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fast_jtnn_hip import parallel

    parallel.init_process_group("gloo")
    lo, hi = parallel.shard_bounds(B, world, rank)
    tr, flat = _run(lo, hi, save_dir=os.path.join(out_dir, "ckpt"))
    np.save(os.path.join(out_dir, "params%d.npy" % rank), flat)
    # a batch that fails on one rank is skipped on both, and the parameters stay as they were
    before = flat.copy()
    bad = _x_batch(lo, hi)
    if rank == 1:
        bad = bad[:1] + (torch.zeros((hi - lo, 5)),) + bad[2:]  # five property columns: the model refuses them
    res = tr.step(bad, epsilons=_eps(7, lo, hi))
    after = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).numpy()
    np.save(os.path.join(out_dir, "skip%d.npy" % rank), np.asarray([float(res.ok), float(np.array_equal(before, after))]))
    torch.distributed.destroy_process_group()


def test_two_ranks_take_the_steps_of_one_rank_on_the_whole_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "params0.npy"), np.load(tmp_path / "params1.npy")
    assert np.array_equal(a, b)  # same reduced gradient, same clip, same Adam step on every rank
    _, one = _run(0, B)
    assert np.abs(a - one).max() <= 2e-5 * np.abs(one).max()
    for r in range(2):
        ok, unchanged = np.load(tmp_path / ("skip%d.npy" % r))
        assert ok == 0.0 and unchanged == 1.0
    # checkpoint: rank 0 wrote a state dict with the reference's keys after step 2 (save_iter)
    sd = torch.load(tmp_path / "ckpt" / "model.iter-2", weights_only=True)
    assert "decoder.U_r.weight" in sd and "jtnn.GRU.W_z.bias" in sd and "propNN.0.weight" in sd
    assert "isomerNN.2.bias" in sd and "jtmpn.W_o.weight" in sd


def test_schedule_follows_the_reference_loop():
    """KL weight and learning rate after N steps (vae_train.py:139-144) without running a model."""
    from fast_jtnn_hip import trainer

    m = torch.nn.Linear(2, 2)
    tr = trainer.Trainer(m, trainer.Schedule(beta=0.0, step_beta=0.002, max_beta=0.005, warmup=4, kl_anneal_iter=2,
                                             anneal_iter=3, anneal_rate=0.5, lr=1e-3, print_iter=100, save_iter=100),
                         log=lambda *a: None)
    tr.forward_backward = lambda batch, eps=None: ((m.weight.sum() * 0 + 1.0).backward() or (torch.tensor(1.0), {
        "kl_div": 0.0, "word_acc": 0.0, "topo_acc": 0.0, "assm_acc": 0.0}))
    betas, lrs = [], []
    for _ in range(10):
        tr.step(None)
        betas.append(round(tr.beta, 6))
        lrs.append(tr.scheduler.get_last_lr()[0])
    assert betas == [0.0, 0.0, 0.0, 0.002, 0.002, 0.004, 0.004, 0.005, 0.005, 0.005]
    assert np.allclose(lrs, [1e-3] * 2 + [5e-4] * 3 + [2.5e-4] * 3 + [1.25e-4] * 2)


def test_bench_and_trainer_start_their_own_ranks():
    from fast_jtnn_hip import parallel

    cmd = parallel.launch_command("/x/bench.py", ["--gpus", "4", "--steps", "3"], 4, port=2345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "2345" and cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]
    os.environ.pop("WORLD_SIZE", None)
    parallel.self_launch_if_needed("/x/bench.py", [], 1)  # one GPU: returns without starting anything


def test_self_launched_ranks_leave_one_json_line_on_stdout(tmp_path):
    """The launcher form of ``bench.py --gpus N``: two ranks started by the parent, banners that the ranks (or
    the communication library) write to stdout end up on stderr, the JSON line alone on stdout, and the
    parent exits with the ranks' return code."""
    import subprocess
    import sys

    script = tmp_path / "two_ranks.py"
    script.write_text(
        "import os, sys\n"
        "print('[banner] rank %s up' % os.environ['RANK'], flush=True)\n"
        "if os.environ['RANK'] == '0':\n"
        "    print('{\"value\": 1.5, \"n_gpus\": %s}' % os.environ['WORLD_SIZE'], flush=True)\n"
        "sys.exit(3 if len(sys.argv) > 1 else 0)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys; sys.path.insert(0, %r); from fast_jtnn_hip import parallel; "
            "parallel.self_launch_if_needed(%r, sys.argv[1:], 2, json_only=True)"
            % (os.path.join(root, "jt-vae-tmcinvdes_amd"), str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    ok = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-1500:]
    assert ok.stdout.strip() == '{"value": 1.5, "n_gpus": 2}'
    assert "[banner] rank 0 up" in ok.stderr and "[banner] rank 1 up" in ok.stderr
    bad = subprocess.run([sys.executable, "-c", prog, "fail"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
