"""GPU: the training-side callers (rsample, teacher-forced decoder, assembly loss, JTNNVAE.forward,
JTpropVAE.forward) through the HIP path against what the reference itself returned on the golden
batches (north-star bound: losses within 1e-4 relative), and against the CPU oracle on larger batches."""
import numpy as np
import pytest
import torch

from common import (RTOL, Golden, check_grad_stats, fp64_training_grads, fro_err, golden_cases, golden_x_batch,
                    make_hip_model, make_oracle_vae, rel_err)

pytestmark = pytest.mark.gpu
CASES = golden_cases()
PROP_CASES = [p for p in CASES if "prop_loss" in np.load(p).files]


def close(a, b, tol=RTOL):
    return abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1])
def test_forward_pieces_match_reference_golden(path):
    g = Golden(path)
    z = g.z
    m = make_hip_model(g, "cuda")
    trees, jt, mp, cj = golden_x_batch(g)
    et, eg = torch.from_numpy(z["fwd_eps_tree"]).cuda(), torch.from_numpy(z["fwd_eps_mol"]).cuda()
    with torch.no_grad():
        tv, ms, mv = m.encode(jt, mp)
        z_tree, tree_kl = m.rsample(tv, m.T_mean, m.T_var, et)
        z_mol, mol_kl = m.rsample(mv, m.G_mean, m.G_var, eg)
        word_loss, topo_loss, word_acc, topo_acc = m.decoder(trees, z_tree)
        assm_loss, assm_acc = m.assm(trees, cj, z_mol, ms)
    assert close(tree_kl.item(), z["fwd_tree_kl"]) and close(mol_kl.item(), z["fwd_mol_kl"])
    assert rel_err(z_tree.cpu().numpy(), z["fwd_z_tree"]) < RTOL and rel_err(z_mol.cpu().numpy(), z["fwd_z_mol"]) < RTOL
    assert np.allclose(z_tree.cpu().numpy(), z["fwd_z_tree"], rtol=1e-4, atol=1e-5)
    assert close(word_loss.item(), z["fwd_word_loss"]) and close(topo_loss.item(), z["fwd_topo_loss"])
    assert close(assm_loss.item(), z["fwd_assm_loss"])
    assert abs(word_acc - float(z["fwd_word_acc"])) < 1e-6 and abs(topo_acc - float(z["fwd_topo_acc"])) < 1e-6
    assert abs(assm_acc - float(z["fwd_assm_acc"])) < 1e-6


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1])
def test_training_forward_backward_matches_reference_golden(path):
    g = Golden(path)
    z = g.z
    m = make_hip_model(g, "cuda")
    eps = (torch.from_numpy(z["fwd_eps_tree"]).cuda(), torch.from_numpy(z["fwd_eps_mol"]).cuda())
    loss, kl, wacc, tacc, sacc = m(golden_x_batch(g), float(z["fwd_beta"]), epsilons=eps)
    assert close(loss.item(), z["fwd_loss"]) and close(kl, z["fwd_kl"])
    assert np.allclose([wacc, tacc, sacc], z["fwd_accs"], atol=1e-6)
    loss.backward()
    check_grad_stats(z, "fg", m.named_parameters(), 2e-4, 5e-4, exact=fp64_training_grads(g))


@pytest.mark.parametrize("path", PROP_CASES, ids=lambda p: p.split("/")[-1])
def test_conditional_forward_backward_matches_reference_golden(path):
    from fast_jtnn_hip import JTpropVAE, synthetic

    g = Golden(path)
    z = g.z
    m = JTpropVAE(synthetic.FakeVocab(g.V), g.H, g.L, g.dT, g.dG, ["denticity", "isomer"])
    synthetic.fill_params(list(m.named_parameters()), seed=g.seed, bias_scale=g.bias_scale)
    m = m.cuda()
    eps = (torch.from_numpy(z["prop_eps_tree"]).cuda(), torch.from_numpy(z["prop_eps_mol"]).cuda())
    trees, jt, mp, cj = golden_x_batch(g)
    loss, metrics = m((trees, torch.from_numpy(z["prop_table"]), jt, mp, cj), float(z["fwd_beta"]), epsilons=eps)
    assert close(loss.item(), z["prop_loss"])
    assert list(metrics.keys()) == [str(k) for k in z["prop_metric_names"]]
    for (k, got), want in zip(metrics.items(), z["prop_metrics"]):
        assert close(got, want), (k, got, want)
    loss.backward()
    check_grad_stats(z, "pg", m.named_parameters(), 2e-4, 5e-4,
                     exact=fp64_training_grads(g, ["denticity", "isomer"]))


def _seeded_training_problem(seed, B, atoms, V=120):
    from fast_jtnn_hip import synthetic, tensorize as T

    class G:
        pass
    g = G()
    g.seed, g.B, g.H, g.L, g.dT, g.dG, g.V, g.bias_scale = seed, B, 450, 56, 20, 3, V, 0.05

    def x_batch():
        trees = synthetic.make_batch(seed, B, vocab_size=V, mean_atoms=atoms, with_cands=True)
        jt, mess_dict = T.jtenc_tensorize(trees)
        mp = T.mpn_tensorize([t.mol for t in trees])
        cands, bi = synthetic.collect_candidates(trees)
        return trees, jt, mp, (T.jtmpn_tensorize(cands, mess_dict), torch.from_numpy(bi))
    return g, x_batch


@pytest.mark.parametrize("seed,B,atoms", [(31, 24, 30.0), (32, 3, 8.0)])
def test_training_step_matches_oracle_on_fresh_batches(seed, B, atoms):
    """Loss terms and every gradient tensor of JTNNVAE.forward against the fp32 CPU oracle (decoder weights
    at the path's 1e-4; encoder-side tensors as in the encoder tests)."""
    g, x_batch = _seeded_training_problem(seed, B, atoms)
    hip, ora = make_hip_model(g, "cuda"), make_oracle_vae(g)
    rs = np.random.RandomState(seed)
    eps = [torch.from_numpy(rs.standard_normal((B, 28)).astype(np.float32)) for _ in range(2)]
    la = hip(x_batch(), 0.5, epsilons=[e.cuda() for e in eps])
    lb = ora(x_batch(), 0.5, epsilons=eps)
    assert close(la[0].item(), lb[0].item()) and close(la[1], lb[1])
    assert np.allclose(la[2:], lb[2:], atol=1e-6)
    la[0].backward()
    lb[0].backward()
    pb = dict(ora.named_parameters())
    for name, p in hip.named_parameters():
        assert p.grad is not None, name
        f = fro_err(p.grad.cpu().numpy(), pb[name].grad.numpy())
        assert f < (1e-3 if name.startswith("jtmpn.") else 2e-4), (name, f)


@pytest.mark.parametrize("B,atoms,crowded", [(256, 40.0, True), (256, 40.0, False), (5, 12.0, True)])
def test_two_stream_step_is_bitwise_the_one_stream_step(B, atoms, crowded):
    """The encoders on two overlapping streams (the tree encoder's backward pass spreading its closing groups
    of launches over a helper stream) against everything on one stream: the same launches on the same data, so
    every output and every gradient carries the same bits - checked over several steps, so that a buffer handed
    back to the allocator while another stream still reads it would show.  The tiling of the tree encoder's short
    products, which by default follows the stream layout (JTNNEncoder.crowded), is pinned to one setting for both
    layouts: that choice changes the order of summation, the streams do not."""
    g, x_batch = _seeded_training_problem(77, B, atoms)
    hip = make_hip_model(g, "cuda")
    hip.jtnn.crowded = crowded
    hip.overlap_min_messages = 0  # two streams whatever the batch size
    _, jt, mp, _ = x_batch()

    def step(overlap):
        hip.overlap_streams = overlap
        for p in hip.parameters():
            p.grad = None
        mean, logvar = hip.encode_latent(jt, mp)
        w = torch.linspace(0.5, 1.5, mean.shape[1], device="cuda")
        ((mean * w).sum() + (logvar * w.flip(0)).sum()).backward()
        torch.cuda.synchronize()
        return ([mean.detach().clone(), logvar.detach().clone()] +
                [p.grad.clone() for n, p in hip.named_parameters() if p.grad is not None])

    ref = step(False)
    assert len(ref) > 12
    for it in range(4):
        got = step(True)
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), it
    again = step(False)
    assert all(torch.equal(a, b) for a, b in zip(again, ref))

    # the same with the candidate-graph encoder reading the tree messages, so that their gradient enters the
    # tree encoder's backward pass from outside (encoders, both KL terms and the assembly loss; no decoder, whose
    # torch-side index adds are not reproducible run to run)
    xb = x_batch()
    trees, jt, mp, assm_holder = xb
    gen = torch.Generator(device="cpu").manual_seed(9)
    eps = [torch.randn((B, 28), generator=gen).cuda() for _ in range(2)]

    def assm_step(overlap):
        hip.overlap_streams = overlap
        for p in hip.parameters():
            p.grad = None
        tree_vecs, tree_mess, mol_vecs = hip.encode(jt, mp)
        _, tree_kl = hip.rsample(tree_vecs, hip.T_mean, hip.T_var, eps[0])
        z_mol, mol_kl = hip.rsample(mol_vecs, hip.G_mean, hip.G_var, eps[1])
        assm_loss, _ = hip.assm(trees, assm_holder, z_mol, tree_mess)
        loss = assm_loss + 0.01 * (tree_kl + mol_kl)
        loss.backward()
        torch.cuda.synchronize()
        return [loss.detach().clone()] + [p.grad.clone() for p in hip.parameters() if p.grad is not None]

    ref = assm_step(False)
    for it in range(3):
        got = assm_step(True)
        assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref)), it


def test_decoder_alone_matches_oracle_and_leaves_the_batch_untouched():
    g, x_batch = _seeded_training_problem(41, 16, 24.0, V=60)
    hip, ora = make_hip_model(g, "cuda"), make_oracle_vae(g)
    rs = np.random.RandomState(5)
    zt = torch.from_numpy(rs.standard_normal((16, 28)).astype(np.float32))
    trees_a, trees_b = x_batch()[0], x_batch()[0]
    order_before = [[y.idx for y in n.neighbors] for t in trees_a for n in t.nodes]
    za = zt.cuda().requires_grad_(True)
    zb = zt.clone().requires_grad_(True)
    a = hip.decoder(trees_a, za)
    b = ora.decoder(trees_b, zb)
    assert [[y.idx for y in n.neighbors] for t in trees_a for n in t.nodes] == order_before
    assert close(a[0].item(), b[0].item()) and close(a[1].item(), b[1].item())
    assert abs(a[2] - b[2]) < 1e-6 and abs(a[3] - b[3]) < 1e-6
    (a[0] + 2.0 * a[1]).backward()
    (b[0] + 2.0 * b[1]).backward()
    assert fro_err(za.grad.cpu().numpy(), zb.grad.numpy()) < RTOL
    pb = dict(ora.decoder.named_parameters())
    for name, p in hip.decoder.named_parameters():
        assert fro_err(p.grad.cpu().numpy(), pb[name].grad.numpy()) < RTOL, name


def test_decoder_edge_cases_single_node_and_two_node_trees():
    from fast_jtnn_hip import synthetic

    g, _ = _seeded_training_problem(1, 2, 8.0, V=9)
    g.H, g.L = 32, 8
    hip, ora = make_hip_model(g, "cuda"), make_oracle_vae(g)

    def trees():
        rs = np.random.RandomState(0)
        mol = synthetic.make_mol(rs, 6.0, 1.0, 4, 8)
        t1 = synthetic.Tree([synthetic.TreeNode(1, 2)], mol)
        n1, n2 = synthetic.TreeNode(1, 0), synthetic.TreeNode(2, 4)
        n1.neighbors.append(n2); n2.neighbors.append(n1)
        out = [t1, synthetic.Tree([n1, n2], mol)]
        synthetic.set_batch_nodeID(out)
        return out
    zt = torch.from_numpy(np.random.RandomState(1).standard_normal((2, 4)).astype(np.float32))
    for batch in (lambda: trees()[:1], trees):  # one single-node tree alone (no message at all), then both
        zb = zt[:len(batch())]
        a = hip.decoder(batch(), zb.cuda())
        b = ora.decoder(batch(), zb)
        assert close(a[0].item(), b[0].item()) and close(a[1].item(), b[1].item())
        assert abs(a[2] - b[2]) < 1e-6 and abs(a[3] - b[3]) < 1e-6


def test_broadcast_gradients_reach_the_kernels():
    """The gradient of tree_vecs.sum(0) / messages.sum(0) is an expanded (stride 0) tensor."""
    g = Golden(CASES[1])
    m = make_hip_model(g, "cuda")
    tv, ms, mv = m.encode(g.jtenc_holder(), g.mpn_holder())
    w = torch.linspace(-1, 1, g.H, device="cuda")
    (tv.sum(0).dot(w) + mv.sum(0).dot(w) + ms.sum(0).dot(w)).backward()
    ref = make_hip_model(g, "cuda")
    tv2, ms2, mv2 = ref.encode(g.jtenc_holder(), g.mpn_holder())
    ((tv2 * w).sum() + (mv2 * w).sum() + (ms2 * w).sum()).backward()
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert torch.equal(p.grad, q.grad), n


def test_full_training_step_at_batch_4096_is_the_mean_of_its_halves():
    """The full-step configuration (4096 molecules per GPU, JTNNVAE.forward + backward + clip + Adam through
    the trainer).  No CPU oracle finishes this size in seconds, so the check is the path's linearity: every
    loss term is a sum over molecules divided by the batch size (fast_jtnn/jtnn_vae.py:85-103, 137;
    jtnn_dec.py:183-190), hence loss and gradients of the batch are the means of those of its two halves.

    The linearity is checked with every product on the fp32 matrix instruction, where batch and halves run the same
    fma chains (1e-4).  The default form of the large products (fp32 operands split into bf16 parts) rounds
    differently in the last bits, and this model's gradients amplify last-bit differences (ReLU units at the
    switching point: the reference's own fp32 run deviates from its fp64 run by up to 2.8e-3 on the golden
    cases), so the default form is held to the fp32 form at 1e-5 on the loss and 2e-3 on every gradient."""
    from fast_jtnn_hip import _capi, synthetic, tensorize as T, trainer

    B, V = 4096, 385
    trees = synthetic.make_batch(4242, B, vocab_size=V, mean_atoms=40.0, with_cands=True)

    def x_batch(part):
        synthetic.set_batch_nodeID(part)
        jt, mess_dict = T.jtenc_tensorize(part)
        mp = T.mpn_tensorize([t.mol for t in part])
        cands, bi = synthetic.collect_candidates(part)
        return part, jt, mp, (T.jtmpn_tensorize(cands, mess_dict), torch.from_numpy(bi))

    class G:
        pass
    g = G()
    g.seed, g.B, g.H, g.L, g.dT, g.dG, g.V, g.bias_scale = 7, B, 450, 56, 20, 3, V, 0.05
    m = make_hip_model(g, "cuda")
    rs = np.random.RandomState(5)
    eps = [torch.from_numpy(rs.standard_normal((B, 28)).astype(np.float32)).cuda() for _ in range(2)]
    beta = 0.005

    def run(lo, hi):
        m.zero_grad(set_to_none=True)
        out = m(x_batch(trees[lo:hi]), beta, epsilons=[e[lo:hi] for e in eps])
        out[0].backward()
        torch.cuda.synchronize()
        return ([float(out[0].item()), float(out[1])] + [float(v) for v in out[2:]],
                {n: p.grad.detach().clone() for n, p in m.named_parameters()})

    full_split, g_split = run(0, B)  # default: large products in the split form
    was = _capi.set_split_products(False)
    try:
        full, g_full = run(0, B)
        a, g_a = run(0, B // 2)
        b, g_b = run(B // 2, B)
    finally:
        _capi.set_split_products(was)
    assert was, "the split form is the default"
    assert all(np.isfinite(full)) and all(np.isfinite(full_split))
    assert close(full_split[0], full[0], 1e-5) and close(full_split[1], full[1], 1e-5)
    for name, gf in g_full.items():
        err = float((g_split[name] - gf).norm() / gf.norm().clamp_min(1e-30))
        assert err < 2e-3, (name, err)
    assert close(full[0], 0.5 * (a[0] + b[0]), 1e-5) and close(full[1], 0.5 * (a[1] + b[1]), 1e-5)
    for name, gf in g_full.items():
        mean = 0.5 * (g_a[name] + g_b[name])
        err = float((gf - mean).norm() / mean.norm().clamp_min(1e-30))
        assert err < 1e-4, (name, err)

    # one iteration of the training loop on the whole batch: same loss, clipped gradient, first Adam step
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    tr = trainer.Trainer(m, trainer.Schedule(beta=beta, print_iter=10 ** 9, save_iter=10 ** 9, anneal_iter=10 ** 9,
                                             kl_anneal_iter=10 ** 9), log=lambda *a: None)
    res = tr.step(x_batch(trees), epsilons=eps)
    assert res.ok and close(res.loss, full_split[0], 1e-5)
    g_full = g_split  # the trainer ran in the default form, like the first run above
    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g_full.values())))
    moved = 0
    for name, p in m.named_parameters():
        assert torch.isfinite(p).all(), name
        # the gradient in the bucket is the full-batch gradient scaled by the clip factor
        scale = min(1.0, 50.0 / (total + 1e-6))
        assert float((p.grad - scale * g_full[name]).norm() / (scale * g_full[name]).norm().clamp_min(1e-30)) < 1e-4, name
        step = (p.detach() - before[name]).abs().max().item()
        assert step <= 1e-3 * 1.001  # Adam's first step moves every weight by at most the learning rate
        moved += step > 0
    assert moved == len(before)


def test_training_cli_writes_checkpoints_the_oracle_model_loads(tmp_path):
    """train.py end to end (the reference's loop, fast_molvae/vae_train.py:97-146, under its flag names): prints the
    reference's progress lines, anneals, and saves state dicts whose keys and shapes are the reference model's."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(
        [sys.executable, os.path.join(root, "train.py"), "--train", "synthetic:192:3", "--vocab", "60", "--save_dir",
         str(tmp_path), "--batch_size", "64", "--epoch", "2", "--print_iter", "2", "--save_iter", "4", "--warmup", "2",
         "--kl_anneal_iter", "2", "--anneal_iter", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "[2] Loss:" in out.stdout and "learning rate: 0.000900" in out.stdout
    names = sorted(os.listdir(tmp_path))
    assert "model.iter-4" in names and "model.epoch-6" in names
    state = torch.load(os.path.join(tmp_path, "model.epoch-6"), map_location="cpu", weights_only=True)

    class G:
        pass
    g = G()
    g.seed, g.B, g.H, g.L, g.dT, g.dG, g.V, g.bias_scale = 0, 64, 450, 56, 20, 3, 60, 0.0
    ora = make_oracle_vae(g)
    want = ora.state_dict()
    assert set(state) == set(want)
    for k, v in want.items():
        assert tuple(state[k].shape) == tuple(v.shape), k
        assert torch.isfinite(state[k]).all(), k
    ora.load_state_dict(state)


def test_training_cli_from_a_flat_data_set(tmp_path):
    """train.py --train flat:<file>: the conditional loop (fast_molopt/vae_train_prop.py:116-150) on a data set stored
    as flat arrays, batches tensorized natively in the prefetch threads."""
    import os
    import subprocess
    import sys

    from fast_jtnn_hip import flat, synthetic

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    trees = synthetic.make_batch(4, 160, vocab_size=45, mean_atoms=16.0, with_cands=True)
    rs = np.random.RandomState(0)
    props = np.stack([rs.standard_normal(160), rs.standard_normal(160), rs.randint(0, 2, 160), rs.randint(0, 3, 160)],
                     axis=1).astype(np.float32)
    data = str(tmp_path / "train.npz")
    flat.save_shard(data, flat.shard_from_trees(trees, with_cands=True), props)
    out = subprocess.run(
        [sys.executable, os.path.join(root, "train.py"), "--train", "flat:" + data, "--vocab", "45", "--save_dir",
         str(tmp_path / "out"), "--batch_size", "64", "--epoch", "2", "--print_iter", "1", "--conditional",
         "--train_mode", "denticity", "isomer"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("Loss:") == 4 and "prop_loss" in out.stdout and "isomer_loss" in out.stdout
    assert "model.epoch-4" in os.listdir(tmp_path / "out")


def test_default_tile_policy_of_the_two_stream_step_stays_within_rounding():
    """By default the tree encoder's short products change tiling when the encoders overlap (JTNNEncoder.crowded
    = None).  The latent means and variances of the two layouts then differ by summation order only: 1e-5 of the
    largest entry (fp32 sums of 450 terms, five levels deep)."""
    g, x_batch = _seeded_training_problem(78, 64, 30.0)
    hip = make_hip_model(g, "cuda")
    hip.overlap_min_messages = 0
    assert hip.jtnn.crowded is None
    _, jt, mp, _ = x_batch()
    out = {}
    for overlap in (False, True):
        hip.overlap_streams = overlap
        with torch.no_grad():
            mean, logvar = hip.encode_latent(jt, mp)
        torch.cuda.synchronize()
        out[overlap] = torch.cat([mean, logvar], 1).double()
    scale = float(out[False].abs().max())
    assert float((out[True] - out[False]).abs().max()) < 1e-5 * scale


@pytest.mark.parametrize("B,H,L,which", [(64, 450, 28, "both"), (1, 450, 28, "both"), (37, 450, 28, "mean"),
                                         (37, 450, 28, "var"), (19, 30, 5, "both"), (300, 450, 28, "frozen_input")])
def test_latent_heads_match_linear_layers_in_fp64(B, H, L, which):
    """JTNNVAE.latent_heads (mean = W_m x + b_m, log-variance = -|W_v x + b_v|, fast_jtnn/jtnn_vae.py:59-64) on
    the package's products against nn.Linear in fp64: outputs, input gradient, weight and bias gradients; with
    only one of the two outputs used, a latent size that is no multiple of four, and an input without gradient."""
    from fast_jtnn_hip import JTNNVAE
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + L)
    Wm, Wv = torch.nn.Linear(H, L), torch.nn.Linear(H, L)
    with torch.no_grad():
        for p in list(Wm.parameters()) + list(Wv.parameters()):
            p.copy_(0.3 * torch.randn(p.shape, generator=gen))
    x = torch.randn((B, H), generator=gen)
    gm, gv = torch.randn((B, L), generator=gen), torch.randn((B, L), generator=gen)

    def run(dev, dtype):
        m, v = (torch.nn.Linear(H, L).to(device=dev, dtype=dtype) for _ in range(2))
        m.load_state_dict(Wm.state_dict())
        v.load_state_dict(Wv.state_dict())
        xx = x.to(device=dev, dtype=dtype).requires_grad_(which != "frozen_input")
        if dev == "cuda":
            mean, lv = JTNNVAE.latent_heads(xx, m, v)
        else:
            mean, lv = m(xx), -torch.abs(v(xx))
        loss = 0.0
        if which != "var":
            loss = loss + (mean * gm.to(device=dev, dtype=dtype)).sum()
        if which != "mean":
            loss = loss + (lv * gv.to(device=dev, dtype=dtype)).sum()
        loss.backward()
        out = {"mean": mean, "lv": lv, "dx": xx.grad}
        for tag, mod in (("m", m), ("v", v)):
            out["dW" + tag], out["db" + tag] = mod.weight.grad, mod.bias.grad
        return {k: (None if t is None else t.detach().double().cpu()) for k, t in out.items()}

    got, want = run("cuda", torch.float32), run("cpu", torch.float64)
    for k in want:
        if want[k] is None:
            assert got[k] is None or float(got[k].abs().max()) == 0.0, k
            continue
        assert got[k] is not None, k
        scale = float(want[k].abs().max()) + 1e-30
        assert float((got[k] - want[k]).abs().max()) < 2e-5 * scale, k


@pytest.mark.parametrize("n,nh,m,ns,out_f,h_first,relu", [(700, 450, 33, 28, 450, True, True),
                                                          (12001, 450, 385, 450, 450, False, True),
                                                          (5, 450, 3, 28, 450, True, False),
                                                          (257, 30, 9, 6, 22, False, True)])
def test_table_linear_matches_the_concatenated_linear_layer_in_fp64(n, nh, m, ns, out_f, h_first, relu):
    """linear.table_linear = act(lin(cat([h, src[idx]]))) (or the other order) as the decoder's heads use it
    (fast_jtnn/jtnn_dec.py:42-53, 160-163): output and the gradients of h, src, weight and bias against the
    concatenation run in fp64; rows of src that no index names get a zero gradient."""
    from fast_jtnn_hip.linear import table_linear
    gen = torch.Generator(device="cpu").manual_seed(n + 7 * m)
    lin = torch.nn.Linear(nh + ns, out_f)
    with torch.no_grad():
        lin.weight.copy_(0.2 * torch.randn(lin.weight.shape, generator=gen))
        lin.bias.copy_(0.2 * torch.randn(lin.bias.shape, generator=gen))
    h, src = torch.randn((n, nh), generator=gen), torch.randn((m, ns), generator=gen)
    idx = torch.randint(0, max(m - 1, 1), (n,), generator=gen)  # the last row of src is never read
    g = torch.randn((n, out_f), generator=gen)

    def run(dev, dtype, on=None):
        mod = torch.nn.Linear(nh + ns, out_f).to(device=dev, dtype=dtype)
        mod.load_state_dict(lin.state_dict())
        hh = h.to(device=dev, dtype=dtype).requires_grad_(True)
        ss = src.to(device=dev, dtype=dtype).requires_grad_(True)
        if dev == "cuda":
            y = table_linear(hh, ss, idx.to(device=dev, dtype=torch.int32), mod, h_first=h_first, relu=relu)
        else:
            parts = [hh, ss[idx]] if h_first else [ss[idx], hh]
            y = mod(torch.cat(parts, 1))
            if relu:
                # a unit whose pre-activation is within rounding of zero may switch differently in fp32: the
                # reference takes the switches of the run it is compared with (and checks that only such units differ)
                differ = y.detach()[(y.detach() > 0) != on]
                assert differ.numel() == 0 or float(differ.abs().max()) < 1e-4
                y = y * on
        (y * g.to(device=dev, dtype=dtype)).sum().backward()
        out = {"y": y, "dh": hh.grad, "dsrc": ss.grad, "dW": mod.weight.grad, "db": mod.bias.grad}
        return {k: t.detach().double().cpu() for k, t in out.items()}

    got = run("cuda", torch.float32)
    want = run("cpu", torch.float64, got["y"] > 0)
    for k in want:
        scale = float(want[k].abs().max()) + 1e-30
        assert float((got[k] - want[k]).abs().max()) < 3e-5 * scale, k
    assert float(got["dsrc"][m - 1].abs().max()) == 0.0 or m == 1


@pytest.mark.parametrize("n,kin,out_f", [(900, 450, 385), (20000, 450, 385), (3, 450, 1), (64, 30, 7)])
def test_linear_matches_nn_linear_in_fp64(n, kin, out_f):
    from fast_jtnn_hip.linear import linear
    gen = torch.Generator(device="cpu").manual_seed(n + kin)
    lin = torch.nn.Linear(kin, out_f)
    with torch.no_grad():
        lin.weight.copy_(0.2 * torch.randn(lin.weight.shape, generator=gen))
        lin.bias.copy_(0.2 * torch.randn(lin.bias.shape, generator=gen))
    x, g = torch.randn((n, kin), generator=gen), torch.randn((n, out_f), generator=gen)

    def run(dev, dtype):
        mod = torch.nn.Linear(kin, out_f).to(device=dev, dtype=dtype)
        mod.load_state_dict(lin.state_dict())
        xx = x.to(device=dev, dtype=dtype).requires_grad_(True)
        y = linear(xx, mod) if dev == "cuda" else mod(xx)
        (y * g.to(device=dev, dtype=dtype)).sum().backward()
        out = {"y": y, "dx": xx.grad, "dW": mod.weight.grad, "db": mod.bias.grad}
        return {k: t.detach().double().cpu() for k, t in out.items()}

    got, want = run("cuda", torch.float32), run("cpu", torch.float64)
    for k in want:
        scale = float(want[k].abs().max()) + 1e-30
        assert float((got[k] - want[k]).abs().max()) < 3e-5 * scale, k


def test_small_batches_stay_on_one_stream():
    """overlap_streams is a request: below overlap_min_messages tree messages the encoders run on the caller's stream
    (the forks and joins cost more host time than a step of that size gains), and the results are those of the
    two-stream step to rounding (the short products' tiling follows the stream layout)."""
    g, x_batch = _seeded_training_problem(79, 16, 20.0)
    hip = make_hip_model(g, "cuda")
    hip.overlap_streams = True
    _, jt, mp, _ = x_batch()
    assert int(jt[3].shape[0]) < hip.overlap_min_messages
    with torch.no_grad():
        a = hip.encode_latent(jt, mp)
    assert hip.jtnn.side_streams == ()  # one stream: no helper stream was handed to the tree encoder
    hip.overlap_min_messages = 0
    with torch.no_grad():
        b = hip.encode_latent(jt, mp)
    assert len(hip.jtnn.side_streams) == 1
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) < 1e-5 * float(y.abs().max())
