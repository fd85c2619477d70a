"""CPU: pin the oracle of the training-side callers (decoder teacher forcing, rsample, assembly loss,
JTNNVAE.forward, JTpropVAE.forward) to what the reference itself returned on the golden batches."""
import numpy as np
import pytest
import torch

from common import (Golden, check_grad_stats, golden_cases, golden_x_batch, make_oracle_vae, rel_err)

CASES = golden_cases()


def close(a, b, tol=1e-5):
    return abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1])
def test_oracle_forward_losses_match_reference(path):
    g = Golden(path)
    z = g.z
    m = make_oracle_vae(g)
    eps = (torch.from_numpy(z["fwd_eps_tree"]), torch.from_numpy(z["fwd_eps_mol"]))
    loss, kl, wacc, tacc, sacc = m(golden_x_batch(g), float(z["fwd_beta"]), epsilons=eps)
    t = m.last_terms
    assert close(loss.item(), z["fwd_loss"]) and close(kl, z["fwd_kl"])
    assert close(t["tree_kl"].item(), z["fwd_tree_kl"]) and close(t["mol_kl"].item(), z["fwd_mol_kl"])
    assert close(t["word_loss"].item(), z["fwd_word_loss"]) and close(t["topo_loss"].item(), z["fwd_topo_loss"])
    assert close(t["assm_loss"].item(), z["fwd_assm_loss"])
    assert np.allclose([wacc, tacc, sacc], z["fwd_accs"], atol=1e-7)
    assert rel_err(t["z_tree"].detach().numpy(), z["fwd_z_tree"]) < 1e-5
    assert rel_err(t["z_mol"].detach().numpy(), z["fwd_z_mol"]) < 1e-5
    loss.backward()
    check_grad_stats(z, "fg", m.named_parameters(), 1e-4, 1e-4)


@pytest.mark.parametrize("path", [p for p in CASES if "prop_loss" in np.load(p).files],
                         ids=lambda p: p.split("/")[-1])
def test_oracle_conditional_forward_matches_reference(path):
    g = Golden(path)
    z = g.z
    m = make_oracle_vae(g, ["denticity", "isomer"])
    eps = (torch.from_numpy(z["prop_eps_tree"]), torch.from_numpy(z["prop_eps_mol"]))
    trees, jt, mp, cj = golden_x_batch(g)
    loss, metrics = m((trees, torch.from_numpy(z["prop_table"]), jt, mp, cj), float(z["fwd_beta"]), epsilons=eps)
    assert close(loss.item(), z["prop_loss"])
    assert list(metrics.keys()) == [str(k) for k in z["prop_metric_names"]]
    for got, want in zip(metrics.values(), z["prop_metrics"]):
        assert close(got, want, 1e-4)
    loss.backward()
    check_grad_stats(z, "pg", m.named_parameters(), 1e-4, 1e-4)
