"""Pin the CPU oracle against fixtures produced by the reference's own code
(oracle/gen_golden.py).  CPU-only: this is what makes the oracle trustworthy as the
checker for the HIP path."""
import pytest
import torch

from conftest import Tol, assert_close, golden_names, load_golden
from graphgps_amd.data import Batch
from oracle.gps_oracle import OracleGPSLayer


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_fixture(name):
    fix = load_golden(name)
    layer = OracleGPSLayer(**fix["ctor"])
    layer.load_state_dict(fix["state_dict"], strict=True)
    layer.train()
    x = fix["x"].clone().requires_grad_(True)
    e = fix["edge_attr"].clone().requires_grad_(True)
    b = Batch(x=x, edge_index=fix["edge_index"], edge_attr=e, batch=fix["batch"], ptr=fix["ptr"])
    pe = None
    if "pe" in fix:                       # EquivStableLapPE fixtures carry the PE input and its gradient
        pe = fix["pe"].clone().requires_grad_(True)
        b.pe_EquivStableLapPE = pe
    out = layer(b)
    ((out.x * fix["wx"]).sum() + (out.edge_attr * fix["we"]).sum()).backward()
    if pe is not None:
        assert_close(pe.grad, fix["grad_pe"], Tol.GRAD_REL, "grad pe", rel_to_max=True)
    assert_close(out.x, fix["out_x"], Tol.ACT, "out.x")
    assert_close(out.edge_attr, fix["out_edge_attr"], Tol.ACT, "out.edge_attr")
    assert_close(x.grad, fix["grad_x"], Tol.GRAD_REL, "grad x", rel_to_max=True)
    assert_close(e.grad, fix["grad_edge_attr"], Tol.GRAD_REL, "grad edge_attr", rel_to_max=True)
    got = dict(layer.named_parameters())
    assert set(fix["param_grads"]) <= set(got)
    for k, g in fix["param_grads"].items():
        assert_close(got[k].grad, g, Tol.GRAD_REL, f"grad {k}", rel_to_max=True)
    after = layer.state_dict()
    for k, v in fix["state_dict_after"].items():
        if v.dtype.is_floating_point:
            assert_close(after[k], v, Tol.ACT, f"state {k}")
        else:
            assert torch.equal(after[k], v), k
    layer.eval()
    with torch.no_grad():
        eb = Batch(x=fix["x"], edge_index=fix["edge_index"], edge_attr=fix["edge_attr"],
                   batch=fix["batch"], ptr=fix["ptr"])
        if "pe" in fix:
            eb.pe_EquivStableLapPE = fix["pe"]
        ob = layer(eb)
    assert_close(ob.x, fix["eval_out_x"], Tol.ACT, "eval out.x")
    assert_close(ob.edge_attr, fix["eval_out_edge_attr"], Tol.ACT, "eval out.edge_attr")


def test_state_dict_keys_match_reference():
    """Checkpoint interchange contract: key set and shapes identical."""
    for name in golden_names():
        fix = load_golden(name)
        layer = OracleGPSLayer(**fix["ctor"])
        mine = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
        ref = {k: tuple(v.shape) for k, v in fix["state_dict"].items()}
        assert mine == ref, name
