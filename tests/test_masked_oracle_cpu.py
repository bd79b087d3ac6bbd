"""CPU checks of the dropout-injection machinery the dropout-ON GPU parity tests rest on (tests/helpers.py): with
all-ones masks the masked oracle IS the oracle (so the stand-in attention module restates nn.MultiheadAttention +
to_dense_batch exactly), and every injected mask is consumed in the oracle's own dropout call order."""
import torch

from conftest import assert_close


def _oracle_layers(local, d, H, n):
    from oracle.gps_oracle import OracleGPSLayer
    torch.manual_seed(3)
    return [OracleGPSLayer(d, local, "Transformer", H, dropout=0.0, attn_dropout=0.0).train() for _ in range(n)]


def _run_plain(layers, b, wx, we, local):
    import copy
    bc = b.clone()
    bc.x = bc.x.clone().requires_grad_(True)
    bc.edge_attr = bc.edge_attr.clone().requires_grad_(True)
    x0, e0 = bc.x, bc.edge_attr
    for lay in [copy.deepcopy(l) for l in layers]:
        bc = lay(bc)
    loss = (bc.x * wx).sum() + ((bc.edge_attr * we).sum() if local == "CustomGatedGCN" else 0.0)
    loss.backward()
    return bc.x, bc.edge_attr, x0.grad, e0.grad


def test_masked_oracle_with_unit_masks_is_the_oracle():
    from graphgps_amd.synthetic import layer_batch
    from test_hip_layer import _masked_oracle_run
    for local, d, H, profile, nb in (("CustomGatedGCN", 48, 4, "P14", 6), ("GINE", 32, 4, "ZINC", 5)):
        layers = _oracle_layers(local, d, H, 2)
        b = layer_batch(profile, nb, d, seed=4)
        gen = torch.Generator().manual_seed(1)
        wx, we = torch.randn(b.x.shape, generator=gen), torch.randn(b.edge_attr.shape, generator=gen)
        ref = _run_plain(layers, b, wx, we, local)
        got = _masked_oracle_run(layers, b, [11, 12], H, 0.0, 0.0, wx, we, torch.float32, local)
        assert_close(got[0], ref[0], 2e-5, "x")
        assert_close(got[2], ref[2], 2e-5, "grad x", rel_to_max=True)
        if local == "CustomGatedGCN":
            assert_close(got[1], ref[1], 2e-5, "e")
            assert_close(got[3], ref[3], 2e-5, "grad e", rel_to_max=True)


def test_masks_are_consumed_in_call_order_and_change_the_result():
    from graphgps_amd.synthetic import layer_batch
    from test_hip_layer import _masked_oracle_run
    layers = _oracle_layers("CustomGatedGCN", 48, 4, 1)
    b = layer_batch("P14", 6, 48, seed=4)
    gen = torch.Generator().manual_seed(1)
    wx, we = torch.randn(b.x.shape, generator=gen), torch.randn(b.edge_attr.shape, generator=gen)
    a = _masked_oracle_run(layers, b, [5], 4, 0.25, 0.3, wx, we, torch.float64, "CustomGatedGCN")
    a2 = _masked_oracle_run(layers, b, [5], 4, 0.25, 0.3, wx, we, torch.float64, "CustomGatedGCN")
    c = _masked_oracle_run(layers, b, [6], 4, 0.25, 0.3, wx, we, torch.float64, "CustomGatedGCN")
    assert torch.equal(a[0], a2[0])                       # same seed -> same masks
    assert float((a[0] - c[0]).detach().abs().max()) > 1e-3        # another seed -> other masks
