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
    bias = None
    if "attn_bias" in fix:                # BiasedTransformer fixtures carry the dense bias and its gradient
        bias = fix["attn_bias"].clone().requires_grad_(True)
        b.attn_bias = bias
    out = layer(b)
    ((out.x * fix["wx"]).sum() + (out.edge_attr * fix["we"]).sum()).backward()
    if pe is not None:
        assert_close(pe.grad, fix["grad_pe"], Tol.GRAD_REL, "grad pe", rel_to_max=True)
    if bias is not None:
        assert_close(bias.grad, fix["grad_attn_bias"], Tol.GRAD_REL, "grad attn_bias", rel_to_max=True)
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
        if "attn_bias" in fix:
            eb.attn_bias = fix["attn_bias"]
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


@pytest.mark.parametrize("variant", ["plain", "token"])
def test_graphormer_oracle_and_encoders_match_reference_fixture(variant):
    """The Graphormer fixture (reference graphormer_encoder.py + graphormer_layer.py run by
    oracle/gen_golden.py) pins, on the CPU: (1) the host pre-processing bit-exactly (degrees, spatial types,
    all-pairs index, edge types along the chosen shortest path: same path as networkx picks), (2) this
    package's BiasEncoder / NodeEncoder (plain torch, ragged evaluation + one scatter) forward and
    parameter gradients, (3) the oracle's GraphormerLayer, which is what the HIP layer is checked against."""
    from conftest import GRAPHORMER_GOLDEN
    from graphgps_amd.encoder.graphormer_encoder import (BiasEncoder, NodeEncoder,
                                                         graphormer_pre_processing)
    from graphgps_amd.graphgym.config import cfg, set_cfg
    from oracle.gps_oracle import OracleGraphormerLayer
    fix = load_golden(GRAPHORMER_GOLDEN)[variant]
    token = variant == "token"
    set_cfg(cfg)
    cfg.posenc_GraphormerBias.num_in_degrees = 16
    cfg.posenc_GraphormerBias.num_out_degrees = 16
    cfg.posenc_GraphormerBias.node_degrees_only = False
    pre = []
    for gr, want in zip(fix["graphs"], fix["pre"]):
        d = Batch(**{k: v.clone() for k, v in gr.items()})
        d = graphormer_pre_processing(d, fix["dist"])
        for k in ("in_degrees", "out_degrees", "spatial_types", "graph_index", "shortest_path_types"):
            assert torch.equal(getattr(d, k), want[k]), k
        pre.append({k: getattr(d, k) for k in want})
    H, D = fix["H"], fix["D"]
    data = Batch.from_graph_list(pre)
    data.x = fix["x0"].clone()
    bias_enc = BiasEncoder(H, fix["dist"], 4, use_graph_token=token)
    node_enc = NodeEncoder(D, 16, 16, input_dropout=0.0, use_graph_token=token)
    layer = OracleGraphormerLayer(D, H, dropout=0.0, attention_dropout=0.0, mlp_dropout=0.0).train()
    bias_enc.load_state_dict(fix["state"]["bias"], strict=True)
    node_enc.load_state_dict(fix["state"]["node"], strict=True)
    layer.load_state_dict(fix["state"]["layer"], strict=True)
    data = node_enc(bias_enc(data))
    assert_close(data.attn_bias, fix["attn_bias"], Tol.ACT, "attn_bias")
    assert_close(data.x, fix["x_enc"], Tol.ACT, "encoded x")
    assert torch.equal(data.batch, fix["batch_after"])
    assert torch.equal(data.ptr, torch.cat([torch.zeros(1, dtype=torch.long),
                                            torch.bincount(fix["batch_after"]).cumsum(0)]))
    bias = data.attn_bias
    bias.retain_grad()
    data = layer(data)
    (data.x * fix["w"]).sum().backward()
    assert_close(data.x, fix["out_x"], Tol.ACT, "layer out")
    assert_close(bias.grad, fix["grad_attn_bias"], Tol.GRAD_REL, "grad attn_bias", rel_to_max=True)
    for part, mod in (("bias", bias_enc), ("node", node_enc), ("layer", layer)):
        got = dict(mod.named_parameters())
        for k, g in fix["grads"][part].items():
            assert_close(got[k].grad, g, Tol.GRAD_REL, f"grad {part}.{k}", rel_to_max=True)


def _signnet_case(model, dev, grad_tol=Tol.GRAD_REL):
    """This package's SignNet encoder against the reference-generated fixture (reference
    signnet_pos_encoder.py hosted on the GINConv / scatter stubs): strict state_dict load = parameter-name
    contract, encoded x, every parameter gradient, BatchNorm running statistics."""
    from conftest import SIGNNET_GOLDEN
    from graphgps_amd.encoder.signnet_encoder import SignNetNodeEncoder
    from graphgps_amd.graphgym.config import cfg, set_cfg
    fix = load_golden(SIGNNET_GOLDEN)[model]
    set_cfg(cfg)
    cfg.share.dim_in = 7
    pe = cfg.posenc_SignNet
    pe.model, pe.dim_pe, pe.layers, pe.post_layers = model, 6, 3, 2
    pe.phi_hidden_dim, pe.phi_out_dim, pe.pass_as_var = 16, 4, False
    pe.eigen.max_freqs = fix["k"]
    enc = SignNetNodeEncoder(20)
    enc.load_state_dict(fix["state_dict"], strict=True)      # fixture tensors are float64 (see gen_golden.py)
    enc.float().to(dev).train()
    b = Batch(x=fix["x"].float().to(dev), edge_index=fix["edge_index"].to(dev), batch=fix["batch"].to(dev),
              ptr=fix["ptr"].to(dev), eigvecs_sn=fix["eigvecs"].float().to(dev),
              eigvals_sn=torch.zeros(fix["x"].shape[0], fix["k"], 1, device=dev))
    b.num_graphs = int(fix["ptr"].numel() - 1)
    out = enc(b)
    (out.x * fix["w"].float().to(dev)).sum().backward()
    assert_close(out.x, fix["out_x"], Tol.ACT, "encoded x")
    got = dict(enc.named_parameters())
    gs = max(float(v.abs().max()) for v in fix["grads"].values())
    for k, g in fix["grads"].items():
        a_, b_ = got[k].grad.detach().double().cpu(), g.double()
        if float(b_.abs().max()) < 1e-6 * gs:
            # a bias that feeds a BatchNorm: mathematically zero (1e-14 in the float64 fixture), float32
            # rounding residue here
            assert float(a_.abs().max()) < 1e-5 * gs, k
            continue
        assert (a_ - b_).abs().max().item() <= grad_tol * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
            f"grad {k}: {(a_ - b_).abs().max().item():.3e}"
    after = enc.state_dict()
    for k, v in fix["state_dict_after"].items():
        if v.dtype.is_floating_point:
            assert_close(after[k], v, Tol.ACT, f"state {k}")


@pytest.mark.parametrize("model", ["MLP", "DeepSet"])
def test_signnet_encoder_matches_reference_fixture_cpu(model):
    _signnet_case(model, torch.device("cpu"))


@pytest.mark.parametrize("name", ["SANLayer", "SAN2Layer"])
def test_san_layers_match_reference_fixture(name):
    """The torch-level SAN layers (graphgps_amd/layer/san_layers.py; real edges + vectorised complement pairs)
    against the reference's san_layer.py / san2_layer.py run by oracle/gen_golden.py: strict state_dict load,
    output, input / edge-feature gradients and every parameter gradient."""
    from conftest import SAN_GOLDEN
    from graphgps_amd.layer import san_layers
    fix = load_golden(SAN_GOLDEN)[name]
    d, H = fix["d"], fix["H"]
    layer = getattr(san_layers, name)(gamma=fix["gamma"], in_dim=d, out_dim=d, num_heads=H, full_graph=True,
                                      fake_edge_emb=torch.nn.Embedding(1, d), dropout=0.0, layer_norm=False,
                                      batch_norm=True, residual=True)
    layer.load_state_dict(fix["state_dict"], strict=True)
    layer.train()
    x = fix["x"].clone().requires_grad_(True)
    e = fix["edge_attr"].clone().requires_grad_(True)
    b = Batch(x=x, edge_index=fix["edge_index"], edge_attr=e, batch=fix["batch"], ptr=fix["ptr"])
    out = layer(b)
    (out.x * fix["w"]).sum().backward()
    assert_close(out.x, fix["out_x"], Tol.ACT, "out.x")
    assert_close(x.grad, fix["grad_x"], Tol.GRAD_REL, "grad x", rel_to_max=True)
    assert_close(e.grad, fix["grad_edge_attr"], Tol.GRAD_REL, "grad edge_attr", rel_to_max=True)
    got = dict(layer.named_parameters())
    assert set(fix["grads"]) <= set(got)
    gs = max(float(v.abs().max()) for v in fix["grads"].values())
    for k, g in fix["grads"].items():
        a_, b_ = got[k].grad.detach().double(), g.double()
        assert (a_ - b_).abs().max().item() <= Tol.GRAD_REL * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
            f"grad {k}: {(a_ - b_).abs().max().item():.3e}"


def test_complement_edge_index_is_the_adjacency_complement():
    """Every ordered same-graph pair that is neither an edge nor a self pair, nothing across graphs; duplicate
    and self-loop columns in the input change nothing (graphgps/utils.py:12-66)."""
    from graphgps_amd.layer.san_layers import complement_edge_index
    gen = torch.Generator().manual_seed(0)
    sizes = [5, 1, 7, 3]
    ptr = [0]
    for n in sizes:
        ptr.append(ptr[-1] + n)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    cols = []
    for g, n in enumerate(sizes):
        if n > 1:
            ei = torch.randint(0, n, (2, 2 * n), generator=gen) + ptr[g]
            cols.append(torch.cat([ei, ei[:, :2]], dim=1))
    ei = torch.cat(cols, dim=1)
    got = {(int(a), int(b)) for a, b in complement_edge_index(ei, batch).t()}
    edges = {(int(a), int(b)) for a, b in ei.t()}
    want = {(i, j) for g in range(len(sizes)) for i in range(ptr[g], ptr[g + 1]) for j in range(ptr[g], ptr[g + 1])
            if i != j and (i, j) not in edges}
    assert got == want


@pytest.mark.parametrize("case", ["DeepSet-3-2-BatchNorm", "DeepSet-1-0-none", "Transformer-2-1-none"])
def test_lappe_encoder_matches_reference_fixture(case):
    """This package's LapPE encoder against the reference's laplace_pos_encoder.py (training mode, the random
    sign flip reproduced by seeding right before the call; NaN-padded frequencies; DeepSet / Transformer; raw
    BatchNorm; post-MLP): strict state_dict load, encoded x, every parameter gradient."""
    from conftest import LAPPE_GOLDEN
    from graphgps_amd.encoder.extra_encoders import LapPENodeEncoder
    from graphgps_amd.graphgym.config import cfg, set_cfg
    fix = load_golden(LAPPE_GOLDEN)[case]
    set_cfg(cfg)
    cfg.share.dim_in = 6
    pe = cfg.posenc_LapPE
    pe.model, pe.dim_pe, pe.layers, pe.post_layers, pe.n_heads = fix["model"], 8, fix["layers"], fix["post"], 2
    pe.raw_norm_type, pe.pass_as_var = fix["norm"], False
    pe.eigen.max_freqs = 5
    enc = LapPENodeEncoder(24)
    enc.load_state_dict(fix["state_dict"], strict=True)
    enc.train()
    b = Batch(x=fix["x"].clone(), EigVals=fix["EigVals"].clone(), EigVecs=fix["EigVecs"].clone())
    torch.manual_seed(fix["seed"])
    out = enc(b)
    (out.x * fix["w"]).sum().backward()
    assert_close(out.x, fix["out_x"], Tol.ACT, "encoded x")
    got = dict(enc.named_parameters())
    gs = max(float(v.abs().max()) for v in fix["grads"].values())
    for k, g in fix["grads"].items():
        a_, b_ = got[k].grad.detach().double(), g.double()
        assert (a_ - b_).abs().max().item() <= Tol.GRAD_REL * max(float(b_.abs().max()), 0.01 * gs, 1.0), \
            f"grad {k}: {(a_ - b_).abs().max().item():.3e}"


@pytest.mark.parametrize("case", ["gatedgcn-res", "gatedgcn-nores", "gine-res", "gine-nores"])
def test_custom_gnn_layer_oracles_match_reference_fixture(case):
    """The oracle twins that ``to_oracle_model`` puts into a ``custom_gnn`` stack -- ``GatedGCNLayer.forward(batch)``
    and ``GINEConvLayer`` -- against the reference's own classes (gatedgcn_layer.py:45-88,
    gine_conv_layer.py:90-116), with and without the residual.  The HIP layers are checked against these oracles
    on the GPU (tests/test_hip_layer.py::test_custom_gnn_vs_oracle)."""
    from conftest import CUSTOM_GNN_GOLDEN
    from oracle.gps_oracle import _OracleGatedGCNBatchLayer, _OracleGINEConvLayer
    fix = load_golden(CUSTOM_GNN_GOLDEN)[case]
    d = fix["d"]
    if fix["kind"] == "gatedgcn":
        layer = _OracleGatedGCNBatchLayer(d, d, 0.0, fix["residual"])
    else:
        layer = _OracleGINEConvLayer(d, d, 0.0, fix["residual"])
    layer.load_state_dict(fix["state_dict"], strict=True)
    layer.train()
    x = fix["x"].clone().requires_grad_(True)
    e = fix["edge_attr"].clone().requires_grad_(True)
    out = layer(Batch(x=x, edge_index=fix["edge_index"], edge_attr=e, batch=fix["batch"], ptr=fix["ptr"]))
    ((out.x * fix["wx"]).sum() + (out.edge_attr * fix["we"]).sum()).backward()
    assert_close(out.x, fix["out_x"], Tol.ACT, "out.x")
    assert_close(out.edge_attr, fix["out_edge_attr"], Tol.ACT, "out.edge_attr")
    assert_close(x.grad, fix["grad_x"], Tol.GRAD_REL, "grad x", rel_to_max=True)
    assert_close(e.grad, fix["grad_edge_attr"], Tol.GRAD_REL, "grad edge_attr", rel_to_max=True)
    got = dict(layer.named_parameters())
    for k, g in fix["grads"].items():
        assert_close(got[k].grad, g, Tol.GRAD_REL, f"grad {k}", rel_to_max=True)


def _aux_cases():
    return ["TypeDictNode", "TypeDictEdge", "ASTNode", "ASTEdge", "RWSE-Linear", "RWSE-mlp", "EquivStableLapPE",
            "san_graph-add", "san_graph-mean", "ogb_code_graph", "graphormer_graph", "inductive_node"]


def aux_case_outputs(case, device="cpu"):
    """One encoder / head of ``aux_modules.pt`` on ``device``: (outputs, the reference's outputs).  On a CUDA device the
    modules take their GPU forms (multi-hot GEMM embeddings, ``ops.embedding``, the ptr-segmented pooling kernels)."""
    from conftest import AUX_GOLDEN
    import graphgps_amd  # noqa: F401  (registrations)
    from graphgps_amd.encoder import encoders as E
    from graphgps_amd.graphgym.config import cfg, set_cfg
    from graphgps_amd.head import heads as H
    allfix = load_golden(AUX_GOLDEN)
    fix, meta = allfix[case], allfix["_meta"]
    set_cfg(cfg)
    cfg.dataset.node_encoder_num_types, cfg.dataset.edge_encoder_num_types = 28, 4
    cfg.gnn.act = "relu"
    outputs = None
    if case == "TypeDictNode":
        mod, outputs = E.TypeDictNodeEncoder(12), lambda b, r: [b.x]
    elif case == "TypeDictEdge":
        mod, outputs = E.TypeDictEdgeEncoder(12), lambda b, r: [b.edge_attr]
    elif case == "ASTNode":
        mod, outputs = E.ASTNodeEncoder(12), lambda b, r: [b.x]
    elif case == "ASTEdge":
        mod, outputs = E.ASTEdgeEncoder(12), lambda b, r: [b.edge_attr]
    elif case.startswith("RWSE"):
        cfg.share.dim_in = 5
        pe = cfg.posenc_RWSE
        pe.model, pe.layers = ("Linear", 1) if case.endswith("Linear") else ("mlp", 3)
        pe.dim_pe, pe.raw_norm_type, pe.pass_as_var = 8, "BatchNorm", True
        pe.kernel.times = list(range(1, 13))
        mod, outputs = E.RWSENodeEncoder(20), lambda b, r: [b.x, b.pe_RWSE]
    elif case == "EquivStableLapPE":
        pe = cfg.posenc_EquivStableLapPE
        pe.eigen.max_freqs, pe.raw_norm_type = 6, "BatchNorm"
        mod, outputs = E.EquivStableLapPENodeEncoder(16), lambda b, r: [b.pe_EquivStableLapPE]
    elif case.startswith("san_graph"):
        cfg.model.graph_pooling = case.split("-")[1]
        mod, outputs = H.SANGraphHead(16, 3), lambda b, r: [r[0]]
    elif case == "ogb_code_graph":
        cfg.model.graph_pooling = "mean"
        mod, outputs = H.OGBCodeGraphHead(16, 5002), lambda b, r: list(r[0])
    elif case == "graphormer_graph":
        cfg.model.graph_pooling = "graph_token"
        mod, outputs = H.GraphormerHead(16, 2), lambda b, r: [r[0]]
    else:
        cfg.gnn.layers_post_mp = 2
        mod, outputs = H.GNNInductiveNodeHead(16, 3), lambda b, r: [r[0]]
    mod.load_state_dict(fix["state_dict"], strict=True)
    mod.eval().to(device)
    b = Batch(**{k: (v.clone().to(device) if torch.is_tensor(v) else v) for k, v in fix["inputs"].items()})
    if "batch" in fix["inputs"]:
        b.num_graphs = meta["num_graphs"]
    with torch.no_grad():
        res = mod(b)
    return outputs(b, res), fix["outputs"]


@pytest.mark.parametrize("case", _aux_cases())
def test_encoders_and_heads_match_reference_fixture(case):
    """The encoders / heads either side of the layers against the reference's own classes (oracle/gen_golden.py:
    run_aux_modules): strict state_dict load = the parameter-name contract, eval-mode forward = the arithmetic
    (embedding sums, depth clipping, RWSE linear / MLP with raw BatchNorm, NaN-padded eigenvectors, add / mean /
    graph_token pooling, halving MLP, 5 x vocabulary classifiers, post-MP MLP)."""
    got, want = aux_case_outputs(case)
    assert len(got) == len(want)
    for i, (a_, w_) in enumerate(zip(got, want)):
        assert_close(a_, w_, Tol.ACT, f"{case} output {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", _aux_cases())
def test_encoders_and_heads_gpu_forms_match_reference_fixture(case):
    """The same reference fixtures through the modules' GPU forms: on the device the TypeDict / AST / Atom
    encoders are multi-hot GEMMs or ``ops.embedding`` lookups and the heads pool with the ptr-segmented kernels
    (csrc/segment_pool.hip) -- different code from the CPU branches the test above pins
    (graphgps/encoder/ast_encoder.py:35-83, type_dict_encoder.py, head/san_graph.py:19-42, head/ogb_code_graph.py)."""
    got, want = aux_case_outputs(case, "cuda:0")
    assert len(got) == len(want)
    for i, (a_, w_) in enumerate(zip(got, want)):
        assert a_.is_cuda
        assert_close(a_, w_, Tol.ACT, f"{case} output {i} (GPU form)")


def test_losses_match_reference_fixture():
    """graphgps_amd/loss/losses.py against the reference's registered losses (graphgps/loss/*.py) on fixed
    logits: class-frequency weighted CE (multiclass with an absent class, binary), multilabel BCE with NaN
    targets, smooth-L1; plus GraphGym's built-in multiclass / binary cross-entropy through compute_loss."""
    from conftest import AUX_GOLDEN
    from graphgps_amd.graphgym.config import cfg, set_cfg
    from graphgps_amd.loss import losses as L
    f = load_golden(AUX_GOLDEN)["_losses"]
    set_cfg(cfg)
    cfg.model.loss_fun = "weighted_cross_entropy"
    for (logits, target, key) in ((f["logits"], f["target"], "weighted_multiclass"),
                                  (f["blogits"], f["btarget"], "weighted_binary")):
        loss, pred = L.compute_loss(logits, target)
        assert_close(loss, f[key][0], 1e-6, key)
        assert_close(pred, f[key][1], 1e-6, key + " pred")
    cfg.model.loss_fun, cfg.dataset.task_type = "cross_entropy", "classification_multilabel"
    loss, pred = L.compute_loss(f["mlogits"], f["mtarget"])
    assert_close(loss, f["multilabel"][0], 1e-6, "multilabel")
    assert torch.equal(pred, f["multilabel"][1])
    cfg.dataset.task_type = "regression"
    cfg.model.loss_fun = "smoothl1"
    assert_close(L.compute_loss(f["blogits"], f["btarget"].float())[0], f["smoothl1"], 1e-6, "smoothl1")
    # GraphGym built-ins (published behaviour): multiclass = mean NLL of log-softmax; binary = BCE with logits
    cfg.model.loss_fun, cfg.dataset.task_type = "cross_entropy", "classification"
    loss, pred = L.compute_loss(f["logits"], f["target"])
    assert_close(loss, torch.nn.functional.cross_entropy(f["logits"], f["target"]), 1e-6, "multiclass CE")
    assert_close(pred, torch.log_softmax(f["logits"], -1), 1e-6, "log-softmax")
    loss, pred = L.compute_loss(f["blogits"].unsqueeze(-1), f["btarget"].unsqueeze(-1))
    assert_close(loss, torch.nn.functional.binary_cross_entropy_with_logits(f["blogits"], f["btarget"].float()),
                 1e-6, "binary CE")
    assert_close(pred, torch.sigmoid(f["blogits"]), 1e-6, "sigmoid")


def test_lr_schedulers_match_reference_fixture():
    """graphgps_amd/schedulers.py against the reference's warm-up schedules (extra_optimizers.py:92-225): the
    learning rate after each of 40 epochs, with and without warm-up, past the end of the schedule."""
    from conftest import AUX_GOLDEN
    import graphgps_amd  # noqa: F401
    from graphgps_amd.graphgym import register
    want = load_golden(AUX_GOLDEN)["_schedules"]
    assert len(want) == 6
    for key, lrs in want.items():
        name, warm, total = key.rsplit("-", 2)
        p0 = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p0], lr=0.01)
        sc = register.scheduler_dict[name](opt, int(warm), int(total))
        got = [sc.get_last_lr()[0]]
        for _ in range(40):
            opt.step()
            sc.step()
            got.append(sc.get_last_lr()[0])
        assert len(got) == len(lrs)
        assert max(abs(a - b) for a, b in zip(got, lrs)) <= 1e-12, key
    assert {'adagrad', 'adamW'} <= set(register.optimizer_dict)
    assert {'plateau', 'reduce_on_plateau'} <= set(register.scheduler_dict)
