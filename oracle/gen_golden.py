#!/usr/bin/env python
"""Generate tests/golden/*.pt by running the REFERENCE's own hot-path files.

TEST INFRASTRUCTURE (see oracle/gps_oracle.py header).  Run in the build container only:
it needs /root/reference, which does not exist on the GPU box.

    python oracle/gen_golden.py            # rewrites tests/golden/*.pt

What executes: ``/root/reference/graphgps/layer/{gps_layer,gatedgcn_layer,gine_conv_layer,
performer_layer,graphormer_layer}.py`` and ``graphgps/encoder/graphormer_encoder.py`` imported unmodified from where they lie, hosted on the stand-in modules
in ``oracle/ref_stubs`` for the third-party packages that are absent here.  So the in-tree
arithmetic of the fixtures is the reference's; PyG / torch_scatter semantics are the stubs'
restatement (see oracle/ref_stubs/README.md).

Each fixture holds: ctor kwargs, state_dict (before the step), inputs, loss weights, the
train-mode outputs, input/parameter gradients, the state_dict after the step (BN running
stats) and an eval-mode forward taken afterwards.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GPS_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)

import graphgps_amd.graphgym.register  # noqa: E402,F401  (before the stubs go on sys.path)
from graphgps_amd.graphgym.config import cfg, set_cfg  # noqa: E402
from graphgps_amd.synthetic import make_structure  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
_pkg = types.ModuleType("graphgps")
_pkg.__path__ = [os.path.join(REF, "graphgps")]   # skip graphgps/__init__.py (imports ogb, wandb, ...)
sys.modules["graphgps"] = _pkg
set_cfg(cfg)
# the reference's files register their GraphGym wrappers under the same names as this package's
for _k in ("gatedgcnconv", "gineconv"):
    graphgps_amd.graphgym.register.layer_dict.pop(_k, None)
for _k in ("GraphormerBias", "SignNet", "LapPE"):
    graphgps_amd.graphgym.register.node_encoder_dict.pop(_k, None)
from graphgps.layer.gps_layer import GPSLayer as RefGPSLayer  # noqa: E402
from torch_geometric.data import Batch as StubBatch  # noqa: E402

CASES = {
    # name: (ctor kwargs, profile, num_graphs, seed)
    "gatedgcn_transformer_d32h4": (dict(dim_h=32, local_gnn_type="CustomGatedGCN",
                                        global_model_type="Transformer", num_heads=4), "P14", 6, 11),
    "gatedgcn_transformer_d48h2": (dict(dim_h=48, local_gnn_type="CustomGatedGCN",
                                        global_model_type="Transformer", num_heads=2), "P30", 5, 12),
    "gine_transformer_d32h2": (dict(dim_h=32, local_gnn_type="GINE",
                                    global_model_type="Transformer", num_heads=2), "ZINC", 4, 13),
    "gatedgcn_performer_d32h2": (dict(dim_h=32, local_gnn_type="CustomGatedGCN",
                                      global_model_type="Performer", num_heads=2), "P30", 4, 14),
    "gatedgcn_only_d16": (dict(dim_h=16, local_gnn_type="CustomGatedGCN",
                               global_model_type="None", num_heads=1), "P14", 5, 15),
    # EquivStableLapPE variants (gatedgcn_layer.py:28-35,101-104; gine_conv_layer.py:11-87): the layer
    # additionally reads batch.pe_EquivStableLapPE [N, d]
    "gatedgcn_eslappe_transformer_d32h4": (dict(dim_h=32, local_gnn_type="CustomGatedGCN",
                                                global_model_type="Transformer", num_heads=4,
                                                equivstable_pe=True), "P14", 6, 16),
    # BiasedTransformer (gps_layer.py:201-203): the layer additionally reads a dense batch.attn_bias
    # [B*H, nmax, nmax] (random here; graphormer_encoder.py produces it in a model)
    "gine_biasedtransformer_d32h4": (dict(dim_h=32, local_gnn_type="GINE",
                                          global_model_type="BiasedTransformer", num_heads=4), "ZINC", 5, 22),
    # GCN local model (gps_layer.py:50-52,183: called without edge attributes; PyG GCNConv semantics are the
    # stub's restatement, the wiring is the reference's)
    "gcn_transformer_d32h4": (dict(dim_h=32, local_gnn_type="GCN", global_model_type="Transformer",
                                   num_heads=4), "P14", 6, 23),
    # gt.layer_norm=True (gps_layer.py:129-134,148,191-192,209-210,226-227): PyG's graph-mode LayerNorm in place of the three
    # BatchNorms (no configs/**/*.yaml sets it; the branch exists in the reference, so it exists here)
    "gine_transformer_layernorm_d32h4": (dict(dim_h=32, local_gnn_type="GINE", global_model_type="Transformer",
                                              num_heads=4, layer_norm=True, batch_norm=False), "P14", 6, 31),
    "gatedgcn_transformer_layernorm_d32h4": (dict(dim_h=32, local_gnn_type="CustomGatedGCN",
                                                  global_model_type="Transformer", num_heads=4, layer_norm=True,
                                                  batch_norm=False), "ZINC", 5, 32),
    # (no GINE + equivstable_pe fixture: the reference's GINEConvESLapPE cannot be constructed -- its
    #  __init__ calls reset_parameters(), which touches self.mlp_r_ij, before defining it:
    #  gine_conv_layer.py:35 vs :43-54.  The HIP layer implements the intended arithmetic and is
    #  checked against the oracle restatement only.)
}


def run_case(name, kw, profile, num_graphs, seed):
    torch.manual_seed(seed)
    kw = dict(kw)
    norms = dict(layer_norm=kw.pop("layer_norm", False), batch_norm=kw.pop("batch_norm", True))
    layer = RefGPSLayer(**kw, act="relu", dropout=0.0, attn_dropout=0.0, **norms)
    layer.train()
    # non-trivial BN affine parameters so that their gradients are exercised
    with torch.no_grad():
        for mod in layer.modules():
            if isinstance(mod, torch.nn.BatchNorm1d) or type(mod).__name__ == "LayerNorm":
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.3, 0.3)
    sd0 = {k: v.clone() for k, v in layer.state_dict().items()}
    sizes, edge_index, bvec, ptr, gen, _ = make_structure(profile, num_graphs, seed)
    N, E, d = int(ptr[-1]), edge_index.shape[1], kw["dim_h"]
    x = torch.randn(N, d, generator=gen, requires_grad=True)
    e = torch.randn(E, d, generator=gen, requires_grad=True)
    wx = torch.randn(N, d, generator=gen)
    we = torch.randn(E, d, generator=gen)
    batch = StubBatch(x=x, edge_index=edge_index, edge_attr=e, batch=bvec)
    pe = None
    if kw.get("equivstable_pe"):
        pe = (torch.randn(N, d, generator=gen) * 0.5).requires_grad_(True)
        batch.pe_EquivStableLapPE = pe
    bias = None
    if kw["global_model_type"] == "BiasedTransformer":
        nmax = int((ptr[1:] - ptr[:-1]).max())
        bias = torch.randn(num_graphs * kw["num_heads"], nmax, nmax, generator=gen, requires_grad=True)
        batch.attn_bias = bias
    out = layer(batch)
    loss = (out.x * wx).sum() + (out.edge_attr * we).sum()
    loss.backward()
    fix = dict(
        name=name, ctor=dict(kw, act="relu", dropout=0.0, attn_dropout=0.0, **norms),
        state_dict=sd0, x=x.detach().clone(), edge_attr=e.detach().clone(),
        edge_index=edge_index, batch=bvec, ptr=ptr, wx=wx, we=we,
        out_x=out.x.detach().clone(), out_edge_attr=out.edge_attr.detach().clone(),
        grad_x=x.grad.clone(), grad_edge_attr=e.grad.clone(),
        param_grads={k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None},
        state_dict_after=({k: v.clone() for k, v in layer.state_dict().items()}),
    )
    if pe is not None:
        fix["pe"] = pe.detach().clone()
        fix["grad_pe"] = pe.grad.clone()
    if bias is not None:
        fix["attn_bias"] = bias.detach().clone()
        fix["grad_attn_bias"] = bias.grad.clone()
    layer.eval()
    with torch.no_grad():
        eb = StubBatch(x=x.detach(), edge_index=edge_index, edge_attr=e.detach(), batch=bvec)
        if pe is not None:
            eb.pe_EquivStableLapPE = pe.detach()
        if bias is not None:
            eb.attn_bias = bias.detach()
        ob = layer(eb)
    fix["eval_out_x"] = ob.x.clone()
    fix["eval_out_edge_attr"] = ob.edge_attr.clone()
    return fix


def _graphormer_graphs(seed, sizes, num_edge_types=4, num_node_types=28):
    """Small directed test graphs: an undirected random tree + a few extra (some one-way) edges, one isolated
    node in the last graph, duplicate edge columns in the first -- the cases the pre-processing must agree on
    (multi-path ties, unreachable pairs, clipping at `distance`)."""
    g = torch.Generator().manual_seed(seed)
    graphs = []
    for gi_, n in enumerate(sizes):
        src, dst = [], []
        last = n - 1 if gi_ == len(sizes) - 1 and n > 2 else n      # leave the last node isolated
        for v in range(1, last):
            u = int(torch.randint(0, v, (1,), generator=g))
            src += [u, v]
            dst += [v, u]
        for _ in range(max(1, n // 3)):
            u, v = (int(t) for t in torch.randint(0, last, (2,), generator=g))
            if u != v:
                src.append(u)
                dst.append(v)
        if gi_ == 0 and src:
            src.append(src[0])
            dst.append(dst[0])
        ei = torch.tensor([src, dst], dtype=torch.long)
        perm = torch.randperm(ei.shape[1], generator=g)
        ei = ei[:, perm]
        graphs.append(dict(x=torch.randint(0, num_node_types, (n, 1), generator=g), edge_index=ei,
                           edge_attr=torch.randint(0, num_edge_types, (ei.shape[1],), generator=g)))
    return graphs


def run_graphormer(seed=21):
    """The reference's Graphormer pieces, unmodified: graphormer_pre_processing (networkx shortest paths),
    BiasEncoder + NodeEncoder (graphormer_encoder.py), GraphormerLayer (graphormer_layer.py) and the
    BiasedTransformer branch of GPSLayer (gps_layer.py:201-203)."""
    from graphgps.encoder.graphormer_encoder import (BiasEncoder, NodeEncoder,
                                                     graphormer_pre_processing)
    from graphgps.layer.graphormer_layer import GraphormerLayer as RefGraphormerLayer
    from graphgps_amd.data import Batch as MyBatch
    torch.manual_seed(seed)
    H, D, dist = 4, 40, 5
    cfg.posenc_GraphormerBias.num_in_degrees = 16
    cfg.posenc_GraphormerBias.num_out_degrees = 16
    cfg.posenc_GraphormerBias.node_degrees_only = False
    out = {}
    for token in (False, True):
        graphs = _graphormer_graphs(seed, [7, 12, 5, 9])
        pre = []
        for gr in graphs:
            d = StubBatch(**{k: v.clone() for k, v in gr.items()})
            d.num_nodes = gr["x"].shape[0]
            d = graphormer_pre_processing(d, dist)
            pre.append({k: getattr(d, k) for k in ("x", "edge_index", "edge_attr", "in_degrees", "out_degrees",
                                                   "spatial_types", "graph_index", "shortest_path_types")})
        b = MyBatch.from_graph_list(pre)
        data = StubBatch(**{k: getattr(b, k) for k in b.keys()})
        bias_enc = BiasEncoder(H, dist, 4, use_graph_token=token)
        node_enc = NodeEncoder(D, 16, 16, input_dropout=0.0, use_graph_token=token)
        layer = RefGraphormerLayer(D, H, dropout=0.0, attention_dropout=0.0, mlp_dropout=0.0)
        layer.train()
        sd = {"bias": {k: v.clone() for k, v in bias_enc.state_dict().items()},
              "node": {k: v.clone() for k, v in node_enc.state_dict().items()},
              "layer": {k: v.clone() for k, v in layer.state_dict().items()}}
        data.x = torch.randn(b.x.shape[0], D)
        x0 = data.x.clone()
        data = bias_enc(data)
        # add_graph_token (graphormer_encoder.py:199-203) groups the token rows with their graphs through
        # ``torch.sort(data.batch)`` and relies on equal keys keeping their order -- which torch.sort only
        # guarantees with stable=True (on this CPU build the default already scrambles the rows of a
        # 13-node graph, after which the reference's attn_bias no longer lines up with its own rows and the
        # token is not the first row ``graph_token`` pooling reads).  The fixture records the intended
        # behaviour: the reference's code, run with a stable sort.
        _sort = torch.sort
        torch.sort = lambda t, *a, **k: _sort(t, *a, **dict(k, stable=True))
        try:
            data = node_enc(data)
        finally:
            torch.sort = _sort
        attn_bias, x_enc = data.attn_bias, data.x
        attn_bias.retain_grad()
        data = layer(data)
        w = torch.randn(data.x.shape)
        (data.x * w).sum().backward()
        key = "token" if token else "plain"
        out[key] = dict(
            graphs=graphs, pre=pre, H=H, D=D, dist=dist, state=sd, x0=x0, w=w,
            attn_bias=attn_bias.detach().clone(), x_enc=x_enc.detach().clone(), batch_after=data.batch.clone(),
            out_x=data.x.detach().clone(), grad_attn_bias=attn_bias.grad.clone(),
            grads={"bias": {k: p.grad.clone() for k, p in bias_enc.named_parameters() if p.grad is not None},
                   "node": {k: p.grad.clone() for k, p in node_enc.named_parameters() if p.grad is not None},
                   "layer": {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None}})
    return out


def run_signnet(seed=31):
    """The reference's SignNetNodeEncoder (signnet_pos_encoder.py), both rho models, on a small batch with
    NaN-padded eigenvectors: state_dict, encoded x, parameter gradients."""
    from graphgps.encoder.signnet_pos_encoder import SignNetNodeEncoder as RefSignNet
    out = {}
    for model, k in (("MLP", 5), ("DeepSet", 9)):
        torch.manual_seed(seed)
        cfg.share.dim_in = 7
        cfg.posenc_SignNet.model = model
        cfg.posenc_SignNet.dim_pe = 6
        cfg.posenc_SignNet.layers = 3
        cfg.posenc_SignNet.post_layers = 2
        cfg.posenc_SignNet.phi_hidden_dim = 16
        cfg.posenc_SignNet.phi_out_dim = 4
        cfg.posenc_SignNet.eigen.max_freqs = k
        cfg.posenc_SignNet.pass_as_var = False
        # float64: the gradient of the first phi layer (in_channels = 1, followed by BatchNorm, summed over
        # the +v / -v branches) is a heavily cancelling sum -- in float32 the reference's own value for it is
        # only good to ~2e-3 relative, which is no yardstick for a 1e-5 comparison
        enc = RefSignNet(20).double()
        enc.train()
        sizes, edge_index, bvec, ptr, gen, _ = make_structure("P14", 5, seed)
        N = int(ptr[-1])
        x = torch.randn(N, 7, generator=gen).double()
        vecs = torch.randn(N, k, generator=gen).double()
        if model == "DeepSet":                 # graphs with fewer than k nodes pad their eigenvectors with NaN
            n_of = (ptr[1:] - ptr[:-1])[bvec]
            vecs[torch.arange(k)[None, :] >= n_of[:, None]] = float("nan")
        w = torch.randn(N, 20, generator=gen).double()
        sd = {kk: v.clone() for kk, v in enc.state_dict().items()}
        b = StubBatch(x=x.clone(), edge_index=edge_index, batch=bvec, eigvecs_sn=vecs.clone(),
                      eigvals_sn=torch.zeros(N, k, 1, dtype=torch.float64))
        o = enc(b)
        (o.x * w).sum().backward()
        out[model] = dict(k=k, state_dict=sd, x=x, edge_index=edge_index, batch=bvec, ptr=ptr, eigvecs=vecs, w=w,
                          out_x=o.x.detach().clone(),
                          grads={kk: p.grad.clone() for kk, p in enc.named_parameters() if p.grad is not None},
                          state_dict_after={kk: v.clone() for kk, v in enc.state_dict().items()})
    return out


def run_san(seed=41):
    """The reference's SANLayer / SAN2Layer (san_layer.py, san2_layer.py; full_graph=True: real edges + the
    complement pairs of graphgps/utils.py:negate_edge_index), one layer each on a small batch."""
    from graphgps.layer.san_layer import SANLayer as RefSAN
    from graphgps.layer.san2_layer import SAN2Layer as RefSAN2

    class _B(StubBatch):
        def size(self, dim=None):
            return self.x.shape[0] if dim == 0 else (self.x.shape[0], self.x.shape[0])

    out = {}
    for name, cls in (("SANLayer", RefSAN), ("SAN2Layer", RefSAN2)):
        torch.manual_seed(seed)
        d, H = 32, 4
        fake = torch.nn.Embedding(1, d)
        layer = cls(gamma=0.1, in_dim=d, out_dim=d, num_heads=H, full_graph=True, fake_edge_emb=fake,
                    dropout=0.0, layer_norm=False, batch_norm=True, residual=True)
        layer.train()
        sd = {k: v.clone() for k, v in layer.state_dict().items()}
        sizes, edge_index, bvec, ptr, gen, _ = make_structure("P14", 5, seed)
        N, E = int(ptr[-1]), edge_index.shape[1]
        x = torch.randn(N, d, generator=gen, requires_grad=True)
        e = torch.randn(E, d, generator=gen, requires_grad=True)
        w = torch.randn(N, d, generator=gen)
        b = _B(x=x, edge_index=edge_index, edge_attr=e, batch=bvec)
        o = layer(b)
        (o.x * w).sum().backward()
        out[name] = dict(d=d, H=H, gamma=0.1, state_dict=sd, x=x.detach().clone(), edge_attr=e.detach().clone(),
                         edge_index=edge_index, batch=bvec, ptr=ptr, w=w, out_x=o.x.detach().clone(),
                         grad_x=x.grad.clone(), grad_edge_attr=e.grad.clone(),
                         grads={k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None})
    return out


def run_lappe(seed=51):
    """The reference's LapPENodeEncoder (laplace_pos_encoder.py), DeepSet and Transformer models, training mode
    (the random sign flip draws ``torch.rand(k)`` right after ``torch.manual_seed(seed + 1)``), NaN-padded
    frequencies, BatchNorm on the raw PE, a post-MLP."""
    from graphgps.encoder.laplace_pos_encoder import LapPENodeEncoder as RefLapPE
    out = {}
    for model, layers, post, norm in (("DeepSet", 3, 2, "BatchNorm"), ("DeepSet", 1, 0, "none"),
                                      ("Transformer", 2, 1, "none")):
        torch.manual_seed(seed)
        cfg.share.dim_in = 6
        pe = cfg.posenc_LapPE
        pe.model, pe.dim_pe, pe.layers, pe.post_layers, pe.n_heads = model, 8, layers, post, 2
        pe.raw_norm_type, pe.pass_as_var = norm, False
        pe.eigen.max_freqs = 5
        enc = RefLapPE(24)
        enc.train()
        gen = torch.Generator().manual_seed(seed)
        N, k = 40, 5
        x = torch.randn(N, 6, generator=gen)
        vecs = torch.randn(N, k, generator=gen)
        vals = torch.randn(N, k, 1, generator=gen)
        vecs[:6, 3:] = float("nan")
        vals[:6, 3:] = float("nan")
        w = torch.randn(N, 24, generator=gen)
        sd = {kk: v.clone() for kk, v in enc.state_dict().items()}
        b = StubBatch(x=x.clone(), EigVals=vals.clone(), EigVecs=vecs.clone())
        torch.manual_seed(seed + 1)
        o = enc(b)
        (o.x * w).sum().backward()
        out[f"{model}-{layers}-{post}-{norm}"] = dict(
            model=model, layers=layers, post=post, norm=norm, state_dict=sd, x=x, EigVals=vals, EigVecs=vecs,
            w=w, seed=seed + 1, out_x=o.x.detach().clone(),
            grads={kk: p.grad.clone() for kk, p in enc.named_parameters() if p.grad is not None})
    return out


def run_custom_gnn_layers(seed=61):
    """The layer classes custom_gnn stacks (network/custom_gnn.py:44-50), from the reference's own files:
    ``GatedGCNLayer.forward(batch)`` (gatedgcn_layer.py:45-88) and ``GINEConvLayer`` (gine_conv_layer.py:90-116),
    with and without the residual."""
    from graphgps.layer.gatedgcn_layer import GatedGCNLayer as RefGated
    from graphgps.layer.gine_conv_layer import GINEConvLayer as RefGINE
    out = {}
    for kind, cls in (("gatedgcn", RefGated), ("gine", RefGINE)):
        for residual in (True, False):
            torch.manual_seed(seed)
            d = 24
            layer = cls(d, d, dropout=0.0, residual=residual)
            layer.train()
            sd = {k: v.clone() for k, v in layer.state_dict().items()}
            sizes, edge_index, bvec, ptr, gen, _ = make_structure("ZINC", 4, seed)
            N, E = int(ptr[-1]), edge_index.shape[1]
            x = torch.randn(N, d, generator=gen, requires_grad=True)
            e = torch.randn(E, d, generator=gen, requires_grad=True)
            wx, we = torch.randn(N, d, generator=gen), torch.randn(E, d, generator=gen)
            b = StubBatch(x=x, edge_index=edge_index, edge_attr=e, batch=bvec)
            o = layer(b)
            ((o.x * wx).sum() + (o.edge_attr * we).sum()).backward()
            out[f"{kind}-{'res' if residual else 'nores'}"] = dict(
                kind=kind, residual=residual, d=d, state_dict=sd, x=x.detach().clone(),
                edge_attr=e.detach().clone(), edge_index=edge_index, batch=bvec, ptr=ptr, wx=wx, we=we,
                out_x=o.x.detach().clone(), out_edge_attr=o.edge_attr.detach().clone(), grad_x=x.grad.clone(),
                grad_edge_attr=e.grad.clone(),
                grads={k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None})
    return out


def run_aux_modules(seed=71):
    """Encoders and heads either side of the layers, from the reference's own files, forward only (they are
    plain torch on both sides; the point is the parameter names -- strict state_dict load -- and the arithmetic):
    TypeDictNode/Edge, ASTNode/ASTEdge, RWSE (linear and 3-layer MLP, BatchNorm on the raw statistics),
    EquivStableLapPE, heads san_graph / ogb_code_graph / graphormer_graph (+ graph_token pooling) /
    inductive_node."""
    import importlib
    reg = graphgps_amd.graphgym.register
    for d_, keys in ((reg.node_encoder_dict, ("TypeDictNode", "ASTNode", "RWSE", "HKdiagSE", "ElstaticSE",
                                              "EquivStableLapPE")),
                     (reg.edge_encoder_dict, ("TypeDictEdge", "ASTEdge")),
                     (reg.head_dict, ("san_graph", "ogb_code_graph", "graphormer_graph", "inductive_node")),
                     (reg.pooling_dict, ("graph_token",))):
        for k in keys:
            d_.pop(k, None)
    enc = {m: importlib.import_module(f"graphgps.encoder.{m}") for m in
           ("type_dict_encoder", "ast_encoder", "kernel_pos_encoder", "equivstable_laplace_pos_encoder")}
    head = {m: importlib.import_module(f"graphgps.head.{m}") for m in
            ("san_graph", "ogb_code_graph", "graphormer_graph", "inductive_node")}
    importlib.import_module("graphgps.pooling.graph_token")
    gen = torch.Generator().manual_seed(seed)
    sizes, edge_index, bvec, ptr, _, _ = make_structure("ZINC", 4, seed)
    N, E = int(ptr[-1]), edge_index.shape[1]
    out = {}

    def record(name, module, inputs, outputs):
        module.eval()
        b = StubBatch(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in inputs.items()})
        res = module(b)
        vals = outputs(b, res)
        out[name] = dict(state_dict={k: v.clone() for k, v in module.state_dict().items()}, inputs=inputs,
                         outputs=[v.detach().clone() for v in vals])

    cfg.dataset.node_encoder_num_types, cfg.dataset.edge_encoder_num_types = 28, 4
    torch.manual_seed(seed)
    record("TypeDictNode", enc["type_dict_encoder"].TypeDictNodeEncoder(12),
           dict(x=torch.randint(0, 28, (N, 1), generator=gen)), lambda b, r: [b.x])
    record("TypeDictEdge", enc["type_dict_encoder"].TypeDictEdgeEncoder(12),
           dict(edge_attr=torch.randint(0, 4, (E,), generator=gen)), lambda b, r: [b.edge_attr])
    record("ASTNode", enc["ast_encoder"].ASTNodeEncoder(12),
           dict(x=torch.stack([torch.randint(0, 98, (N,), generator=gen),
                               torch.randint(0, 10030, (N,), generator=gen)], 1),
                node_depth=torch.randint(0, 30, (N, 1), generator=gen)), lambda b, r: [b.x])
    record("ASTEdge", enc["ast_encoder"].ASTEdgeEncoder(12),
           dict(edge_attr=torch.randint(0, 2, (E, 2), generator=gen)), lambda b, r: [b.edge_attr])
    cfg.share.dim_in = 5
    for model, layers in (("Linear", 1), ("mlp", 3)):
        pe = cfg.posenc_RWSE
        pe.model, pe.layers, pe.dim_pe, pe.raw_norm_type, pe.pass_as_var = model, layers, 8, "BatchNorm", True
        pe.kernel.times = list(range(1, 13))
        torch.manual_seed(seed)
        record(f"RWSE-{model}", enc["kernel_pos_encoder"].RWSENodeEncoder(20),
               dict(x=torch.randn(N, 5, generator=gen), pestat_RWSE=torch.rand(N, 12, generator=gen)),
               lambda b, r: [b.x, b.pe_RWSE])
    pe = cfg.posenc_EquivStableLapPE
    pe.eigen.max_freqs, pe.raw_norm_type = 6, "BatchNorm"
    vecs = torch.randn(N, 6, generator=gen)
    vecs[:3, 4:] = float("nan")
    torch.manual_seed(seed)
    record("EquivStableLapPE", enc["equivstable_laplace_pos_encoder"].EquivStableLapPENodeEncoder(16),
           dict(x=torch.randn(N, 16, generator=gen), EigVals=torch.zeros(N, 6, 1), EigVecs=vecs),
           lambda b, r: [b.pe_EquivStableLapPE])
    hx = torch.randn(N, 16, generator=gen)
    cfg.gnn.act = "relu"
    for pool in ("add", "mean"):
        cfg.model.graph_pooling = pool
        torch.manual_seed(seed)
        record(f"san_graph-{pool}", head["san_graph"].SANGraphHead(16, 3),
               dict(x=hx, batch=bvec, y=torch.zeros(4)), lambda b, r: [r[0]])
    cfg.model.graph_pooling = "mean"
    torch.manual_seed(seed)
    record("ogb_code_graph", head["ogb_code_graph"].OGBCodeGraphHead(16, 5002),
           dict(x=hx, batch=bvec, y=torch.zeros(4), y_arr=torch.zeros(4, 5)), lambda b, r: list(r[0]))
    cfg.model.graph_pooling = "graph_token"
    torch.manual_seed(seed)
    record("graphormer_graph", head["graphormer_graph"].GraphormerHead(16, 2),
           dict(x=hx, batch=bvec, y=torch.zeros(4)), lambda b, r: [r[0]])
    cfg.gnn.layers_post_mp = 2
    torch.manual_seed(seed)
    record("inductive_node", head["inductive_node"].GNNInductiveNodeHead(16, 3),
           dict(x=hx, batch=bvec, y=torch.zeros(N)), lambda b, r: [r[0]])
    # registered losses (graphgps/loss/*.py), forward values on fixed logits / targets
    for k in ("l1_losses", "subtoken_cross_entropy", "weighted_cross_entropy", "multilabel_cross_entropy"):
        reg.loss_dict.pop(k, None)
    losses = {m: importlib.import_module(f"graphgps.loss.{m}") for m in
              ("l1", "subtoken_prediction_loss", "weighted_cross_entropy", "multilabel_classification_loss")}
    logits = torch.randn(50, 6, generator=gen)
    target = torch.randint(0, 5, (50,), generator=gen)            # class 5 never occurs
    blogits, btarget = torch.randn(50, generator=gen), torch.randint(0, 2, (50,), generator=gen)
    mlogits = torch.randn(20, 7, generator=gen)
    mtarget = torch.randint(0, 2, (20, 7), generator=gen).float()
    mtarget[torch.rand(20, 7, generator=gen) < 0.2] = float("nan")
    cfg.model.loss_fun = "weighted_cross_entropy"
    wl, wp = losses["weighted_cross_entropy"].weighted_cross_entropy(logits, target)
    bl, bp = losses["weighted_cross_entropy"].weighted_cross_entropy(blogits, btarget)
    cfg.model.loss_fun, cfg.dataset.task_type = "cross_entropy", "classification_multilabel"
    ml, mp = losses["multilabel_classification_loss"].multilabel_cross_entropy(mlogits, mtarget)
    cfg.dataset.task_type = "regression"
    cfg.model.loss_fun = "smoothl1"
    sl, _ = losses["l1"].l1_losses(blogits, btarget.float())
    out["_losses"] = dict(logits=logits, target=target, blogits=blogits, btarget=btarget, mlogits=mlogits,
                          mtarget=mtarget, weighted_multiclass=(wl.clone(), wp.clone()),
                          weighted_binary=(bl.clone(), bp.clone()), multilabel=(ml.clone(), mp.clone()),
                          smoothl1=sl.clone())
    # LR schedules (graphgps/optimizer/extra_optimizers.py:92-225): the learning rate after each of 40 epochs
    for k in ("adagrad", "adamW"):
        reg.optimizer_dict.pop(k, None)
    for k in ("plateau", "reduce_on_plateau", "linear_with_warmup", "cosine_with_warmup", "polynomial_with_warmup"):
        reg.scheduler_dict.pop(k, None)
    _gg = types.ModuleType("torch_geometric.graphgym.optim")

    class SchedulerConfig:            # the dataclass base the reference extends (fields unused here)
        pass
    _gg.SchedulerConfig = SchedulerConfig
    sys.modules["torch_geometric.graphgym.optim"] = _gg
    xo = importlib.import_module("graphgps.optimizer.extra_optimizers")
    sched = {}
    for name, fn in (("linear_with_warmup", xo.linear_with_warmup_scheduler),
                     ("cosine_with_warmup", xo.cosine_with_warmup_scheduler),
                     ("polynomial_with_warmup", xo.polynomial_with_warmup_scheduler)):
        for warm, total in ((5, 30), (0, 12)):
            p0 = torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.SGD([p0], lr=0.01)
            sc = fn(opt, warm, total)
            lrs = [sc.get_last_lr()[0]]
            for _ in range(40):
                opt.step()
                sc.step()
                lrs.append(sc.get_last_lr()[0])
            sched[f"{name}-{warm}-{total}"] = lrs
    out["_schedules"] = sched
    out["_meta"] = dict(N=N, E=E, num_graphs=4, ptr=ptr)
    return out


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    fix = run_custom_gnn_layers()
    torch.save(fix, os.path.join(outdir, "custom_gnn_layers.pt"))
    print("custom_gnn_layers:", sorted(fix))
    fix = run_lappe()
    torch.save(fix, os.path.join(outdir, "lappe_encoder.pt"))
    print("lappe_encoder:", {k: tuple(v["out_x"].shape) for k, v in fix.items()})
    fix = run_san()
    torch.save(fix, os.path.join(outdir, "san_layers.pt"))
    print("san_layers:", {k: tuple(v["out_x"].shape) for k, v in fix.items()})
    fix = run_signnet()
    torch.save(fix, os.path.join(outdir, "signnet_encoder.pt"))
    print("signnet_encoder:", {k: tuple(v["out_x"].shape) for k, v in fix.items()})
    fix = run_graphormer()
    torch.save(fix, os.path.join(outdir, "graphormer_encoder_layer.pt"))
    print("graphormer_encoder_layer:", {k: tuple(v["attn_bias"].shape) for k, v in fix.items()})
    for name, (kw, profile, nb, seed) in CASES.items():
        fix = run_case(name, kw, profile, nb, seed)
        path = os.path.join(outdir, f"{name}.pt")
        torch.save(fix, path)
        print(f"{name}: N={fix['x'].shape[0]} E={fix['edge_index'].shape[1]} "
              f"|out_x|max={fix['out_x'].abs().max():.4f} -> {os.path.getsize(path)/1024:.0f} KiB")
    fix = run_aux_modules()
    torch.save(fix, os.path.join(outdir, "aux_modules.pt"))
    print("aux_modules:", sorted(k for k in fix if not k.startswith("_")))


if __name__ == "__main__":
    main()
