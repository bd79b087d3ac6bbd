"""yacs-style global ``cfg`` for the GPS hot path.

The reference reads one process-global ``cfg`` (a ``yacs.config.CfgNode``
owned by ``torch_geometric.graphgym.config``; both third-party and absent from
this image) at 30 import sites, e.g. ``/root/reference/graphgps/network/gps_model.py:3,67-98``.
This module restates that surface on PyYAML:

* ``CfgNode``: attribute-style nested dict with ``merge_from_file`` /
  ``merge_from_list`` / ``clone`` and yacs' *unknown key -> KeyError* rule, so a
  typo in a YAML fails here exactly as it fails in the reference.
* ``set_cfg``: GraphGym's core defaults (PyG 2.2 ``graphgym/config.py:set_cfg``)
  followed by every ``@register_config`` extension the reference adds
  (``/root/reference/graphgps/config/*.py``), expressed as one data table.
* ``load_cfg(cfg, cfg_file, opts)``: YAML + ``key value`` CLI overrides, the
  reference's override style (``main.py:118-121``, ``tests/graph_run.sh:27``).

All 139 YAMLs under the reference's ``configs/`` and ``tests/configs/`` load
unmodified (``tests/test_config.py``).
"""
from __future__ import annotations

import copy
from ast import literal_eval
from typing import Any, Dict, Iterable, List, Optional

import yaml

from . import register


class CfgNode(dict):
    """Nested attribute dict with yacs merge semantics."""

    def __init__(self, init: Optional[Dict[str, Any]] = None):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        del self[name]

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    # -- merging ---------------------------------------------------------
    def merge_from_other_cfg(self, other: Dict[str, Any]) -> None:
        _merge(other, self, [])

    def merge_from_file(self, path: str) -> None:
        with open(path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        _merge(loaded, self, [])

    def merge_from_list(self, opts: Iterable[Any]) -> None:
        opts = list(opts)
        if len(opts) % 2 != 0:
            raise AssertionError(
                f"Override list has odd length: {opts}; it must be a list of pairs")
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent key: {full_key}")
                node = node[p]
            leaf = parts[-1]
            if leaf not in node:
                raise KeyError(f"Non-existent key: {full_key}")
            node[leaf] = _coerce(_decode(v), node[leaf], full_key)

    def to_dict(self) -> Dict[str, Any]:
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw) -> str:
        return yaml.safe_dump(self.to_dict(), **kw)


def _decode(v: Any) -> Any:
    """yacs decodes CLI strings with ``literal_eval`` and falls back to the raw string."""
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new: Any, old: Any, key: str) -> Any:
    """yacs' type-compat rule: same type, or None either side, or tuple<->list,
    or int->float; anything else is an error."""
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, CfgNode) and isinstance(new, dict):
        return CfgNode(new)
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and not isinstance(new, (dict, list, tuple)):
        # GraphGym YAMLs write e.g. ``weight_decay: 1e-5`` (a YAML *string*) for float
        # defaults and ``name: none`` for string defaults; the str->str case is covered
        # above, everything else falls through to the error.
        pass
    if isinstance(old, float) and isinstance(new, str):
        try:
            return float(new)
        except ValueError:
            pass
    raise ValueError(
        f"Type mismatch ({type(old).__name__} vs. {type(new).__name__}) with values "
        f"({old!r} vs. {new!r}) for config key: {key}")


def _merge(src: Dict[str, Any], dst: CfgNode, path: List[str]) -> None:
    for k, v in src.items():
        full = ".".join(path + [k])
        if k not in dst:
            raise KeyError(f"Non-existent config key: {full}")
        if isinstance(dst[k], CfgNode):
            if v is None:
                continue
            if not isinstance(v, dict):
                raise ValueError(f"Expected a mapping for config group: {full}")
            _merge(v, dst[k], path + [k])
        else:
            dst[k] = _coerce(_decode(v) if isinstance(v, str) else v, dst[k], full)


# ---------------------------------------------------------------------------
# Defaults.  Section A = GraphGym core (PyG 2.2 graphgym/config.py, third-party);
# section B = the reference's @register_config extensions, each tagged with the
# file it restates.
# ---------------------------------------------------------------------------
def _pe_common() -> Dict[str, Any]:
    # graphgps/config/posenc_config.py:19-46
    return dict(enable=False, model="none", dim_pe=16, layers=3, n_heads=4,
                post_layers=0, raw_norm_type="none", pass_as_var=False)


def _eigen() -> Dict[str, Any]:
    # graphgps/config/posenc_config.py:53-65
    return dict(laplacian_norm="sym", eigvec_norm="L2", max_freqs=10)


def _kernel(times_func: str = "") -> Dict[str, Any]:
    # graphgps/config/posenc_config.py:71-87
    return dict(times=[], times_func=times_func)


def _defaults() -> Dict[str, Any]:
    d: Dict[str, Any] = dict(
        # ---- A. GraphGym core ----
        print="both", accelerator="auto", devices=1, out_dir="results",
        cfg_dest="config.yaml", custom_metrics=[], seed=0, round=4,
        tensorboard_each_run=False, tensorboard_agg=True, num_workers=0,
        num_threads=6, metric_best="auto", metric_agg="argmax", view_emb=False,
        gpu_mem=False, benchmark=False,
        share=dict(dim_in=1, dim_out=1, num_splits=1),
        dataset=dict(
            name="Cora", format="PyG", dir="./datasets", task="node",
            task_type="classification", transductive=True, split=[0.8, 0.1, 0.1],
            shuffle_split=True, split_mode="random", encoder=True,
            encoder_name="db", encoder_bn=True, node_encoder=False,
            node_encoder_name="Atom", node_encoder_bn=True, edge_encoder=False,
            edge_encoder_name="Bond", edge_encoder_bn=True, encoder_dim=128,
            edge_dim=128, edge_train_mode="all", edge_message_ratio=0.8,
            edge_negative_sampling_ratio=1.0, resample_disjoint=False,
            resample_negative=False, transform="none", cache_save=False,
            cache_load=False, remove_feature=False, tu_simple=True,
            to_undirected=False, location="local", label_table="none",
            label_column="none"),
        train=dict(
            batch_size=16, sampler="full_batch", sample_node=False,
            node_per_graph=32, radius="extend", eval_period=10,
            skip_train_eval=False, ckpt_period=100, enable_ckpt=True,
            auto_resume=False, epoch_resume=-1, ckpt_clean=True,
            iter_per_epoch=32, walk_length=4, neighbor_sizes=[20, 15, 10, 5]),
        val=dict(sample_node=False, sampler="full_batch", node_per_graph=32,
                 radius="extend"),
        model=dict(type="gnn", match_upper=True, loss_fun="cross_entropy",
                   size_average="mean", thresh=0.5, edge_decoding="dot",
                   graph_pooling="add"),
        gnn=dict(
            head="default", layers_pre_mp=0, layers_mp=2, layers_post_mp=0,
            dim_inner=16, layer_type="generalconv", stage_type="stack",
            skip_every=1, batchnorm=True, act="relu", dropout=0.0, agg="add",
            normalize_adj=False, msg_direction="single", self_msg="concat",
            att_heads=1, att_final_linear=False, att_final_linear_bn=False,
            l2norm=True, keep_edge=0.5, clear_feature=True),
        optim=dict(optimizer="adam", base_lr=0.01, weight_decay=5e-4,
                   momentum=0.9, scheduler="cos", steps=[30, 60, 90],
                   lr_decay=0.1, max_epoch=200),
        bn=dict(eps=1e-5, mom=0.1),
        mem=dict(inplace=False),
    )
    # ---- B. reference extensions ----
    # defaults_config.py:17-36 (overwrite_defaults + extended_cfg)
    d["train"]["mode"] = "custom"
    d["dataset"]["name"] = "none"
    d["round"] = 5
    d["name_tag"] = ""
    d["train"]["ckpt_best"] = False
    # custom_gnn_config.py:11
    d["gnn"]["residual"] = False
    # dataset_config.py:10-19
    d["dataset"].update(node_encoder_num_types=0, edge_encoder_num_types=0,
                        slic_compactness=10, infer_link_label="None")
    # split_config.py:13-23
    d["dataset"].update(split_mode="standard", split_index=0, split_dir="./splits")
    d["run_multiple_splits"] = []
    # optimizers_config.py:10-28
    d["optim"].update(batch_accumulation=1, reduce_factor=0.1, schedule_patience=10,
                      min_lr=0.0, num_warmup_epochs=50, clip_grad_norm=False,
                      clip_grad_norm_value=1.0)
    # pretrained_config.py:10-20
    d["pretrained"] = dict(dir="", reset_prediction_head=True, freeze_main=False)
    # wandb_config.py:10-23
    d["wandb"] = dict(use=False, entity="gtransformers", project="gtblueprint", name="")
    # example.py:14-22
    d["example_arg"] = "example"
    d["example_group"] = dict(example_arg="example")
    # gt_config.py:14-72
    d["gt"] = dict(
        layer_type="SANLayer", layers=3, n_heads=8, dim_hidden=64, full_graph=True,
        gamma=1e-5, pna_degrees=[], dropout=0.0, attn_dropout=0.0, layer_norm=False,
        batch_norm=True, residual=True,
        bigbird=dict(attention_type="block_sparse", chunk_size_feed_forward=0,
                     is_decoder=False, add_cross_attention=False, hidden_act="relu",
                     max_position_embeddings=128, use_bias=False, num_random_blocks=3,
                     block_size=3, layer_norm_eps=1e-6))
    # graphormer_config.py:7-23
    d["graphormer"] = dict(num_layers=6, embed_dim=80, num_heads=4, dropout=0.0,
                           attention_dropout=0.0, mlp_dropout=0.0, input_dropout=0.0,
                           use_graph_token=True)
    d["posenc_GraphormerBias"] = dict(enable=False, node_degrees_only=False, dim_pe=0,
                                      num_spatial_types=None, num_in_degrees=None,
                                      num_out_degrees=None)
    # posenc_config.py:12-87
    for name in ("LapPE", "SignNet", "RWSE", "HKdiagSE", "ElstaticSE"):
        d[f"posenc_{name}"] = _pe_common()
    d["posenc_EquivStableLapPE"] = dict(enable=False, raw_norm_type="none")
    for name in ("LapPE", "SignNet", "EquivStableLapPE"):
        d[f"posenc_{name}"]["eigen"] = _eigen()
    d["posenc_SignNet"].update(phi_out_dim=4, phi_hidden_dim=64)
    for name in ("RWSE", "HKdiagSE"):
        d[f"posenc_{name}"]["kernel"] = _kernel()
    d["posenc_ElstaticSE"]["kernel"] = _kernel("range(10)")
    return d


def set_cfg(cfg: CfgNode) -> CfgNode:
    """Reset ``cfg`` in place to the defaults, then run user ``register_config`` hooks."""
    cfg.clear()
    for k, v in CfgNode(_defaults()).items():
        cfg[k] = v
    for fn in register.config_dict.values():  # same hook GraphGym runs
        fn(cfg)
    return cfg


def resolve_posenc_times(cfg: CfgNode) -> None:
    """``posenc_*.kernel.times_func`` is ``eval``-ed into ``kernel.times`` by the
    reference's loader (``graphgps/loader/master_loader.py:192-198``); the encoders
    read ``len(kernel.times)`` (``graphgps/encoder/kernel_pos_encoder.py:39``)."""
    for key, node in cfg.items():
        if key.startswith("posenc_") and isinstance(node, CfgNode) and "kernel" in node:
            if node.kernel.times_func:
                node.kernel.times = list(eval(node.kernel.times_func, {"range": range, "list": list}))


def load_cfg(cfg: CfgNode, cfg_file: Optional[str] = None,
             opts: Optional[Iterable[Any]] = None) -> CfgNode:
    if cfg_file:
        cfg.merge_from_file(cfg_file)
    if opts:
        cfg.merge_from_list(opts)
    resolve_posenc_times(cfg)
    assert_cfg(cfg)
    return cfg


def assert_cfg(cfg: CfgNode) -> None:
    """GraphGym's ``assert_cfg`` (PyG 2.2 ``graphgym/config.py``, third-party), which its
    ``load_cfg`` runs after merging: the rules that change what gets built."""
    if cfg.dataset.task not in ('node', 'edge', 'graph', 'link_pred'):
        raise ValueError(f"Task {cfg.dataset.task} not supported, must be one of node, edge, graph, "
                         f"link_pred")
    if 'classification' in cfg.dataset.task_type and cfg.model.loss_fun == 'mse':
        cfg.model.loss_fun = 'cross_entropy'
    elif cfg.dataset.task_type == 'regression' and cfg.model.loss_fun == 'cross_entropy':
        cfg.model.loss_fun = 'mse'
    if cfg.dataset.task == 'graph' and cfg.dataset.transductive:
        cfg.dataset.transductive = False
    if cfg.gnn.layers_post_mp < 1:
        cfg.gnn.layers_post_mp = 1
    if cfg.gnn.head == 'default':
        cfg.gnn.head = cfg.dataset.task


cfg = CfgNode()
set_cfg(cfg)
