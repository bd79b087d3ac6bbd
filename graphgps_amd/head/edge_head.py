"""The inductive edge / link-prediction head (plain torch, after the layers).

  * ``head_dict['inductive_edge']``: /root/reference/graphgps/head/inductive_edge.py:9-155 (``layer_post_mp``;
    'dot' / 'cosine_similarity' / 'concat' decoding of ``batch.edge_index_labeled``; Hits@k / MRR in eval mode).
"""
import torch
import torch.nn as nn

from ..graphgym.config import cfg
from ..graphgym.layers import MLP, new_layer_config
from ..graphgym.register import register_head


def _hits_and_mrr(pos, neg):
    """OGB link-prediction ranks: position of the positive among [positive | negatives], descending."""
    scores = torch.cat([pos.view(-1, 1), neg], dim=1)
    order = torch.argsort(scores, dim=1, descending=True)
    rank = torch.nonzero(order == 0, as_tuple=False)[:, 1] + 1
    out = {f'hits@{k}': (rank <= k).float() for k in (1, 3, 10)}
    out['mrr'] = 1.0 / rank.float()
    return out


@register_head('inductive_edge', overwrite=True)
class GNNInductiveEdgeHead(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        mode = cfg.model.edge_decoding
        if mode == 'concat':
            self.layer_post_mp = MLP(new_layer_config(dim_in * 2, dim_out, cfg.gnn.layers_post_mp,
                                                      has_act=False, has_bias=True, cfg=cfg))
            self.decode_module = lambda a, b: self.layer_post_mp(torch.cat((a, b), dim=-1))
        else:
            if dim_out > 1:
                raise ValueError('Binary edge decoding ({})is used for multi-class '
                                 'edge/link prediction.'.format(mode))
            self.layer_post_mp = MLP(new_layer_config(dim_in, dim_in, cfg.gnn.layers_post_mp,
                                                      has_act=False, has_bias=True, cfg=cfg))
            if mode == 'dot':
                self.decode_module = lambda a, b: torch.sum(a * b, dim=-1)
            elif mode == 'cosine_similarity':
                self.decode_module = nn.CosineSimilarity(dim=-1)
            else:
                raise ValueError(f'Unknown edge decoding {mode}.')

    def _apply_index(self, batch):
        return batch.x[batch.edge_index_labeled], batch.edge_label

    def forward(self, batch):
        if cfg.model.edge_decoding != 'concat':
            batch = self.layer_post_mp(batch)
        ends, label = self._apply_index(batch)
        pred = self.decode_module(ends[0], ends[1])
        if self.training:
            return pred, label
        return pred, label, self.compute_mrr(batch)

    def compute_mrr(self, batch):
        """Per graph: every positive labelled edge (u, v) ranked against (u, w) for all w != v; the batch
        statistic is the mean over graphs of the per-graph means (inductive_edge.py:62-107)."""
        if cfg.model.edge_decoding != 'dot':
            raise ValueError(f'Unsupported edge decoding {cfg.model.edge_decoding}.')
        ptr = batch.ptr.tolist()
        lab_graph = batch.batch[batch.edge_index_labeled[0]]
        per_graph = {}
        for g in range(len(ptr) - 1):
            x = batch.x[ptr[g]:ptr[g + 1]]
            sel = lab_graph == g
            pos = (batch.edge_index_labeled[:, sel] - ptr[g])[:, batch.edge_label[sel] == 1]
            scores = x @ x.t()
            pos_score = scores[pos[0], pos[1]]
            if pos.shape[1] > 0:
                keep = torch.ones(pos.shape[1], x.shape[0], dtype=torch.bool, device=x.device)
                keep[torch.arange(pos.shape[1]), pos[1]] = False
                neg = scores[pos[0]][keep].view(pos.shape[1], -1)
            else:
                neg = pos_score
            if pos.shape[1] == 0:
                vals = {k: 0.0 for k in ('hits@1', 'hits@3', 'hits@10', 'mrr')}
            else:
                vals = {k: float(v.mean().item()) for k, v in _hits_and_mrr(pos_score, neg).items()}
            for k, v in vals.items():
                per_graph.setdefault(k, []).append(0.0 if v != v else v)
        return {k: sum(v) / len(v) for k, v in per_graph.items()}
