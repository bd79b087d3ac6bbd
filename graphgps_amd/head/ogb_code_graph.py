"""ogbg-code2 sequence head: pool + 5 x Linear(dim_in -> 5002).
Mirror of ``/root/reference/graphgps/head/ogb_code_graph.py:8-45``."""
import torch.nn as nn

from ..graphgym import register
from ..graphgym import pooling as _pooling  # noqa: F401
from ..graphgym.config import cfg
from ..graphgym.register import register_head


@register_head('ogb_code_graph', overwrite=True)
class OGBCodeGraphHead(nn.Module):
    def __init__(self, dim_in, dim_out, L=1):
        super().__init__()
        self.pooling_fun = register.pooling_dict[cfg.model.graph_pooling]
        self.L = L
        num_vocab = 5002
        self.max_seq_len = 5
        if self.L != 1:
            raise ValueError("Multilayer prediction heads are not supported.")
        self.graph_pred_linear_list = nn.ModuleList(
            [nn.Linear(dim_in, num_vocab) for _ in range(self.max_seq_len)])

    def _apply_index(self, batch):
        return batch.pred_list, {'y_arr': batch.y_arr, 'y': batch.y}

    def forward(self, batch):
        gi = batch.__dict__.get("_gps_index") if hasattr(batch, "__dict__") else None
        try:
            graph_emb = self.pooling_fun(batch.x, batch.batch, batch.num_graphs, gi=gi)
        except TypeError:
            graph_emb = self.pooling_fun(batch.x, batch.batch)
        batch.pred_list = [lin(graph_emb) for lin in self.graph_pred_linear_list]
        return self._apply_index(batch)
