"""Stub of torch_geometric.data.Batch: attribute container (the only use on the path is the
keyword constructor at graphgps/layer/gps_layer.py:167-171)."""


class Batch:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
