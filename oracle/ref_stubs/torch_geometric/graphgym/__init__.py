from graphgps_amd.graphgym.config import cfg  # noqa: F401  (the reference does `from torch_geometric.graphgym import cfg`)
