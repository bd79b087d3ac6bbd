from graphgps_amd.graphgym.config import cfg, set_cfg, load_cfg  # noqa: F401
