from graphgps_amd.graphgym.config import CfgNode  # noqa: F401
