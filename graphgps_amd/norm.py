"""Host side of the task-list BatchNorm / residual / dropout kernels (csrc/block_norm.hip, include/gps_hip.h
``gps_norm_fwd`` / ``gps_norm_bwd_partial`` / ``gps_norm_bwd_apply``).

A launch is a LIST of up to four independent row-stream tasks; the column reductions a BatchNorm needs (batch statistics
forward -- graphgps/layer/gatedgcn_layer.py:72-73, gps_layer.py:191-194,212-229 --, ``sum g`` / ``sum g * zhat``
backward) complete inside the producing launch (csrc/col_tree.hpp), so there is no finalize launch.  The fused blocks
(``layer/gps_block.py``) compose their stages from the helpers below; nothing here allocates more than the scratch for
the partial records and nothing synchronises.
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_int32, c_int64, c_uint64, c_void_p

import torch

from . import lib as _lib
from .lib import BnDesc, check, current_stream

LOAD, ADD_DROP, BN_ACT, BN_DUAL = 0, 1, 2, 3


class FwdTask(ctypes.Structure):
    """``gps_norm_fwd_task`` (include/gps_hip.h)."""
    _fields_ = [("kind", c_int32), ("relu", c_int32), ("a", c_void_p), ("b", c_void_p), ("res", c_void_p),
                ("bn1", c_void_p), ("bn2", c_void_p), ("p", c_float), ("seed", c_uint64), ("out", c_void_p),
                ("R", c_int64), ("stats", c_void_p), ("amax", c_void_p), ("rdev", c_void_p)]


class BwdTask(ctypes.Structure):
    """``gps_norm_bwd_task`` (include/gps_hip.h)."""
    _fields_ = [("z", c_void_p), ("g_y", c_void_p), ("bn", c_void_p), ("relu", c_int32), ("p", c_float),
                ("seed", c_uint64), ("z2", c_void_p), ("bn2", c_void_p),
                ("g_gamma", c_void_p), ("g_beta", c_void_p), ("g_gamma2", c_void_p), ("g_beta2", c_void_p),
                ("g_z", c_void_p), ("g_sum", c_void_p), ("g_drop", c_void_p),
                ("p2", c_float), ("seed2", c_uint64), ("p1x", c_float), ("seed1x", c_uint64), ("R", c_int64),
                ("cz", c_void_p), ("cbn", c_void_p), ("crelu", c_int32), ("cp", c_float), ("cseed", c_uint64),
                ("cg_gamma", c_void_p), ("cg_beta", c_void_p), ("amax_drop", c_void_p), ("rdev", c_void_p)]


def _p(t):
    return None if t is None else t.data_ptr()


def _bn(desc):
    return None if desc is None else ctypes.addressof(desc)


def bn_desc(bn, mean, rstd) -> BnDesc:
    """``gps_bn`` descriptor of a BatchNorm1d + the [d] buffers holding its batch statistics."""
    # (straight from the module's dictionaries: nn.Module.__getattr__ is the slow path, and this runs 100 times per step)
    pr, bf = bn._parameters, bn._buffers
    return BnDesc(pr["weight"].data_ptr(), pr["bias"].data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  bf["running_mean"].data_ptr(), bf["running_var"].data_ptr(), float(bn.eps), float(bn.momentum))


def fwd_task(kind, a, R, *, b=None, res=None, bn1=None, bn2=None, relu=False, p=0.0, seed=0, out=None, stats=None,
             amax=None, rdev=None):
    """Keeps the ``BnDesc`` objects it points to alive through ``._keep``.  ``amax`` (a max|.| record): raised to max|out|;
    ``rdev`` (int32 [1] on the device): the number of real rows of a padded batch."""
    t = FwdTask(kind, int(relu), _p(a), _p(b), _p(res), _bn(bn1), _bn(bn2), float(p), int(seed), _p(out), int(R),
                _bn(stats), _p(amax), _p(rdev))
    t._keep = (bn1, bn2, stats)
    return t


def bwd_task(z, g_y, bn, R, g_gamma, g_beta, *, relu=False, p=0.0, seed=0, z2=None, bn2=None, g_gamma2=None,
             g_beta2=None, g_z=None, g_sum=None, g_drop=None, p2=0.0, seed2=0, p1x=0.0, seed1x=0,
             cz=None, cbn=None, crelu=False, cp=0.0, cseed=0, cg_gamma=None, cg_beta=None, amax_drop=None, rdev=None):
    t = BwdTask(_p(z), _p(g_y), _bn(bn), int(relu), float(p), int(seed), _p(z2), _bn(bn2),
                _p(g_gamma), _p(g_beta), _p(g_gamma2), _p(g_beta2), _p(g_z), _p(g_sum), _p(g_drop),
                float(p2), int(seed2), float(p1x), int(seed1x), int(R),
                _p(cz), _bn(cbn), int(crelu), float(cp), int(cseed), _p(cg_gamma), _p(cg_beta), _p(amax_drop), _p(rdev))
    t._keep = (bn, bn2, cbn)
    return t


_tree_floats = {}
_sync_words = None


def tree_floats(R: int, d: int) -> int:
    key = (R, d)
    v = _tree_floats.get(key)
    if v is None:
        v = _tree_floats[key] = int(_lib.load().gps_norm_tree_floats(R, d))
    return v


def sync_words() -> int:
    global _sync_words
    if _sync_words is None:
        _sync_words = int(_lib.load().gps_norm_sync_words())
    return _sync_words


N_SITES = 16     # arrival-counter regions per owner: one per launch site that may overlap another in time


class SyncArena:
    """Arrival counters of the in-launch reductions of ONE owner (a GPS layer): zeroed once, left zero by every launch
    (csrc/col_tree.hpp), never touched by the host again -- so the same words serve eager calls and hipGraph replays.
    ``site(k)`` is the region of launch site k; sites that can be in flight together use different k."""

    def __init__(self, device):
        # a site holds the counters of ONE launch: the task-list kernels' trees (gps_norm_sync_words) or a producer's own
        # (GatedGCN: one tree; ring GEMM statistics epilogue: one tree of 32 words per column panel, up to 16 panels of 64
        # columns at the widest d = 1024 the blocks accept -- ADVICE r3: 11-15 panels used to spill into the next site)
        self.words = max(sync_words(), 16 * 32)
        self.buf = torch.zeros(N_SITES * self.words, dtype=torch.int32, device=device)

    def site(self, k: int, need_words: int = 0) -> int:
        assert 0 <= k < N_SITES
        if need_words > self.words:
            raise _lib.GpsHipError(f"SyncArena: a launch needs {need_words} counter words, a site holds {self.words}")
        return self.buf.data_ptr() + 4 * k * self.words

    def tail(self, k: int):
        """(address, words) of everything from site k to the end of the arena: for the one launch whose counter need
        grows with the problem (csrc/wgrad.hip: one counter per output tile); no site above k may be in use."""
        assert 0 <= k < N_SITES
        return self.buf.data_ptr() + 4 * k * self.words, (N_SITES - k) * self.words

    def nonzero_words(self) -> int:
        """Number of non-zero counters (a host read: debugging / tests).  Between launches it must be 0."""
        L = _lib.load()
        cnt = torch.zeros(1, dtype=torch.int32, device=self.buf.device)
        check(L.gps_sync_nonzero(self.buf.data_ptr(), self.buf.numel(), cnt.data_ptr(), current_stream(self.buf.device)),
              "gps_sync_nonzero")
        return int(cnt)

    def reset(self) -> None:
        check(_lib.load().gps_sync_reset(self.buf.data_ptr(), self.buf.numel(), current_stream(self.buf.device)),
              "gps_sync_reset")


def sync_arena(owner, device) -> SyncArena:
    sa = getattr(owner, "_gps_sync", None)
    if sa is None or sa.buf.device != device:
        sa = SyncArena(device)
        owner._gps_sync = sa        # plain attribute: not a buffer, not in the state_dict
    return sa


def _launch(fn, name, tasks, struct, d, R_list, dev, sync_ptr):
    n = len(tasks)
    arr = (struct * n)(*tasks)
    floats = sum(tree_floats(R, d) for R in R_list)
    ws = torch.empty(max(floats, 4), dtype=torch.float32, device=dev)
    check(fn(n, arr, d, ws.data_ptr(), floats, sync_ptr, current_stream(dev)), name)


def fwd(tasks, d, dev, sync_ptr):
    L = _lib.load()
    _launch(L.gps_norm_fwd, "gps_norm_fwd", tasks, FwdTask, d, [t.R for t in tasks if t.stats], dev, sync_ptr)


def bwd_partial(tasks, d, dev, sync_ptr):
    L = _lib.load()
    _launch(L.gps_norm_bwd_partial, "gps_norm_bwd_partial", tasks, BwdTask, d, [t.R for t in tasks], dev, sync_ptr)


def bwd_apply(tasks, d, dev, sync_ptr):
    L = _lib.load()
    _launch(L.gps_norm_bwd_apply, "gps_norm_bwd_apply", tasks, BwdTask, d, [t.R for t in tasks if t.cz], dev, sync_ptr)
