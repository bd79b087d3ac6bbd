"""Seeded synthetic batches shaped like the BASELINE.json workloads.

No dataset can be downloaded here, so every measured or tested batch comes from
these generators.  Profiles:

* ``P30``   BASELINE.json's "~30-node mols": n_g = clamp(round(N(30, 7.5^2)), 4, 64)
* ``P14``   dataset-faithful PCQM4Mv2:        clamp(round(N(14.1, 2.9^2)), 2, 20)
* ``ZINC``  clamp(round(N(23.2, 4.5^2)), 9, 37)
* ``CODE2_LONG`` n_g ~ U{600..1000};  ``CODE2_REAL`` lognormal(mean 125) clipped to 1000

Topology is molecule-like: a random spanning tree with a locality window of 4
plus floor(n/12) ring-closing chords, both directions emitted as adjacent
``(i,j),(j,i)`` pairs (OGB order, i.e. NOT sorted by target).  code2 graphs are
an AST: tree + reverse + next-token chain over ~50 % of the nodes + reverse,
concatenated in four groups as ``/root/reference/graphgps/loader/ogbg_code2_utils.py:81-127``
emits them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .data import Batch

# OGB feature vocabulary sizes (ogb.utils.features.get_{atom,bond}_feature_dims; third-party)
ATOM_FEATURE_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


def graph_sizes(profile: str, num_graphs: int, gen: torch.Generator) -> List[int]:
    def normal(mu, sd, lo, hi):
        v = torch.randn(num_graphs, generator=gen) * sd + mu
        return v.round().clamp(lo, hi).long().tolist()
    p = profile.upper()
    if p == "P30":
        return normal(30.0, 7.5, 4, 64)
    if p == "P14":
        return normal(14.1, 2.9, 2, 20)
    if p == "ZINC":
        return normal(23.2, 4.5, 9, 37)
    if p == "CODE2_LONG":
        return torch.randint(600, 1001, (num_graphs,), generator=gen).tolist()
    if p == "CODE2_REAL":
        sd = 0.8
        mu = math.log(125.0) - 0.5 * sd * sd
        v = torch.exp(torch.randn(num_graphs, generator=gen) * sd + mu)
        return v.round().clamp(11, 1000).long().tolist()
    if p.startswith("U") and p[1:].isdigit():        # probes: every graph of the same size (tools/kernel_probe.py U30)
        return [int(p[1:])] * num_graphs
    raise ValueError(f"unknown synthetic profile {profile!r}")


def molecule_edges(n: int, gen: torch.Generator) -> torch.Tensor:
    """[2, E] int64 local edge_index of one molecule-like graph."""
    if n <= 1:
        return torch.zeros(2, 0, dtype=torch.long)
    k = torch.arange(1, n)
    lo = (k - 4).clamp(min=0)
    parent = lo + (torch.rand(n - 1, generator=gen) * (k - lo)).long()
    und = {(int(min(a, b)), int(max(a, b))) for a, b in zip(k.tolist(), parent.tolist())}
    pairs: List[Tuple[int, int]] = [(int(a), int(b)) for a, b in zip(k.tolist(), parent.tolist())]
    for _ in range(n // 12):
        a = int(torch.randint(0, n, (1,), generator=gen))
        b = a + int(torch.randint(2, 7, (1,), generator=gen))
        if b < n and (a, b) not in und:
            und.add((a, b))
            pairs.append((a, b))
    e = torch.tensor(pairs, dtype=torch.long)
    # adjacent (i,j),(j,i)
    return torch.stack([torch.stack([e[:, 0], e[:, 1]], 1).reshape(-1),
                        torch.stack([e[:, 1], e[:, 0]], 1).reshape(-1)])


def ast_edges(n: int, gen: torch.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """code2-like AST: returns (edge_index [2,E], edge_attr int64 [E,2])."""
    k = torch.arange(1, n)
    lo = (k - 12).clamp(min=0)
    parent = lo + (torch.rand(n - 1, generator=gen) * (k - lo)).long()
    tree = torch.stack([parent, k])
    attributed = torch.nonzero(torch.rand(n, generator=gen) < 0.5).flatten()
    chain = (torch.stack([attributed[:-1], attributed[1:]]) if attributed.numel() > 1
             else torch.zeros(2, 0, dtype=torch.long))
    ei = torch.cat([tree, tree.flip(0), chain, chain.flip(0)], dim=1)
    nt, nc = tree.shape[1], chain.shape[1]
    # (edge type: 0 AST / 1 next-token, direction: 0 fwd / 1 inverse) -- ast_encoder.py:78-83
    ea = torch.cat([torch.tensor([[0, 0]]).repeat(nt, 1), torch.tensor([[0, 1]]).repeat(nt, 1),
                    torch.tensor([[1, 0]]).repeat(nc, 1), torch.tensor([[1, 1]]).repeat(nc, 1)])
    return ei, ea


def make_structure(profile: str, num_graphs: int, seed: int = 1234):
    """Topology only: (sizes, edge_index [2,E] global ids, batch [N], ptr [B+1], gen, extras)."""
    gen = torch.Generator().manual_seed(seed)
    sizes = graph_sizes(profile, num_graphs, gen)
    eis, eas, off = [], [], 0
    for n in sizes:
        if profile.upper().startswith("CODE2"):
            ei, ea = ast_edges(n, gen)
            eas.append(ea)
        else:
            ei = molecule_edges(n, gen)
        eis.append(ei + off)
        off += n
    edge_index = torch.cat(eis, dim=1) if eis else torch.zeros(2, 0, dtype=torch.long)
    ptr = torch.tensor([0] + torch.tensor(sizes).cumsum(0).tolist(), dtype=torch.long)
    batch = torch.repeat_interleave(torch.arange(num_graphs), torch.tensor(sizes))
    extras = {"code2_edge_attr": torch.cat(eas)} if eas else {}
    return sizes, edge_index, batch, ptr, gen, extras


def layer_batch(profile: str, num_graphs: int, dim: int, seed: int = 1234) -> Batch:
    """Layer-level micro-bench / parity input: x ~ N(0,1) [N,d], edge_attr ~ N(0,1) [E,d]."""
    sizes, edge_index, batch, ptr, gen, _ = make_structure(profile, num_graphs, seed)
    N, E = int(ptr[-1]), edge_index.shape[1]
    b = Batch(x=torch.randn(N, dim, generator=gen), edge_index=edge_index,
              edge_attr=torch.randn(E, dim, generator=gen), batch=batch, ptr=ptr)
    b.num_graphs = num_graphs
    return b


def model_batch(kind: str, num_graphs: int, seed: int = 1234, profile: str = None) -> Batch:
    """Model-level input for ``GPSModel``: integer features + PE statistics + targets."""
    kind = kind.lower()
    default_profile = {"pcqm4m": "P30", "zinc": "ZINC", "code2": "CODE2_LONG"}[kind]
    sizes, edge_index, batch, ptr, gen, extras = make_structure(profile or default_profile,
                                                                num_graphs, seed)
    N, E = int(ptr[-1]), edge_index.shape[1]

    def randint_cols(rows, dims):
        return torch.stack([torch.randint(0, d, (rows,), generator=gen) for d in dims], 1)

    b = Batch(edge_index=edge_index, batch=batch, ptr=ptr)
    b.num_graphs = num_graphs
    if kind == "pcqm4m":
        b.x = randint_cols(N, ATOM_FEATURE_DIMS)
        b.edge_attr = randint_cols(E, BOND_FEATURE_DIMS)
        b.pestat_RWSE = torch.rand(N, 16, generator=gen)
        b.y = torch.randn(num_graphs, generator=gen) * 1.2 + 5.7
    elif kind == "zinc":
        b.x = torch.randint(0, 28, (N, 1), generator=gen)
        b.edge_attr = torch.randint(0, 4, (E,), generator=gen)
        b.pestat_RWSE = torch.rand(N, 20, generator=gen)
        b.y = torch.randn(num_graphs, generator=gen)
    elif kind == "code2":
        b.x = torch.stack([torch.randint(0, 98, (N,), generator=gen),
                           torch.randint(0, 10030, (N,), generator=gen)], 1)
        b.node_depth = torch.randint(0, 21, (N, 1), generator=gen)
        b.edge_attr = extras["code2_edge_attr"]
        # 5 sub-token targets per graph out of a 5002-word vocabulary
        b.y_arr = torch.randint(0, 5002, (num_graphs, 5), generator=gen)
        b.y = b.y_arr
    else:
        raise ValueError(kind)
    return b
