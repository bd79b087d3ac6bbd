#!/usr/bin/env python
"""pmc_<n>.txt files of tools/pmc_collect.sh -> one JSON: per kernel the mean of every counter, the HBM bytes per
launch (2 * FETCH_SIZE + WRITE_SIZE KiB: on gfx950 FETCH_SIZE reports half of a 16 B/lane coalesced read --
MI355X_MICROARCH.md, HBM section) and the MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed
over the 8 XCDs * 128 SIMDs per XCD))."""
import glob
import json
import os
import re
import sys


def main():
    d = sys.argv[1]
    ker = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc_*.txt"))):
        for line in open(f):
            m = re.match(r"(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.eE+-]+)\s+min=.*?max=\s*[\d.eE+-]+\s+(.*)$", line)
            if not m:
                continue
            ctr, n, avg, name = m.group(1), int(m.group(2)), float(m.group(3)), m.group(4)
            k = re.search(r"k_gemm_(?:panel|ring|ring16)<[^>]*>|k_wgrad_stream<[^>]*>|k_\w+", name)
            if not k:
                continue
            ker.setdefault(k.group(0), {})[ctr] = avg
            ker[k.group(0)].setdefault("_launches_seen", n)
    out = {}
    for k, c in sorted(ker.items()):
        e = dict(c)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)
        if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            e["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        out[k] = e
    json.dump({"note": __doc__.strip(), "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
