"""Register / scratch / LDS footprint of every kernel in an object file built by csrc/Makefile (no GPU needed).

    python tools/kernel_regs.py graphgps_amd/csrc/gemm_panel.o [substring ...]

Reads the AMDGPU metadata notes of the gfx950 code object embedded in the host object's .hip_fatbin section."""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except FileNotFoundError:
        return names


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([LLVM + "llvm-objcopy", f"--dump-section=.hip_fatbin={d}/fat.bin", obj], check=True)
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/dev.co"], check=True)
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", f"{d}/dev.co"], capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        f = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        rows.append(dict(name=re.search(r"\.name:\s+(\S+)", blk).group(1), agpr=int(re.match(r"\s*(\d+)", blk).group(1)),
                         vgpr=f("vgpr_count"), sgpr=f("sgpr_count"), vspill=f("vgpr_spill_count"),
                         sspill=f("sgpr_spill_count"), scratch=f("private_segment_fixed_size"),
                         lds=f("group_segment_fixed_size")))
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["pretty"] = n
    return rows


if __name__ == "__main__":
    pats = sys.argv[2:]
    for r in kernels(sys.argv[1]):
        if pats and not any(p in r["pretty"] for p in pats):
            continue
        print(f'{r["pretty"][:96]:96s} vgpr {r["vgpr"]:3d} agpr {r["agpr"]:3d} sgpr {r["sgpr"]:3d} '
              f'spill v{r["vspill"]} s{r["sspill"]} scratch {r["scratch"]} lds {r["lds"]}')
