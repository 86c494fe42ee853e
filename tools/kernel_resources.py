#!/usr/bin/env python3
"""Per-kernel resource usage of the gfx950 code objects inside reak_amd/librkh.so, read from the AMDGPU metadata notes
(`.vgpr_count`, `.agpr_count`, `.vgpr_spill_count`, `.sgpr_count`, `.private_segment_fixed_size`,
`.group_segment_fixed_size`).  Needs no GPU.

    python tools/kernel_resources.py                    # table on stdout
    python tools/kernel_resources.py --out profiles/r03_kernel_resources.txt

`kernel_resources(path)` returns {demangled kernel name: dict}; tests/test_kernel_resources.py asserts the figures the
bench line and DESIGN.md quote against it."""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
          "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def _fatbin(so_path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={out}", so_path,
                        os.path.join(td, "discard")], check=True)
        return open(out, "rb").read()


def code_objects(so_path, arch="gfx950"):
    """The device code objects (ELF images) of every translation unit bundled in the library."""
    blob = _fatbin(so_path)
    pos, objs = 0, []
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if arch in triple and size:
                objs.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return objs


def _demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines() if r.returncode == 0 else names
    return [re.sub(r"^void\s+", "", re.sub(r"\(.*$", "", d.replace("(anonymous namespace)::", ""))) for d in out]


def kernel_resources(so_path=None):
    so_path = so_path or os.path.join(ROOT, "reak_amd", "librkh.so")
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(so_path)):
            p = os.path.join(td, f"co{i}.o")
            open(p, "wb").write(co)
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", p], capture_output=True, text=True,
                                 check=True).stdout
            # one block per kernel: starts at "  - .agpr_count" (first key, alphabetical) and runs to the next one
            blocks = re.split(r"\n  - (?=\.agpr_count|\.args)", txt)
            for b in blocks[1:]:
                m = re.search(r"\.name:\s+(\S+)", b)
                if not m:
                    continue
                d = {}
                for f in FIELDS:
                    mm = re.search(r"\." + f + r":\s+(\d+)", b)
                    d[f] = int(mm.group(1)) if mm else 0
                res[m.group(1)] = d
    names = list(res)
    return {dn: res[mn] for mn, dn in zip(names, _demangle(names))}


def waves_per_simd(d):
    """Occupancy bound from the register allocation (MI355X_MICROARCH.md, register files: 512 per lane and SIMD, granule 8)."""
    alloc = -(-max(1, d["vgpr_count"]) // 8) * 8  # .vgpr_count is the unified total (architectural + accumulation registers)
    return max(1, min(8, 512 // alloc))


def table(res):
    rows = ["# kernel resources of reak_amd/librkh.so (gfx950), from the code objects' AMDGPU metadata",
            "# vgpr = .vgpr_count (unified total: architectural + accumulation registers; agpr = the accumulation part), spill = .vgpr_spill_count,",
            "# scratch = .private_segment_fixed_size [B], lds = .group_segment_fixed_size [B], w/SIMD = register-limited waves per SIMD",
            f"{'kernel':78s} {'vgpr':>5s} {'agpr':>5s} {'spill':>6s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s} {'w/SIMD':>6s}"]
    for k in sorted(res):
        d = res[k]
        rows.append(f"{k[:78]:78s} {d['vgpr_count']:5d} {d['agpr_count']:5d} {d['vgpr_spill_count']:6d} {d['sgpr_count']:5d} "
                    f"{d['private_segment_fixed_size']:8d} {d['group_segment_fixed_size']:7d} {waves_per_simd(d):6d}")
    return "\n".join(rows) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    t = table(kernel_resources(a.so))
    if a.out:
        open(a.out, "w").write(t)
    sys.stdout.write(t)
