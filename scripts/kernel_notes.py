#!/usr/bin/env python3
"""Register / spill figures of every kernel in libbtf_hip.so, from the code-object notes.

The .so embeds one clang offload bundle per translation unit; each holds a gfx950 ELF whose NT_AMDGPU_METADATA note
(msgpack) lists, per kernel, the VGPR / SGPR counts and the spill counts.  No GPU needed.

    python scripts/kernel_notes.py [--spills] [--match REGEX] [lib]
"""
import argparse
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """The gfx950 ELF images inside the library, as bytes."""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(MAGIC, blob):
        o = m.start()
        (n,) = struct.unpack_from("<Q", blob, o + 24)
        p = o + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + ts].decode()
            p += ts
            if "gfx950" in triple and size:
                out.append(blob[o + off:o + off + size])
    return out


def _notes(elf):
    """Parse the NT_AMDGPU_METADATA (type 32) note of an ELF image with msgpack."""
    import msgpack
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        stype, = struct.unpack_from("<I", elf, sh + 4)
        if stype != 7:      # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return out.stdout.split("\n")[:len(names)]
    except Exception:
        return names


def kernels(path=None):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] for every kernel of the library."""
    path = path or os.path.join(ROOT, "functionalmf_amd", "libbtf_hip.so")
    rows = []
    for elf in code_objects(path):
        md = _notes(elf)
        if not md:
            continue
        for k in md.get("amdhsa.kernels", []):
            rows.append(dict(mangled=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                             vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                             scratch=k.get(".private_segment_fixed_size", 0), lds=k.get(".group_segment_fixed_size", 0)))
    for r, d in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = re.sub(r"^void ", "", d)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?")
    ap.add_argument("--spills", action="store_true", help="only kernels that spill VGPRs or SGPRs")
    ap.add_argument("--match", default=None)
    a = ap.parse_args()
    rows = kernels(a.lib)
    if a.match:
        rows = [r for r in rows if re.search(a.match, r["name"])]
    shown = [r for r in rows if not a.spills or r["vgpr_spill"] or r["sgpr_spill"]]
    for r in sorted(shown, key=lambda r: (-r["vgpr_spill"], -r["sgpr_spill"], r["name"])):
        print("%4d vgpr %3d agpr %3d sgpr  spills v %4d s %4d  scratch %6d  %s" % (r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"],
                                                                               r["sgpr_spill"], r["scratch"], r["name"][:150]))
    print("%d kernels, %d spill VGPRs, %d spill SGPRs" % (len(rows), sum(1 for r in rows if r["vgpr_spill"]),
                                                          sum(1 for r in rows if r["sgpr_spill"])), file=sys.stderr)


if __name__ == "__main__":
    main()
