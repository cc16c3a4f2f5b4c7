#!/usr/bin/env python3
"""Register allocation and memory footprint of the kernels in a built library, from the code objects themselves.

  python tools/kernel_meta.py [gym-os2r_amd/libos2r.so | file.hsaco] [substring of the demangled kernel name]

rocprofv3's kernel-trace fields VGPR_Count / Accum_VGPR_Count are not the allocation: the numbers
that are come from the AMDGPU metadata note of the gfx950 code object -- .vgpr_count (the unified VGPR + AGPR
budget as the hardware sees it is vgpr_count + agpr_count), .sgpr_spill_count, .vgpr_spill_count,
.private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size (LDS bytes per workgroup).
A shared library built by hipcc carries one clang offload bundle per translation unit in .hip_fatbin; each is
unpacked here (magic, entry table, the hipv4-amdgcn-amd-amdhsa--gfx950 entry) and read with llvm-readelf --notes.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "kernarg_segment_size")


def code_objects(path):
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and MAGIC not in data:
        yield data
        return
    for m in re.finditer(MAGIC, data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                yield data[p + o:p + o + sz]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")[:len(names)]


def kernel_meta(path):
    """{demangled kernel name: {field: value}} for every kernel of every gfx950 code object in `path`"""
    import yaml
    out = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", notes, re.S | re.M)
        if not m:
            continue
        kernels = (yaml.safe_load(m.group(1)) or {}).get("amdhsa.kernels", [])
        for k, dn in zip(kernels, demangle([k[".name"] for k in kernels])):
            out[dn] = {f: k.get("." + f) for f in FIELDS}
    return out


def norm(name):
    return re.sub(r"\s+", "", name).replace("(anonymousnamespace)::", "")


def lookup(meta, kernel_name):
    """the entry whose demangled name equals rocprofv3's Kernel_Name (modulo spacing and the leading 'void ')"""
    want = norm(kernel_name)
    want = want[4:] if want.startswith("void") else want
    for k, v in meta.items():
        kk = norm(k)
        kk = kk[4:] if kk.startswith("void") else kk
        if kk == want or kk.split("(")[0] == want.split("(")[0]:
            return v
    return None


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gym-os2r_amd", "libos2r.so")
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    meta = {k: v for k, v in kernel_meta(path).items() if sub in k}
    print(json.dumps(meta, indent=1))
