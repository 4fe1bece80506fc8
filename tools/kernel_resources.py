#!/usr/bin/env python3
"""Per-kernel resource usage of the gfx950 code objects inside an object file or libFL.so.

Reads the .hip_fatbin section, unpacks the clang offload bundle(s) in it and prints, for every
kernel, the figures of the code object's notes: VGPRs, spilled VGPRs / SGPRs, scratch bytes per
lane, static LDS bytes.  tests/test_host_logic.py uses it to keep `vgpr_spill_count` at 0.

    python tools/kernel_resources.py fortran-library_amd/lib/libFL.so [--demangle] [--spills]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin",
                               path, out])
        with open(out, "rb") as fh:
            return fh.read()


def code_objects(path, arch="gfx950"):
    """the device ELF images for `arch` in path's .hip_fatbin (one per translation unit)"""
    blob = _fatbin(path)
    out = []
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        (nb,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = q
    return out


def kernels(path, arch="gfx950"):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] for every kernel"""
    res = []
    for img in code_objects(path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(img)
            fh.flush()
            txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", fh.name]).decode()
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == "agpr_count" and cur.get("agpr_count") is not None:
                res.append(cur)
                cur = {}
            if k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                     "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                cur[k] = int(v)
            elif k == "name":
                cur["name"] = v
            elif k == "symbol":
                cur["symbol"] = v
        if cur:
            res.append(cur)
    return [r for r in res if "name" in r and "vgpr_count" in r]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names).encode(),
                         stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_spills = "--spills" in sys.argv
    ks = []
    for p in args:
        ks += kernels(p)
    names = demangle([k["name"] for k in ks])
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for k, nm in zip(ks, names):
        if only_spills and not (k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)):
            continue
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*$", "", nm)
        print(f"{k['vgpr_count']:>5} {k.get('agpr_count', 0):>5} {k.get('sgpr_count', 0):>5} "
              f"{k.get('vgpr_spill_count', 0):>6} {k.get('sgpr_spill_count', 0):>6} "
              f"{k.get('private_segment_fixed_size', 0):>7} {k.get('group_segment_fixed_size', 0):>7}  {nm}")


if __name__ == "__main__":
    main()
