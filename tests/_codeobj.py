"""Read the gfx950 code objects out of libsqd_hip.so and list every kernel's resource record (test infrastructure).

The library is linked from ten translation units, so its ``.hip_fatbin`` section is a sequence of clang offload bundles;
each bundle's ``hipv4-amdgcn-amd-amdhsa--gfx950`` entry is an ELF whose ``NT_AMDGPU_METADATA`` note (msgpack, printed as
YAML by ``llvm-readelf --notes``) carries, per kernel: VGPR / SGPR / AGPR counts, spill counts, the private (scratch)
segment size, the static LDS size and whether the kernel makes calls through a dynamic stack.
"""

from __future__ import annotations

import re
import shutil
import struct
import subprocess
import tempfile
from pathlib import Path

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")


def tools_available() -> bool:
    return (LLVM_BIN / "llvm-readelf").exists() and (LLVM_BIN / "llvm-objcopy").exists()


def code_objects(lib: Path) -> list[bytes]:
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM_BIN / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(lib), str(Path(td) / "x.so")],
                       check=True, capture_output=True)  # fmt: skip
        blob = fat.read_bytes()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24 : p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off : pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def kernel_records(lib: Path) -> list[dict]:
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, dynamic_stack}] over every kernel of the library."""
    recs = []
    names = []
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib)):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            txt = subprocess.run([str(LLVM_BIN / "llvm-readelf"), "--notes", str(f)], check=True, capture_output=True, text=True).stdout
            for ent in txt.split("- .agpr_count:")[1:]:
                def g(key, default="0"):
                    m = re.search(r"\." + key + r":\s+(\S+)", ent)
                    return m.group(1) if m else default
                recs.append({
                    "name": g("name", "?"), "agpr": int(ent.split()[0]), "vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")),
                    "vgpr_spill": int(g("vgpr_spill_count")), "sgpr_spill": int(g("sgpr_spill_count")),
                    "scratch": int(g("private_segment_fixed_size")), "lds": int(g("group_segment_fixed_size")),
                    "dynamic_stack": g("uses_dynamic_stack", "false") == "true",
                })  # fmt: skip
                names.append(recs[-1]["name"])
    filt = shutil.which("c++filt")
    if filt and names:
        dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        for r, d in zip(recs, dem):
            r["demangled"] = d
    return recs


if __name__ == "__main__":
    import sys

    lib = Path(sys.argv[1] if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "qiskit-addon-sqd_amd" / "csrc" / "libsqd_hip.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else "."
    for r in kernel_records(lib):
        d = r.get("demangled", r["name"])
        if re.search(pat, d):
            print(f"{d[:90]:90s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} spill {r['vgpr_spill']:3d} scratch {r['scratch']:4d}"
                  f"{' DYNSTACK' if r['dynamic_stack'] else ''}")
