"""Register / scratch / LDS table of every gfx950 kernel inside libdynam3d_hip.so, read from the code objects' own metadata notes
(NT_AMDGPU_METADATA, msgpack) -- no ROCm tool needed, runs without a GPU.

    python tools/kernel_resources.py [--scratch] [substring ...]

Why it exists (round 6): wrapping the 256 x 256 GEMM's body in a one-trip `for` made hipcc spill 9-56 VGPRs to SCRATCH in every
instantiation; on the GPU that build produced wrong logits or a memory access fault in ~1 of 3 processes (tests/test_gpu_full_step.py;
bisect in profiles/r06_gemm_scratch_regression.txt).  tests/test_kernel_resources.py now holds the product kernels to zero scratch."""
import os
import re
import struct
import sys

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynam3d_amd", "libdynam3d_hip.so")


def _code_objects(blob: bytes):
    for m in re.finditer(b"\x7fELF\x02\x01\x01", blob):
        i = m.start()
        if struct.unpack_from("<H", blob, i + 18)[0] != 224:           # EM_AMDGPU
            continue
        shoff = struct.unpack_from("<Q", blob, i + 40)[0]
        shentsize, shnum = struct.unpack_from("<HH", blob, i + 58)
        yield blob[i:i + shoff + shentsize * shnum]


def _metadata(elf: bytes):
    import msgpack
    shoff = struct.unpack_from("<Q", elf, 40)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 58)
    for s in range(shnum):
        typ, = struct.unpack_from("<I", elf, shoff + s * shentsize + 4)
        if typ != 7:                                                    # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, shoff + s * shentsize + 24)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:
                return msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
            p = d0 + (descsz + 3) // 4 * 4
    return None


def demangle_hint(name: str) -> str:
    """`_ZN..13k_gemm_nt_256ILb1ELi6ELb1ELb0ELi2ELb0EEEv...` -> `k_gemm_nt_256<1,6,1,0,2,0>` (enough to read a table; not a demangler)."""
    m = re.search(r"\d+(k_[a-z0-9_]+?)I((?:L[a-z]\d+E)+)E", name)
    if not m:
        m2 = re.search(r"\d+(k_[a-z0-9_]+)", name)
        return m2.group(1) if m2 else name
    return m.group(1) + "<" + ",".join(re.findall(r"L[a-z](\d+)E", m.group(2))) + ">"


def kernel_table(lib: str = LIB):
    """[{name, short, vgpr, agpr, sgpr, spills, scratch, lds}] for every kernel of every code object in `lib`."""
    blob = open(lib, "rb").read()
    rows = []
    for elf in _code_objects(blob):
        md = _metadata(elf)
        for k in (md or {}).get("amdhsa.kernels", []):
            rows.append(dict(name=k[".name"], short=demangle_hint(k[".name"]), vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0),
                             sgpr=k.get(".sgpr_count", 0), spills=k.get(".vgpr_spill_count", 0) + k.get(".sgpr_spill_count", 0),
                             scratch=k.get(".private_segment_fixed_size", 0), lds=k.get(".group_segment_fixed_size", 0),
                             dynamic_stack=bool(k.get(".uses_dynamic_stack", False))))
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_scratch = "--scratch" in sys.argv
    rows = kernel_table()
    print(f"{len(rows)} kernels in {LIB}")
    for r in sorted(rows, key=lambda r: r["short"]):
        if only_scratch and not (r["scratch"] or r["dynamic_stack"]):
            continue
        if args and not any(a in r["short"] or a in r["name"] for a in args):
            continue
        print(f"{r['short'][:64]:64s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} spills {r['spills']:3d} scratch {r['scratch']:4d} B  lds {r['lds']:6d} B")
