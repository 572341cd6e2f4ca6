"""What the built library's gfx950 code objects say about the kernels that issue the double-rate MFMA (no GPU needed).

DESIGN.md 5.10: dense v_mfma_f32_32x32x16_bf16 issue damages packed-fp32 results of FOREIGN waves resident on the same compute unit,
so the kernels that use it must own their CU: 8-wave workgroups whose waves take 256 registers each (two per SIMD = its whole
register file: not even an 8-register fill kernel fits beside them) and more than half of the CU's LDS.  The counts are pinned in the
sources by asm clobbers; a compiler that allocates differently would reopen the hole silently, so this test reads them back from
semantic-gaussians_amd/sgs_hip/libsgs_hip.so itself."""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "semantic-gaussians_amd", "sgs_hip", "libsgs_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
X16 = "v_mfma_f32_32x32x16_bf16"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def _code_objects(td):
    """The gfx950 ELF images of every translation unit: .hip_fatbin is a sequence of clang offload bundles."""
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, SO])
    d = open(fat, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = d.find(magic, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", d, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size > 64:
                out.append(d[i + off:i + off + size])
        pos = i + 1


def _kernel_metadata(elf):
    """amdhsa.kernels of the NT_AMDGPU_METADATA note (msgpack)."""
    import msgpack
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    out = []
    for k in range(shnum):
        sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
        if sh[1] != 7:   # SHT_NOTE
            continue
        q, end = sh[4], sh[4] + sh[5]
        while q < end:
            namesz, descsz, typ = struct.unpack_from("<III", elf, q)
            q += 12 + ((namesz + 3) & ~3)
            desc = elf[q:q + descsz]
            q += (descsz + 3) & ~3
            if typ == 32:
                out += msgpack.unpackb(desc, raw=False, strict_map_key=False).get("amdhsa.kernels", [])
    return out


def _x16_kernels():
    if not os.path.exists(SO):
        pytest.skip("libsgs_hip.so is not built")
    if not _tool("llvm-objcopy") or not _tool("llvm-objdump"):
        pytest.skip("llvm-objcopy / llvm-objdump not found")
    pytest.importorskip("msgpack")
    found = {}
    with tempfile.TemporaryDirectory() as td:
        for n, elf in enumerate(_code_objects(td)):
            meta = {k[".symbol"][:-3] if k[".symbol"].endswith(".kd") else k[".name"]: k for k in _kernel_metadata(elf)}
            fn = os.path.join(td, f"co{n}.elf")
            open(fn, "wb").write(elf)
            dis = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", fn], capture_output=True, text=True, check=True).stdout
            cur = None
            packed = {}
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    continue
                op = line.split()[0] if line.split() else ""
                if op.startswith("v_pk_") and op.endswith("_f32"):
                    packed[cur] = packed.get(cur, 0) + 1
                if X16 in line:
                    assert cur in meta, f"{X16} outside a kernel body: {cur}"   # (a device function that was not inlined would need its callers checked)
                    found[cur] = meta[cur]
            for name in found:
                found[name].setdefault("packed_f32_instructions", packed.get(name, 0))
    return found


def test_every_kernel_on_the_double_rate_mfma_owns_its_compute_unit():
    ks = _x16_kernels()
    names = sorted(ks)
    # the product library ships exactly these (make X16=1 / EXPERIMENTS=1 add reproducers whose POINT is that they do not own a CU)
    ships = [n for n in names if "blend_accum_sweep3_kernel" in n or "bwd_fused_kernel" in n]
    assert any("blend_accum_sweep3_kernel" in n for n in ships) and any("bwd_fused_kernel" in n for n in ships), names
    import ctypes
    flags = ctypes.CDLL(SO).sgs_build_flags()
    if flags == 0:
        assert names == ships, f"a kernel outside the two CU-owning designs issues {X16}: {set(names) - set(ships)}"
    for n in ships:
        k = ks[n]
        # unified register count (VGPRs + AccVGPRs) of a wave: 256 = half of a SIMD's 512-entry file; a workgroup of 8 waves = 2 per SIMD
        assert k[".vgpr_count"] == 256, (n, k[".vgpr_count"], k.get(".agpr_count"))
        assert k[".max_flat_workgroup_size"] == 512, (n, k[".max_flat_workgroup_size"])
        assert k[".group_segment_fixed_size"] > 80 * 1024, (n, k[".group_segment_fixed_size"])   # no second workgroup of its kind either
        assert k.get(".private_segment_fixed_size", 0) == 0, (n, "spills")
        # the waves of one workgroup are each other's neighbours on the CU: the damaged class -- packed-fp32 VALU results -- must not occur in
        # the kernel itself (blend_bwd_mfma.hip is compiled with -fno-slp-vectorize for this; blend_sweep2.hip's matrix phases are inline asm)
        assert k["packed_f32_instructions"] == 0, (n, k["packed_f32_instructions"])
