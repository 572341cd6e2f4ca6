"""What the built library's gfx950 code objects say (no GPU needed) -- the same reading `make` does after the link
(semantic-gaussians_amd/csrc/check_code_object.py), kept as a test so that a library built some other way is caught too.

DESIGN.md 5.10: dense v_mfma_f32_32x32x16_bf16 issue damages packed-fp32 results of FOREIGN waves resident on the same compute unit,
so the kernels that use it must own their CU: 8-wave workgroups whose waves take 256 registers each (two per SIMD = its whole
register file: not even an 8-register fill kernel fits beside them) and more than half of the CU's LDS.  Round 6 closes the victim side
as well: no kernel of the product library holds a compiler-made v_pk_*_f32 (device target feature -packed-fp32-ops), what is left is
the hand-written arithmetic of the listed kernels."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "semantic-gaussians_amd", "sgs_hip", "libsgs_hip.so")
_spec = importlib.util.spec_from_file_location("check_code_object", os.path.join(ROOT, "semantic-gaussians_amd", "csrc", "check_code_object.py"))
cco = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cco)


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(SO):
        pytest.skip("libsgs_hip.so is not built")
    if not cco.tool("llvm-objcopy") or not cco.tool("llvm-objdump"):
        pytest.skip("llvm-objcopy / llvm-objdump not found")
    pytest.importorskip("msgpack")
    return cco.scan(SO)


def _product():
    from sgs_hip import _lib   # (not ctypes.CDLL(SO): the binding loads torch's HIP runtime first, so that a whole-suite run on a GPU box keeps ONE runtime)
    return _lib.load().sgs_build_flags() == 0


def test_every_kernel_on_the_double_rate_mfma_owns_its_compute_unit(kernels):
    x16 = sorted(n for n, k in kernels.items() if k["x16_instructions"])
    ships = [n for n in x16 if any(o in n for o in cco.OWNERS)]
    assert any("blend_accum_sweep3_kernel" in n for n in ships) and any("bwd_fused_kernel" in n for n in ships), x16
    if _product():   # (make X16=1 / EXPERIMENTS=1 add reproducers and ablations whose POINT is that they do not own a CU)
        assert x16 == ships, f"a kernel outside the two CU-owning designs issues {cco.X16}: {set(x16) - set(ships)}"
        for n in ships:
            k = kernels[n]
            assert k[".vgpr_count"] == 256, (n, k[".vgpr_count"], k.get(".agpr_count"))
            assert k[".max_flat_workgroup_size"] == 512, (n, k[".max_flat_workgroup_size"])
            assert k[".group_segment_fixed_size"] > 80 * 1024, (n, k[".group_segment_fixed_size"])
            assert k.get(".private_segment_fixed_size", 0) == 0, (n, "spills")
            assert k["packed_f32_instructions"] == 0, (n, k["packed_f32_instructions"])
    assert cco.violations(kernels, _product()) == []   # (what `make` checks: in an experiments build, the shipping instantiations only)


def test_no_packed_fp32_outside_the_hand_written_kernels(kernels):
    """The victim class of DESIGN.md 5.10 -- v_pk_{add,mul,fma}_f32 -- occurs in the product library only where it is written by hand."""
    if not _product():
        pytest.skip("experiments build: the development kernels are not held to this")
    holders = sorted(n for n, k in kernels.items() if k["packed_f32_instructions"])
    stray = [n for n in holders if not any(h in n for h in cco.PACKED_BY_HAND)]
    assert stray == [], stray
    assert any("blend_weights2_sb_kernel" in n for n in holders)   # (the check sees packed instructions at all)


def test_product_library_carries_no_library_kernels(kernels):
    if not _product():
        pytest.skip("experiments build: binning modes 1 / 2 bring rocPRIM's scan and radix sort")
    assert [n for n in kernels if "rocprim" in n] == []
    assert os.path.getsize(SO) < 2_500_000, os.path.getsize(SO)   # (round 5: 5.3 MB, 3.2 MB of it rocPRIM instantiations)


def test_the_check_catches_a_kernel_that_lost_its_registers(kernels):
    """The build step must fail on what it guards against: a sweep at 248 registers (round 5's near miss), x16 in a foreign kernel,
    stray packed fp32."""
    import copy
    ks = copy.deepcopy(kernels)
    owner = next(n for n, k in ks.items() if k["x16_instructions"] and "blend_accum_sweep3_kernel" in n)
    ks[owner][".vgpr_count"] = 248
    assert any("248 registers" in v for v in cco.violations(ks, True))
    ks = copy.deepcopy(kernels)
    victim = next(n for n in ks if "preprocess_fwd_kernel" in n)
    ks[victim]["packed_f32_instructions"] = 5
    ks[victim]["x16_instructions"] = 1
    v = cco.violations(ks, True)
    assert any("outside the two CU-owning designs" in x for x in v) and any("packed-fp32" in x for x in v)
