"""Build-time guards (no GPU): properties of the generated code that the kernels rely on and that no numerical test would pin
down deterministically."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiler_leaves_the_accumulator_agprs_alone():
    """the tile kernels keep their accumulators in AGPRs that only inline asm touches; hipcc must not allocate temporaries there
    (it did once, in a kernel that needed more than its VGPR budget: tools/check_acc_regs.py has the story)."""
    import check_acc_regs
    from concurrent.futures import ThreadPoolExecutor

    def one(item):
        name, expected = item
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-S",
                            "--cuda-device-only", os.path.join(ROOT, "micromix_amd", "csrc", name), "-o", out],
                           check=True, cwd=tmp, stderr=subprocess.DEVNULL)
            bad, examined = check_acc_regs.check_counted(open(out).read())
        return name, expected, bad, examined

    items = list(check_acc_regs.EXPECTED_BY_FILE.items())
    with ThreadPoolExecutor(max_workers=3) as pool:
        for name, expected, bad, examined in pool.map(one, items):
            assert len(examined) >= expected, (name, examined)     # the regex must have matched every tile kernel
            assert not bad, name + "\n" + "\n".join(f"{s}: {c}" for s, c in bad[:10])


VIOLATING = ("_ZN2mm3g6417mx_gemm256_kernelILb0ELb0EEEvNS_8GemmArgsE: ; @x\n"
             "\tv_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[0:7], v[8:11], a[0:15], v3, v4 op_sel_hi:[0,0,0] cbsz:0 blgp:4\n"
             "\tds_read2st64_b64 a[0:3], v113 offset0:16 offset1:24\n"
             ".end_amdhsa_kernel\n")


def test_the_build_guard_raises_on_a_violating_assembly(tmp_path):
    """python -m micromix_amd.build feeds the assembly it kept of mx_gemm256.hip through verify_acc_regs, which must raise on a
    violation and on a file in which the tile kernels were not found (micromix_amd/build.py)"""
    from micromix_amd import build
    with pytest.raises(RuntimeError, match="no device assembly"):
        build.verify_acc_regs(str(tmp_path))
    (tmp_path / "mx_gemm256-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(VIOLATING)
    with pytest.raises(RuntimeError, match="accumulator AGPR"):
        build.verify_acc_regs(str(tmp_path))
    (tmp_path / "mx_gemm256-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(VIOLATING.replace("\tds_read2st64_b64 a[0:3], v113 offset0:16 offset1:24\n", ""))
    with pytest.raises(RuntimeError, match="expected >="):
        build.verify_acc_regs(str(tmp_path))


def test_an_interrupted_build_with_new_flags_is_forced_again(tmp_path, monkeypatch):
    """the flags stamp is written after the link, not before the first compile"""
    from micromix_amd import build
    monkeypatch.setattr(build, "OBJDIR", str(tmp_path))
    assert build._flags_changed(["-O3"])
    assert build._flags_changed(["-O3"])            # asking does not write the stamp
    open(build._flags_stamp(), "w").write("-O3")
    assert not build._flags_changed(["-O3"]) and build._flags_changed(["-O3", "-DX"])


def test_the_guard_itself_detects_a_violation():
    import check_acc_regs
    fake = ("_ZN2mm3g6417mx_gemm256_kernelILb0ELb0EEEvNS_8GemmArgsE: ; @x\n"
            "\tv_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[0:7], v[8:11], a[0:15], v3, v4 op_sel_hi:[0,0,0] cbsz:0 blgp:4\n"
            "\tv_accvgpr_read_b32 v1, a[3]\n"
            "\tds_read2st64_b64 a[0:3], v113 offset0:16 offset1:24\n"
            "\tv_accvgpr_write_b32 a40, v2\n"
            "\tv_accvgpr_read_b32 v5, a7\n"
            ".end_amdhsa_kernel\n")
    bad = check_acc_regs.check(fake)
    assert [c.split()[0] for _, c in bad] == ["ds_read2st64_b64", "v_accvgpr_read_b32"]


def test_every_csrc_header_is_a_build_dependency():
    """a header missing from micromix_amd.build.HEADERS would not trigger a rebuild when it changes (mx_decode_quant.h in round 4)"""
    from micromix_amd import build
    listed = {os.path.basename(h) for h in build.HEADERS}
    present = {f for f in os.listdir(build.CSRC) if f.endswith((".h", ".inc"))}
    assert present <= listed, sorted(present - listed)


STREAM_OK = ("_ZN2mm6stream21mx_gemm_stream_kernelILi2ELi1ELi2ELi8ELb1EEEvNS_8GemmArgsE: ; @x\n"
             "\tv_accvgpr_write_b32 a[0], 0\n"
             "\tbuffer_load_dwordx2 v[10:11], v17, s[8:11], s1 offen\n"
             "\tbuffer_load_dwordx4 v18, s[12:15], s2 offen lds\n"
             "\tv_add_u32_e32 v3, v4, v5\n"
             "\ts_waitcnt vmcnt(1)\n"
             "\tv_lshrrev_b32_e32 v6, 8, v10\n"
             "\tv_mfma_scale_f32_16x16x128_f8f6f4 a[0:3], v[20:23], v[24:27], a[0:3], v6, v7 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n"
             "\tv_accvgpr_read_b32 v1, a[23]\n"
             "\t.amdhsa_private_segment_fixed_size 0\n"
             ".end_amdhsa_kernel\n")


def test_the_stream_guard_detects_planted_violations():
    """mx_gemm_stream.hip keeps 12 F T16 accumulators in asm-owned AGPRs and reads registers the hardware fills after the asm
    statement returned: accumulator touched by the compiler, scratch, and a load destination read
    before its wait are each reported; the clean text is clean; NACC follows the template arguments in the mangled name."""
    import check_acc_regs
    from micromix_amd import _check_acc_regs as c
    assert c.stream_nacc("mx_gemm_stream_kernel", "Li2ELi4ELi2ELi4E") == 96 and c.stream_nacc("mx_qlinear_stream_kernel", "Li4ELi2ELi4E") == 48
    bad, examined = c.check_stream(STREAM_OK)
    assert not bad and len(examined) == 1
    # (F, T16) = (2, 1): a[0:23] are accumulators, a24 is the compiler's
    assert not c.check_stream(STREAM_OK.replace("\tv_add_u32_e32 v3, v4, v5\n", "\tv_accvgpr_write_b32 a24, v2\n"))[0]
    planted = {
        "compiler temporary in an accumulator": STREAM_OK.replace("\tv_add_u32_e32 v3, v4, v5\n", "\tv_accvgpr_write_b32 a5, v2\n"),
        "LDS read into accumulators": STREAM_OK.replace("\tv_add_u32_e32 v3, v4, v5\n", "\tds_read_b128 a[20:23], v9\n"),
        "scratch": STREAM_OK.replace("fixed_size 0", "fixed_size 16"),
        "load destination read before its wait": STREAM_OK.replace("\tv_add_u32_e32 v3, v4, v5\n", "\tv_add_u32_e32 v3, v11, v5\n"),
        "wait that leaves the load in flight": STREAM_OK.replace("vmcnt(1)", "vmcnt(2)"),
    }
    for what, text in planted.items():
        assert c.check_stream(text)[0], what
    # in FRONT of the kernel's first accumulator instruction (the quantization phase of the kernels that quantize their rows themselves)
    # the compiler may park values there -- unless a later branch can bring execution back in front of it
    prefix = STREAM_OK.replace("\tv_accvgpr_write_b32 a[0], 0\n", ".LBB0_1:\n\tv_accvgpr_write_b32 a2, v7\n\tv_accvgpr_read_b32 v20, a2\n\tv_accvgpr_write_b32 a[0], 0\n")
    assert not c.check_stream(prefix)[0]
    assert c.check_stream(prefix.replace("\tv_accvgpr_read_b32 v1, a[23]\n", "\tv_accvgpr_read_b32 v1, a[23]\n\ts_cbranch_scc1 .LBB0_1\n"))[0]
    with pytest.raises(RuntimeError, match="asm-owned registers"):
        c.verify_stream(planted["scratch"])
    with pytest.raises(RuntimeError, match="expected >="):
        c.verify_stream(STREAM_OK)                      # one kernel where 52 are expected: the name regex no longer matches
    assert check_acc_regs                                # (tools/ front end imports)


def test_the_build_guards_both_sources(tmp_path):
    from micromix_amd import build
    assert set(build.GUARDED) == {"mx_gemm256.hip", "mx_gemm_tiles_small.hip", "mx_gemm_stream.hip", "rmsnorm_quantize.hip", "qlinear_decode.hip"}
    assert not any("_w1" in s or "persist" in s for s in build.SOURCES)      # round 6: the round-5 experiments are out of the product library
    from micromix_amd import _check_acc_regs as c
    planted = ("_ZN2mm28rmsnorm_quantize_ring_kernelILb1ELi3EEEvPKt: ; @x\n\tglobal_load_dwordx4 v[14:17], v[4:5], off\n"
               "\tbuffer_load_dwordx4 v1, s[52:55], s62 offen lds\n\tv_lshlrev_b32_e32 v2, 1, v14\n\ts_waitcnt vmcnt(1)\n.end_amdhsa_kernel\n")
    assert c.check_pending(planted)[0] and not c.check_pending(planted.replace("\tv_lshlrev_b32_e32 v2, 1, v14\n\ts_waitcnt vmcnt(1)\n", "\ts_waitcnt vmcnt(1)\n\tv_lshlrev_b32_e32 v2, 1, v14\n"))[0]
    with pytest.raises(RuntimeError, match="no device assembly of mx_gemm_stream.hip"):
        build.verify_acc_regs(str(tmp_path), "mx_gemm_stream.hip")
    (tmp_path / "mx_gemm_stream-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(STREAM_OK.replace("fixed_size 0", "fixed_size 8"))
    with pytest.raises(RuntimeError, match="asm-owned registers"):
        build.verify_acc_regs(str(tmp_path), "mx_gemm_stream.hip")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiler_leaves_the_stream_kernels_registers_alone():
    from micromix_amd import _check_acc_regs as c
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-S",
                        "--cuda-device-only", os.path.join(ROOT, "micromix_amd", "csrc", "mx_gemm_stream.hip"), "-o", out],
                       check=True, cwd=tmp, stderr=subprocess.DEVNULL)
        bad, examined = c.check_stream(open(out).read())
    assert len(examined) >= c.EXPECTED_STREAM_KERNELS == 68, len(examined)      # (+ the four launches with the activation inside, round 6)
    assert not bad, "\n".join(f"{s}: {x}" for s, x in bad[:10])


SCRATCH_OK = ("_ZN2mm4g25617mx_gemm256_kernelILb0ELb0EEEvNS_8GemmArgsE: ; @x\n"
              "\tv_add_u32_e32 v3, v4, v5\n"
              "\t.amdhsa_group_segment_fixed_size 0\n"
              "\t.amdhsa_private_segment_fixed_size 0\n"
              "\t.amdhsa_uses_dynamic_stack 0\n"
              ".end_amdhsa_kernel\n"
              "_ZN2mm23reorder_quantize_kernelILb1EEEvPKt: ; @y\n"
              "\tv_add_u32_e32 v3, v4, v5\n"
              "\t.amdhsa_group_segment_fixed_size 4096\n"
              "\t.amdhsa_private_segment_fixed_size 0\n"
              ".end_amdhsa_kernel\n")


def test_the_scratch_guard_covers_every_kernel_and_detects_planted_violations(tmp_path):
    """`g256::mx_gemm256_kernel<false,false>` shipped with one spilled VGPR whose reload drained the DMA ring inside
    the K loop, and no guard looked.  Now every kernel of every product object fails the build on scratch (a non-zero
    private_segment_fixed_size, a dynamic stack or a scratch instruction), and the tile kernels on static LDS."""
    from micromix_amd import _check_acc_regs as c
    from micromix_amd import build
    bad, examined = c.check_scratch(SCRATCH_OK)
    assert not bad and len(examined) == 2          # any kernel name is examined; static LDS is fine outside the tile kernels
    planted = {
        "spill": SCRATCH_OK.replace("private_segment_fixed_size 0", "private_segment_fixed_size 8", 1),
        "spill in a quantizer": SCRATCH_OK[::-1].replace("0 ezis_dexif_tnemges_etavirp", "25 ezis_dexif_tnemges_etavirp", 1)[::-1],
        "scratch instruction": SCRATCH_OK.replace("\tv_add_u32_e32 v3, v4, v5\n", "\tscratch_load_dword v1, off, off ; 4-byte Folded Reload\n", 1),
        "dynamic stack": SCRATCH_OK.replace("uses_dynamic_stack 0", "uses_dynamic_stack 1"),
        "static LDS in a tile kernel": SCRATCH_OK.replace("group_segment_fixed_size 0", "group_segment_fixed_size 16"),
    }
    for what, text in planted.items():
        bad, _ = c.check_scratch(text)
        assert len(bad) == 1, (what, bad)
        with pytest.raises(RuntimeError, match="scratch"):
            c.verify_scratch(text)
    # the build runs it on the assembly of EVERY product source, and removes the object when it fails
    (tmp_path / "reorder_quantize-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(planted["spill in a quantizer"])
    with pytest.raises(RuntimeError, match="register spill"):
        build.verify_no_scratch(str(tmp_path), "reorder_quantize.hip")
    (tmp_path / "reorder_quantize-hip-amdgcn-amd-amdhsa-gfx950.s").write_text(SCRATCH_OK)
    assert build.verify_no_scratch(str(tmp_path), "reorder_quantize.hip") == 2
    with pytest.raises(RuntimeError, match="no device assembly"):
        build.verify_no_scratch(str(tmp_path), "capi.hip")


W_MODE_PROBE = """
#include "mx_gemm_prelude.h"
namespace mm {
#define MM_NS g256
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 4
#define MM_TM 2
#define MM_TN 4
#define MM_ACC MM_ACC_CLOBBER
#include "mx_gemm_tile.inc"
template __global__ void g256::mx_gemm256_kernel<false, false>(GemmArgs);
}
"""


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_w_mode_256_row_kernel_keeps_a_register_margin(tmp_path):
    """The matching-precision 256 x 256 kernel (fp8 weights: two fragment sets of 8-register fragments) is the kernel closest to its
    register budget: 128 VGPRs beside the 128 accumulators.  Round 5 shipped it one register over (a spill whose reload drained the DMA ring
    every slab).  Besides the scratch guard of every build, this compiles that ONE kernel with four extra values kept alive across its K
    loops (-DMM_PRESSURE=4, mx_gemm_tile.inc): still no scratch, i.e. the product kernel has at least four registers to spare."""
    from micromix_amd import _check_acc_regs as c
    src = tmp_path / "w_mode_probe.hip"
    src.write_text(W_MODE_PROBE)
    out = tmp_path / "k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                    "-DMM_PRESSURE=4", "-I", os.path.join(ROOT, "micromix_amd", "csrc"), str(src), "-o", str(out)],
                   check=True, cwd=str(tmp_path), stderr=subprocess.DEVNULL)
    text = out.read_text()
    bad, examined = c.check_scratch(text)
    assert any("mx_gemm256_kernelILb0ELb0E" in s for s in examined), examined
    assert not bad, bad
    assert not c.check(text)          # and the accumulator AGPRs are still the asm's own
