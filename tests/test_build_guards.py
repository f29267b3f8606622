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
    (it did once, in a kernel that needed more than its VGPR budget: tools/check_acc_regs.py has the story)"""
    import check_acc_regs
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-S",
                        "--cuda-device-only", os.path.join(ROOT, "micromix_amd", "csrc", "mx_gemm256.hip"), "-o", out],
                       check=True, cwd=tmp, stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad, examined = check_acc_regs.check_counted(text)
    assert len(examined) >= check_acc_regs.EXPECTED_KERNELS, examined     # the regex must have matched every tile kernel
    assert not bad, "\n".join(f"{s}: {c}" for s, c in bad[:10])


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
