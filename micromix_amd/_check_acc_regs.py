"""Build-time guard for the tile kernels of mx_gemm256.hip: their fp32 accumulators live in AGPRs a[0 : NACC-1] that only the inline
asm touches (MFMAs, v_accvgpr_read/write) -- the compiler sees them as clobbered, not as live, so under register pressure it may
allocate a temporary there between two asm statements (it did: `ds_read2st64_b64 a[0:3]` in the 64x128 matching-precision kernel
while the 4-wave tiles still kept their accumulators this way, which corrupted results; they now leave them to the compiler).  This script compiles the file to assembly and fails if any instruction other than the inline asm's own
names an accumulator register of its kernel.  Lives in the package because micromix_amd.build runs it on every build of that file;
`python tools/check_acc_regs.py [-DFLAG ...]` is the command-line front end (exit code 1 on a violation)."""
import os, re, subprocess, sys, tempfile
PKG = os.path.dirname(os.path.abspath(__file__))
NACC = {"g256": 128, "g128": 64, "g64": 32, "g32": 0, "g32n": 0, "g16": 0}   # the 4-wave tiles leave their accumulators to the compiler
ASM_OWN = re.compile(r"^\s*(v_mfma_scale_f32_32x32x64_f8f6f4|v_mfma_f32_32x32x16_bf16|v_accvgpr_read_b32|v_accvgpr_write_b32)\b")


def agprs(line):
    """AGPR indices named by an instruction line (single registers a5 / a[5] and tuples a[0:3])"""
    out = []
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", line):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\ba\[(\d+)\]|\ba(\d+)\b", line):
        out.append(int(m.group(1) or m.group(2)))
    return out


EXPECTED_KERNELS = 12   # mx_gemm256.hip -- g256: 2 + 2 grouped + 1 fused gate/up; g128: 2 + 2 split-K + 2 grouped + 1 fused gate/up
EXPECTED_SMALL = 4      # mx_gemm_tiles_small.hip -- g64: 2 + 2 grouped (the 4-wave tiles leave their accumulators to the compiler)


def check(asm_text):
    """violations only (see check_counted)"""
    return check_counted(asm_text)[0]


def check_counted(asm_text):
    """(violations, symbols of the kernels with asm-owned accumulators that were examined).  A caller must also require
    len(examined) >= EXPECTED_KERNELS: a name-mangling change would otherwise make the check pass with nothing examined."""
    bad, examined = [], []
    for m in re.finditer(r"^(_ZN2mm\d(g(?:256|128|64|32n|32|16))(?:17|21|25)mx_gemm256_(?:grouped_|act_)?kernel\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel",
                         asm_text, re.S | re.M):
        sym, ns, body = m.group(1), m.group(2), m.group(3)
        n = NACC[ns]
        if n:
            examined.append(sym)
        for line in body.split("\n"):
            code = line.split(";")[0]
            if not code.strip() or code.strip().startswith("."):
                continue
            regs = [r for r in agprs(code) if r < n]
            if not regs:
                continue
            own = ASM_OWN.match(code) is not None
            # the asm's own accumulator accesses use the bracket form a[i] / a[i:j]; a compiler-generated accvgpr move prints a5
            if own and not re.search(r"\ba\d+\b", code):
                continue
            bad.append((sym, code.strip()))
    return bad, examined


# ---------------------------------------------------------------------------------------------------------
# mx_gemm_stream.hip (the weight-streaming kernels, every M <= 64 path): NACC = 12 * F * T16 accumulators a[0 : NACC-1] named only by
# inline asm, and registers that the hardware writes asynchronously AFTER the asm statement that requested them has returned
# (buffer_load_dwordx2 / global_load_dwordx4 issued from asm with "=&v" outputs, merged across the segment branches).  Three checks per
# kernel: (1) as above, no instruction but the asm's own names an accumulator; (2) no scratch (a spilled accumulator copy or pending
# load destination would be silently wrong); (3) no instruction reads -- or overwrites -- the destination of a vector-memory load before an
# s_waitcnt vmcnt that covers it (vmcnt counts loads, stores and LDS-DMA together, in issue order; basic-block boundaries reset
# the model, so the scan is exact for straight-line code and silent across branches).
# ---------------------------------------------------------------------------------------------------------
STREAM_OWN = re.compile(r"^\s*(v_mfma_scale_f32_16x16x128_f8f6f4|v_accvgpr_read_b32|v_accvgpr_write_b32)\b")
STREAM_KERNEL = re.compile(r"^(_ZN2mm6stream\d+(mx_gemm_stream_kernel|mx_gemm_stream_grouped_kernel|mx_qlinear_stream_kernel|mx_qlinear_stream_rms_kernel)I((?:Li\d+E)+)Lb[01]E\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel",
                           re.S | re.M)
# ... and the launches with the activation inside (stream_body ACT: F = 4): mx_gemm_stream_act_kernel<T16 = 1, 2>,
# mx_qlinear_stream_act_kernel<RMS>
STREAM_ACT_KERNEL = re.compile(r"^(_ZN2mm6stream\d+(mx_gemm_stream_act_kernel|mx_qlinear_stream_act_kernel)I(?:Lb[01]|Li(\d+))EE\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel",
                               re.S | re.M)
EXPECTED_STREAM_KERNELS = 68   # 22 plain + 18 grouped + 16 with the quantizer inside + 8 with norm and quantizer inside + 4 with the activation inside
VMEM = re.compile(r"^\s*(buffer_|global_|scratch_|flat_)(load|store|atomic)")


def vgprs(text):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.append(int(m.group(1)))
    return out


def stream_nacc(kind, ints):
    """accumulator registers of an instantiation, from the template arguments in its mangled name: <F, T16, D, NW> (plain, grouped) or
    <F, D, NW> with T16 = 1 (quantizer inside)"""
    v = [int(x) for x in re.findall(r"Li(\d+)E", ints)]
    f, t16 = (v[0], 1) if kind.startswith("mx_qlinear_stream") else (v[0], v[1])
    return 12 * f * t16


def pending_load_violations(body):
    """check (3): instructions that touch the destination of a load still counted in vmcnt"""
    bad, pending = [], []          # pending: per outstanding VMEM op, the set of destination VGPRs (empty for stores / LDS-DMA)
    for line in body.split("\n"):
        code = line.split(";")[0].strip()
        if not code:
            continue
        if code.endswith(":") or code.startswith("s_cbranch") or code.startswith("s_branch") or code.startswith("s_setpc") or code.startswith("s_endpgm"):
            pending = []
            continue
        if code.startswith("."):
            continue
        m = re.match(r"s_waitcnt\b(.*)", code)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if v:
                n = int(v.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        live = set().union(*pending) if pending else set()
        if live:
            hit = [r for r in vgprs(code) if r in live]
            if hit:
                bad.append(code)
        if VMEM.match(code):
            dst = set()
            if re.match(r"^\s*(buffer|global|scratch|flat)_load", code) and " lds" not in code and not code.rstrip().endswith("lds") and "_load_lds_" not in code:
                first = code.split(None, 1)[1].split(",")[0]
                dst = set(vgprs(first))
            pending.append(dst)
    return bad


def check_stream(asm_text):
    """(violations, kernels examined) for the assembly of mx_gemm_stream.hip"""
    bad, examined = [], []
    found = [(m.group(1), stream_nacc(m.group(2), m.group(3)), m.group(4)) for m in STREAM_KERNEL.finditer(asm_text)]
    found += [(m.group(1), 48 * int(m.group(3) or 1), m.group(4)) for m in STREAM_ACT_KERNEL.finditer(asm_text)]      # <T16> / <RMS>: 12 * 4 * T16
    for sym, n, body in found:
        examined.append(sym)
        lines = [l.split(";")[0] for l in body.split("\n")]
        # The accumulators come to life at the kernel's first asm-owned instruction (acc_zero, placed BEHIND the quantization phase of the
        # kernels that quantize their rows themselves).  In front of it the compiler may park values in those AGPRs -- provided the code in
        # front of it can never run again: no branch at or behind that point may target a label defined in front of it.
        first = next((k for k, c in enumerate(lines) if STREAM_OWN.match(c) and re.search(r"\ba\[", c)), len(lines))
        early_labels = {m2.group(1) for c in lines[:first] for m2 in [re.match(r"^(\.?\w+):\s*$", c.strip())] if m2}
        reentered = [c.strip() for c in lines[first:] if re.match(r"\s*s_c?branch\w*\s", c) and c.split()[-1] in early_labels]
        parked = []          # compiler uses of accumulator AGPRs in front of that point: fine unless that code can run again
        for k, code in enumerate(lines):
            if not code.strip():
                continue
            if code.strip().startswith("."):
                sm = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", code)
                if sm and int(sm.group(1)) != 0:
                    bad.append((sym, f"scratch: {sm.group(1)} bytes per lane (spill)"))
                continue
            regs = [r for r in agprs(code) if r < n]
            if not regs:
                continue
            if STREAM_OWN.match(code) is not None and not re.search(r"\ba\d+\b", code):
                continue
            if k < first:
                parked.append(code.strip())     # the accumulators are not live yet (see above)
                continue
            bad.append((sym, code.strip()))
        if parked and reentered:
            bad.append((sym, f"parks values in accumulator AGPRs in front of the first accumulator instruction ({parked[0]}) and can branch back there: {reentered[0]}"))
        bad += [(sym, "reads a load destination before its s_waitcnt vmcnt: " + c) for c in pending_load_violations(body)]
    return bad, examined


ANY_KERNEL = re.compile(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", re.S | re.M)


def check_pending(asm_text):
    """(violations, kernels examined): check (3) alone -- no instruction touches the destination of a vector-memory load before an
    s_waitcnt vmcnt that covers it -- for files whose kernels issue plain loads from inline asm with "=&v" outputs and certify them
    with ONE counted wait (rmsnorm_quantize.hip: the ring kernel's prologue; qlinear_decode.hip: mx_decode_quant.h's early phase)"""
    bad, examined = [], []
    for m in ANY_KERNEL.finditer(asm_text):
        examined.append(m.group(1))
        bad += [(m.group(1), "reads a load destination before its s_waitcnt vmcnt: " + c) for c in pending_load_violations(m.group(2))]
    return bad, examined


def verify_pending(asm_text):
    bad, examined = check_pending(asm_text)
    if bad:
        raise RuntimeError("a register that an asm-issued load is still writing is read before its wait (results would be corrupted):\n" +
                           "\n".join(f"  {s}: {c}" for s, c in bad[:10]))
    if not examined:
        raise RuntimeError("pending-load check found no kernel")
    return len(examined)


# ---------------------------------------------------------------------------------------------------------
# EVERY kernel of EVERY product object (round 6): no scratch.  A spilled VGPR is not only slow: its reload is a
# scratch_load, hipcc certifies it with `s_waitcnt vmcnt(0)`, and inside a K loop that drains the LDS-DMA ring once per slab -- the
# matching-precision 256-row kernel lost 2.3x that way (124 us against 53) while every numerical test stayed green.  The tile kernels
# also may not own static LDS: FragOfsC (mx_gemm_tile.inc) relies on the dynamic LDS starting at byte 0.
# ---------------------------------------------------------------------------------------------------------
TILE_KERNEL = re.compile(r"mx_gemm256_(?:grouped_|act_)?kernel")


def check_scratch(asm_text):
    """(violations, kernels examined): private_segment_fixed_size != 0, a dynamic stack, or scratch instructions in any kernel;
    static LDS in a tile kernel"""
    bad, examined = [], []
    for m in ANY_KERNEL.finditer(asm_text):
        sym, body = m.group(1), m.group(2)
        examined.append(sym)
        sm = re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body)
        if sm is None:
            bad.append((sym, "no .amdhsa_private_segment_fixed_size directive found"))
        elif int(sm.group(1)) != 0:
            bad.append((sym, f"scratch: {sm.group(1)} bytes per lane (register spill)"))
        if re.search(r"\.amdhsa_uses_dynamic_stack\s+1", body):
            bad.append((sym, "dynamic stack"))
        for line in body.split("\n"):
            code = line.split(";")[0].strip()
            if code.startswith("scratch_"):
                bad.append((sym, "scratch instruction: " + code))
                break
        if TILE_KERNEL.search(sym):
            gm = re.search(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", body)
            if gm is None or int(gm.group(1)) != 0:
                bad.append((sym, f"static LDS in a tile kernel ({gm.group(1) if gm else '?'} bytes): the dynamic LDS no longer starts at 0"))
    return bad, examined


def verify_scratch(asm_text):
    bad, examined = check_scratch(asm_text)
    if bad:
        raise RuntimeError("a kernel of the product library uses scratch memory (a spilled register; its reload drains the DMA queue):\n" +
                           "\n".join(f"  {s}: {c}" for s, c in bad[:10]))
    return len(examined)


def verify_stream(asm_text):
    bad, examined = check_stream(asm_text)
    if bad:
        raise RuntimeError("mx_gemm_stream.hip: hipcc touched asm-owned registers of a streaming kernel (results would be corrupted):\n" +
                           "\n".join(f"  {s}: {c}" for s, c in bad[:10]))
    if len(examined) < EXPECTED_STREAM_KERNELS:
        raise RuntimeError(f"accumulator-register check found {len(examined)} streaming kernels, expected >= {EXPECTED_STREAM_KERNELS} "
                           "(kernel names changed? update micromix_amd/_check_acc_regs.py)")
    return len(examined)


# tile kernels with asm-owned accumulators per translation unit
EXPECTED_BY_FILE = {"mx_gemm256.hip": EXPECTED_KERNELS, "mx_gemm_tiles_small.hip": EXPECTED_SMALL}


def verify(asm_text, expected=EXPECTED_KERNELS):
    """raises RuntimeError on a violation or when fewer kernels than expected were found; returns the number examined"""
    bad, examined = check_counted(asm_text)
    if bad:
        raise RuntimeError("hipcc allocated a temporary in an accumulator AGPR of a tile kernel (results would be corrupted):\n" +
                           "\n".join(f"  {s}: {c}" for s, c in bad[:10]))
    if len(examined) < expected:
        raise RuntimeError(f"accumulator-register check found {len(examined)} tile kernels, expected >= {expected} "
                           "(kernel names changed? update micromix_amd/_check_acc_regs.py)")
    return len(examined)


def main():
    rc = 0
    for name, expected in EXPECTED_BY_FILE.items():
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(PKG, "csrc", name)
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                   src, "-o", os.path.join(tmp, "k.s")] + sys.argv[1:]
            subprocess.run(cmd, check=True, cwd=tmp)
            bad, examined = check_counted(open(os.path.join(tmp, "k.s")).read())
        for sym, code in bad[:20]:
            print(f"accumulator register used by the compiler in {sym}: {code}")
        print(f"{name}: {len(bad)} violation(s) in {len(examined)} tile kernels with asm-owned accumulators (expected >= {expected})")
        if bad or len(examined) < expected:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
