"""ISA-level guard of the hand-placed wait states in the one-pass kernels.

fused2.hip / fused3.hip issue some MFMAs as inline-asm statements (accumulators tied in place); hipcc's
hazard recognizer does not look inside an asm statement, so the wait states between a VALU instruction
that writes an operand register and the MFMA that reads it are placed by hand (`s_nop` in
settle_operands / onehot_bf16, csrc/onepass.hpp).  A compiler upgrade could reorder or drop them
silently; the numerical stress tests would only catch that on a GPU.  This test disassembles the
gfx950 code objects of the built library ON THE CPU and checks the property itself: in every
fused2 / fused3 kernel, no v_mfma reads (as A or B operand) a VGPR that a VALU instruction wrote fewer
than REQUIRED wait states earlier in the same basic block."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "duckdb-imputation_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
REQUIRED = 2          # wait states (independent instructions or s_nop counts) between the VALU write and the MFMA read


def _disassemble(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                          capture_output=True, text=True).stdout


def _regs(tok):
    """'v[4:7]' -> {4..7}, 'v12' -> {12}; anything else (a[..], s.., literals) -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _violations(text, name_filter):
    bad, seen = [], 0
    func, window = None, []            # window: (mnemonic, written vgprs, wait states it provides)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            func, window = m.group(1), []
            continue
        ins = line.strip()
        if not ins or func is None or name_filter not in func:
            continue
        ins = re.sub(r"\s*//.*$", "", ins)
        parts = ins.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")):
            window = []
            continue
        if op.startswith("v_mfma"):
            seen += 1
            reads = _regs(args[1]) | _regs(args[2]) if len(args) >= 3 else set()
            states = 0
            for pop, pw, pstates in reversed(window):
                if states >= REQUIRED:
                    break
                if pw & reads:
                    bad.append((func[:60], pop, ins, states))
                    break
                states += pstates
        written = set()
        if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_cmpx")) and args:
            written = _regs(args[0])
        states = int(args[0]) + 1 if op == "s_nop" and args else 1
        window.append((op, written, states))
        window = window[-8:]
    return bad, seen


@pytest.mark.parametrize("stem,kernel", [("fused2", "fused2_kernel"), ("fused3", "fused3_kernel")])
def test_no_mfma_reads_a_register_a_valu_instruction_just_wrote(tmp_path, stem, kernel):
    obj = os.path.join(BUILD, stem + ".o")
    if not os.path.exists(obj):
        pytest.skip("library not built (run __graft_entry__.build())")
    bad, seen = _violations(_disassemble(obj, str(tmp_path)), kernel)
    assert seen > 1000, "the disassembly holds the kernels' MFMAs"
    assert not bad, "VALU write -> MFMA read without wait states: %s" % bad[:5]
