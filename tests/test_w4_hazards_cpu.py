"""The 4-wave prefill kernel issues its MFMAs from inline asm, so hipcc pads none of their hazards
(cdna_hip_programming.md section 5.7).  tools/check_mfma_hazards.py reads the compiled ISA and checks the two
that matter -- a VALU write of an MFMA operand less than two wait states ahead of it, and anything touching an
MFMA's result right behind it -- on every build of the kernel (hipcc cross-compiles here, no GPU needed)."""
import os
import subprocess
import sys
import tempfile

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_mfma_hazards as chk        # noqa: E402

SYNTHETIC = """
_Z4testv:
	v_cvt_pk_bf16_f32 v7, v1, v2
	v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[4:7], a[0:15]
	v_add_f32 v30, v30, v31
	v_cvt_pk_bf16_f32 v7, v1, v2
	s_nop 1
	v_mfma_f32_32x32x16_bf16 a[16:31], v[20:23], v[4:7], a[16:31]
	v_mfma_f32_32x32x16_bf16 v[40:55], v[20:23], a[64:67], 0
	v_add_f32 v30, v30, v31
	v_add_f32 v30, v30, v31
	v_max3_f32 v60, v60, v40, v41
	v_mfma_f32_32x32x16_bf16 v[40:55], v[24:27], a[68:71], v[40:55]
	v_mfma_f32_32x32x16_bf16 v[40:55], v[24:27], a[68:71], v[40:55]
	v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[4:7], a[0:15]
	v_max3_f32 v60, v60, v40, v41
	v_mfma_f32_32x32x16_bf16 v[80:95], v[20:23], a[64:67], 0
	v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[4:7], a[0:15]
	v_mfma_f32_32x32x16_bf16 a[16:31], v[20:23], v[4:7], a[16:31]
	v_max3_f32 v60, v60, v80, v81
	v_mfma_f32_32x32x16_bf16 v[96:111], v[20:23], a[64:67], 0
	s_nop 11
	v_max3_f32 v60, v60, v96, v97
	v_mfma_f32_32x32x16_bf16 v[112:127], v[20:23], a[64:67], 0
	s_cbranch_scc1 .LBB0_2
	s_nop 15
.LBB0_2:
	v_max3_f32 v60, v60, v112, v113
	v_mfma_f32_16x16x32_bf16 v[128:131], v[20:23], a[64:67], 0
	s_nop 7
	v_add_f32 v30, v30, v128
	s_endpgm
"""


def test_checker_sees_both_hazards_and_nothing_else():
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(SYNTHETIC)
    try:
        # flagged: the cvt that writes v7 right ahead of the first MFMA (hazard 2); the max3 that reads v40 TWO plain
        # instructions behind its MFMA; the max3 that reads v40 with only ONE independent MFMA in between; the max3
        # that reads v112 on the path that branches around the s_nop 15.
        # fine: the padded repeat (s_nop 1), the back-to-back accumulation chain, a read behind two independent MFMAs
        # (the matrix pipe is paced), a read behind s_nop 11, a 4-pass MFMA's result behind s_nop 7.
        assert chk.check(f.name) == 4
    finally:
        os.unlink(f.name)


import pytest


def agpr_names_outside_asm(text, kernel):
    """instructions hipcc itself issued (outside ;;#ASMSTART .. ;;#ASMEND) that name an accumulator register, in the
    kernels whose symbol contains `kernel`"""
    import re
    bad, inasm, on = [], False, False
    for ln, line in enumerate(text.split("\n"), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            on = kernel in m.group(1)
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif on and not inasm and t and not t.startswith((";", ".")):
            code = t.split(";")[0]
            if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", code):
                bad.append((ln, code))
    return bad


@pytest.mark.parametrize("name,mfmas", [("prefill_w4_kernel", 500), ("prefill_w4d_kernel", 300)])
def test_compiled_w4_kernel_has_no_unpadded_mfma_hazard(name, mfmas):
    src = os.path.join(ROOT, "starflashattention_amd", "csrc", name + ".hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "w4.s")
        # (SFA_W4_DEV: the bf16 exact-scale pair of prefill_w4_kernel.hip only -- every instantiation runs the same gap
        # program, and the full set takes minutes to compile)
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + ROOT, "-DSFA_W4_DEV", "-S",
                            "--cuda-device-only", src, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        text = open(out).read()
        assert text.count("v_mfma_f32_32x32x16") > mfmas        # the kernels are really in there
        assert chk.check(out, name) == 0
        # the register files stay where the design puts them: no scratch, no VGPR spills, no calls
        assert ".vgpr_spill_count: 0" in text and "scratch_" not in text and "s_swappc" not in text
        if name == "prefill_w4_kernel":
            # O^T and Q^T live in a0..a191 BY NAME inside the asm statements (prefill_w4_kernel.hip, "the asm-owned half
            # of the register file"): hipcc must not put anything of its own there -- it has no reason to touch the
            # accumulator file at all (190 of 256 arch VGPRs used)
            assert agpr_names_outside_asm(text, name) == []
            assert ".agpr_count:     192" in text or ".agpr_count: 192" in text
