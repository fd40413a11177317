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
	v_max3_f32 v60, v60, v40, v41
	v_mfma_f32_32x32x16_bf16 v[40:55], v[24:27], a[68:71], v[40:55]
	v_mfma_f32_32x32x16_bf16 v[40:55], v[24:27], a[68:71], v[40:55]
	s_endpgm
"""


def test_checker_sees_both_hazards_and_nothing_else():
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(SYNTHETIC)
    try:
        # line 4: cvt writes v7, an operand of the very next MFMA; line 10: max3 reads v40 right behind its MFMA.
        # The padded repeat (s_nop 1) and the back-to-back accumulation chain are fine.
        assert chk.check(f.name) == 2
    finally:
        os.unlink(f.name)


import pytest


@pytest.mark.parametrize("name,mfmas", [("prefill_w4_kernel", 500), ("prefill_w4d_kernel", 300)])
def test_compiled_w4_kernel_has_no_unpadded_mfma_hazard(name, mfmas):
    src = os.path.join(ROOT, "starflashattention_amd", "csrc", name + ".hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "w4.s")
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + ROOT, "-S", "--cuda-device-only",
                            src, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        text = open(out).read()
        assert text.count("v_mfma_f32_32x32x16") > mfmas        # the kernels are really in there
        assert chk.check(out, name) == 0
        # the register files stay where the design puts them: no scratch, no VGPR spills
        assert ".vgpr_spill_count: 0" in text and "scratch_" not in text
