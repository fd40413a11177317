"""CPU oracle for the star_flash_attn hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in numpy float64, the *intended* semantics of the
reference's fused decode attention (pure-PyTorch ground truth at
/root/reference/examples/python/testFlashDecoder.py:61-94) and of plain
scaled-dot-product attention (the prefill path, which has no reference code).

Rules (enforced by tests/test_cabi_cpu.py::test_product_never_imports_oracle):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here;
  * the product (starflashattention_amd/, star_flash_attn) never does, and has
    no CPU fallback: it raises when the HIP library is missing.

Parity status: PINNED.  tests/golden/*.npz were produced by importing the
reference's own LlamaAttention in the authoring container
(tests/golden/make_golden.py); tests/test_oracle_golden.py checks this
restatement against them, plus the reference's only known-answer test
("all-ones in -> all 1.0 out", examples/cpp/testFlashDecoder.cc:63-78,116-129).
"""
from .numerics import (  # noqa: F401
    bf16_round, fp16_round, round_to, to_bits16, from_bits16,
)
from .decode_ref import decode_ref, rope_interleaved, rotary_table_ref  # noqa: F401
from .sdpa_ref import sdpa_ref, sdpa_torch_cpu  # noqa: F401
