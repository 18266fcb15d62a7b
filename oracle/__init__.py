"""CPU oracle for the ProCyon forward/generation hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It restates, op by op on torch-CPU tensors, the arithmetic the reference performs
through `transformers` Llama / ESM and its own glue code (SURVEY.md section 8a rows
A0-A12).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; nothing under `procyon_amd/` does.

Pinning status (see DESIGN.md "Oracle"):
  * splitter / pooler / create_mlp / left_pad / splice / greedy+nucleus+beam loops are
    pinned against the reference's own functions, AST-extracted from /root/reference
    and executed in the build container (tests/golden/make_golden.py, fixtures g1-g4,
    g7, g8).
  * Llama and ESM2 layer arithmetic lives in third-party wheels the reference pins
    (transformers==4.31.0, fair-esm==2.0.0) that are absent here; the oracle is pinned
    against the container's transformers 5.15 LlamaForCausalLM / EsmForMaskedLM with
    the version deltas made explicit switches (rope table precision, RMSNorm cast
    order) -- fixtures g5, g6, g9.  The reference's own tests hold no vectors for this
    path, so versus *true* 4.31 / fair-esm numerics parity is
    UNPINNED for the switch defaults that differ from 5.15.
"""
