"""TEST INFRASTRUCTURE ONLY.

`oracle/` holds a CPU restatement of the LGTEUN unfolding hot path (reference:
models/unlg_former.py, models/common/LGT.py, models/common/basic_module_unformer_v2.py).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker / the timed CPU baseline.  The product path (`lgteun_amd`) never
imports this package and fails loudly when its HIP extension is missing.

Parity pin: the reference ships no tests / golden vectors, so the oracle is
pinned by outputs of the reference itself, generated in the build container by
`tools/gen_goldens.py` (which imports /root/reference) and committed as plain-array
fixtures under `tests/golden/`.
"""
