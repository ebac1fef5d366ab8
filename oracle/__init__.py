"""CPU oracle for the MSI infer->render hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy fp32 op-by-op for the geometry,
torch-CPU fp32 for conv / LayerNorm) of the reference's algorithm for the path
`BASELINE.json:north_star` names.  Every function cites the reference
file:line it follows (paths relative to the reference checkout).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product package (`matryodshka_amd/`) never imports, calls or
links anything in here.

PARITY UNPINNED.  The reference is Python 2.7 + TensorFlow 1.14 graph code; it
has no tests, no golden vectors and no fixtures for this path, TensorFlow is not
installed in the build container and there is no network, so the reference
itself cannot be run to produce vectors.  The oracle is therefore pinned only by
the analytic known-answer tests in `tests/test_oracle_kat.py` and
by self-generated regression fixtures under `tests/golden/` (generator script
committed).  Statements about TensorFlow op semantics (SAME padding, LayerNorm
epsilon, tf.linspace, floor-mod, add_n order) are knowledge of TF 1.14, not
something verified against a TF binary here.

TRANSCRIPTION CROSS-CHECK (round 4; does not change "unpinned").  `oracle/crosscheck_reference.py` (build container
only: it reads /root/reference where it lies) imports the reference's OWN geometry/*.py, matryodshka/nets.py and
matryodshka/msi.py with `tensorflow` / `slim` / `tensorflow_graphics` replaced by a numpy stand-in that evaluates each
TF op the way this oracle assumes TF does, runs the whole infer -> render path from the reference's function bodies and
asserts bit-equality with this package: 57 stages, all bit-identical (profiles/r04_crosscheck_reference.txt;
tests/test_oracle_kat.py::test_reference_function_bodies_agree_with_the_oracle runs it where the reference exists).
It pins no TensorFlow semantics -- the stand-in IS the oracle's reading of them -- but it removes the risk of a mis-read
association, operand order, branch, axis or index in the restatement.
"""
