"""CPU oracle for the pb_sed FBCRNN/BiCRNN hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product path (``pb_sed_amd``) never does and fails loudly when
its HIP library is missing.

Parity status (see DESIGN.md "Oracle"):
* reference-OWNED maths (losses, inference heads, median/step filters, ensemble
  post-processing) are pinned by golden vectors produced by executing the reference's
  own source under import shims (``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``).
* third-party layers (padertorch/paderbox STFT, MelTransform, Normalization, CNN2d/CNN1d,
  GRU wrapper; sed_scores_eval event extraction) are NOT present in /root/reference and
  cannot be installed: they are restated from their published semantics on stock
  ``torch`` -> **parity unpinned** for those pieces (``torch.nn.GRU`` itself is exact).
"""
