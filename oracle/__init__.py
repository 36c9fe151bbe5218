"""CPU oracle for the ViT-AE++ pre-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the timed CPU baseline.  The product package
(``vit_ae_plus_plus_amd``) never imports it and fails loudly when its HIP library is
missing.

Contents
--------
``mae_ref.py``      plain-PyTorch (CPU, fp32/fp64) functional restatement of the reference
                    path: every function cites the reference file:line it follows.
``train_ref.py``    restatement of ``utils/train_one_epoch.py:21-110`` on top of ``mae_ref``.
``_refharness.py``  imports the *real* reference from ``/root/reference`` (only exists in the
                    build container) behind the three stubs of SURVEY Appendix A.
``gen_golden.py``   runs the real reference and writes ``tests/golden/*.npz`` fixtures.

Pinning status: **pinned** — ``tests/test_oracle_golden.py`` checks ``mae_ref`` against the
fixtures that ``gen_golden.py`` produced from the imported reference (the reference ships no
golden vectors of its own).
"""
