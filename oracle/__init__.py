"""CPU oracle for the dvae hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(sp-uhh/disentangled-vae, ``packages/models`` + ``packages/processing/stft.py``)
used only as the checker.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``packages/`` + ``disentangled-vae_amd/``) never imports this package and
raises when the HIP library is missing instead of falling back to it.

Parity pin (see DESIGN.md "Oracle"):
  * model / loss / backward / Adam half: pinned against golden vectors captured
    by importing the reference itself on CPU (tests/golden/make_golden.py).
  * forward STFT half: pinned against the reference's own HDF5 fixtures
    (tests/golden/make_stft_golden.py) and pad-rule known answers.
  * ISTFT: parity unpinned by anything in the reference (librosa is not
    installable here); pinned only by torch.istft cross-checks and round trips.
"""
