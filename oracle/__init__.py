"""CPU oracle for the audio-diffusion sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``audio_diffusion_amd/`` may import this
package: it is the checker for the HIP path, never the thing measured or shipped.
Allowed importers: ``tests/``, ``__graft_entry__.smoke()``, and the ``cpu_baseline``
leg of ``bench.py``.

The oracle is a functional restatement (plain PyTorch CPU fp32 ops / numpy / scipy)
of the reference algorithm; every function cites the reference file:line it follows
(paths relative to the reference checkout).

Parity pinning:
  * U-Net / DDIM / resamplers / attention: PINNED by golden vectors generated here
    from the imported reference (``tools/make_goldens.py`` -> ``tests/golden/``).
  * PQMF: the reference delegates its two convolutions to the third-party package
    ``cached_conv`` (unpinned ``git+https`` HEAD in the reference's setup.py:37), which
    is absent from the reference tree and from this image.  Its published behaviour
    (centred zero padding of (k-1)//2 ... see pqmf_oracle.py) is restated; the
    reference's own test (test/pqmf_test.py:9-26) pins only the output SHAPE, so PQMF
    sample values are "parity unpinned" beyond shape + round-trip reconstruction +
    agreement with the reference's own (cached_conv-free) polyphase_forward.
"""
