"""CPU oracle for the SEDnet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and only as the checker / the timed CPU
baseline.  ``sed_crnn_amd`` never imports this package.

Contents
--------
crnn_ref.py     parameterised torch.nn (CPU, fp32) restatement of the two
                reference networks (sed.py:82-112, crnn_lightning.py:41-73),
                the focal loss (crnn_lightning.py:27-35) and one fit step.
metrics_ref.py  numpy restatement of metrics.py:14-74 / utils.py:4-12.
logmel_ref.py   numpy restatement of feature.py:55-59 (librosa defaults).
make_goldens.py imports /root/reference in the build container and writes the
                golden vectors under tests/golden/ (numbers only).

Pinning: crnn_ref/metrics_ref are pinned by goldens captured from the imported
reference (tests/golden/*.npz, generator committed).  logmel_ref is "parity
unpinned": librosa is absent from this image and the reference holds no
fixture for feature.py, so it restates librosa's published defaults only.
"""
