"""CPU oracle for the fxs MTIP phasing path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain numpy/scipy restatement of the reference algorithm
(European-XFEL/xFrame, ``xframe/projects/fxs/reconstruct.py`` and its
``projectLibrary``).  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- there only as the checker / the timed
CPU baseline, never as the thing shipped.  ``xframe_amd`` (the product) never
imports from here and fails loudly when its HIP library is missing.

Parity pinning
--------------
* Everything except the spherical-harmonic transform itself is pinned against
  golden vectors produced by importing the reference's own numeric modules in
  the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
  ``tests/test_oracle_golden.py`` checks the restatement against them).
* The SHT arithmetic lives in the third-party C library ``shtns`` (pyproject
  extra ``fxs``, version unpinned; call sites
  ``xframe/externalLibraries/shtns_plugin.py:20,130-131,205-261``), which is not
  vendored in the reference and not installed here, and no reference test pins
  its values: **the SHT is "parity unpinned"**.  ``oracle/sht.py`` restates the
  published convention (orthonormal Y_lm, Condon-Shortley phase, Gauss-Legendre
  nodes north->south, index l(l+1)+m) and is pinned by analytic known answers
  (``tests/test_oracle_sht.py``).  When the reference operators are driven to
  produce golden vectors, this SHT is what is injected at the reference's
  ``xframe.library.mathLibrary.shtns`` slot (``mathLibrary.py:29-34``).
"""
from . import sht, hankel, fourier, projections, mtip  # noqa: F401
