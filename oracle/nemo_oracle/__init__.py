"""CPU oracle for the nemo map-filtering + detection hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy/scipy restatement of the reference algorithm
(simonsobs/nemo, file:line cited per function).  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` - never by the product
package ``nemo_amd`` (which must fail loudly when its HIP library is missing).

Pinning status:

* numpy/scipy-only reference functions (``photometry.getObjectPositions``, ``MapFilter.makeNoiseMap``,
  ``RealSpaceMatchedFilter.applyFilter`` without bckSub) are pinned against the reference itself, run in
  the build container through a stub importer (``tests/golden/make_golden.py``), with the vectors
  committed under ``tests/golden/``.
* everything that goes through pixell / astLib / pyccl in the reference (FFT normalisation, apodisation,
  pixel window, template painter, WCS) is restated from the published behaviour of those packages:
  PARITY UNPINNED for those pieces (none of them is available offline, and the reference holds no
  golden vectors).
"""
from . import pixell_restated, painter, filters, photometry, maps, completeness  # noqa: F401
