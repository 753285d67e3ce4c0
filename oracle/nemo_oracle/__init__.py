"""CPU oracle for the nemo map-filtering + detection hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy/scipy restatement of the reference algorithm
(simonsobs/nemo, file:line cited per function).  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` - never by the product
package ``nemo_amd`` (which must fail loudly when its HIP library is missing).

Pinning status:

* the reference's own classes and functions - ``filters.filterMaps`` / ``MatchedFilter.buildAndApply``,
  ``RealSpaceMatchedFilter.buildKernel / buildAndApply / applyFilter``, ``MapFilter.makeNoiseMap``,
  ``photometry.getObjectPositions / findObjects / measureFluxes / getSNRValues``, ``completeness.getRMSTab``,
  ``maps.getPixelAreaArcmin2Map``, ``gnfw`` - were run unmodified in the build container through a stub importer
  (``tests/golden/make_golden.py``, ``make_golden_filters.py``) with only their third-party calls served from outside;
  the vectors are committed under ``tests/golden/`` and this package reproduces them bit for bit
  (``tests/test_cpu_oracle_golden.py``, ``tests/test_cpu_reference_filters.py``).
* what the reference delegates to pixell / astLib / pyccl (FFT normalisation, apodisation, pixel window, template painter,
  WCS, cosmology) is restated from the published behaviour of those packages: PARITY UNPINNED for those calls only (none
  of them is available offline, and the reference holds no golden vectors).
"""
from . import pixell_restated, painter, filters, photometry, maps, completeness  # noqa: F401
