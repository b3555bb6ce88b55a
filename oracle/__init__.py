"""CPU oracle for the IDEAL-NeRF per-ray hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- there as the checker / the timed CPU
baseline, never as the thing shipped.  The product path (``ideal-nerf_amd/``)
never imports this package and fails loudly when its HIP library is missing.

Pinning status: PINNED.  Every function in ``render_oracle.py`` is checked
against outputs of the reference itself (``/root/reference`` imported on CPU in
the build container by ``tests/golden/make_golden.py``; vectors committed under
``tests/golden/*.npz``) by ``tests/test_oracle_golden.py``.  ``philox.py`` (the table of the library's in-kernel
random draws -- the reference has no counterpart: it calls ``torch.rand``) is pinned by the generator's published
known-answer vectors.
"""
from . import philox  # noqa: F401
from .render_oracle import *  # noqa: F401,F403
