"""rd_vio_amd -- MI355X (gfx950) implementation of rd_vio's data-parallel hot path.

The product is ``librdvio_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/rdvio_hip.h``).  This package is the thin Python binding used by tests/ and
bench.py; it mirrors the reference's seam names (rdvio::Image, PreIntegrator, Solver
factor evaluation) so parity tests read like calls on the reference's own classes.

There is no CPU fallback: if the shared library is missing or no GPU is present the
calls raise.  (The CPU oracle lives in ``oracle/`` and is never imported from here.)
"""
from .binding import (  # noqa: F401
    RdvioError, Context, HipImage, PyrLayout, lib_path, load_library, have_gpu,
    STATE_SIZE, PREINT_SIZE,
)
