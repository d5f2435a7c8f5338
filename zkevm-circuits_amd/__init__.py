"""zkevm-circuits_amd: MI355X (gfx950) accelerator for the halo2 `create_proof` polynomial pipeline.

The directory name carries a hyphen, so import it through `zkhip_pkg.load()` at the repository root
(registered in sys.modules as `zkevm_circuits_amd`).
"""
from . import ffi, plonk  # noqa: F401
from .ffi import Context, Domain, ZkhipError  # noqa: F401
