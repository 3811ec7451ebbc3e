"""bgls_amd -- MI355X-native aggregate-signature verification engine (hot path of Project-Arda/bgls).

The product is bgls_amd/libbgls_hip.so (HIP kernels behind the C ABI in include/bgls_hip.h);
`curves` and `bgls` are the thin host-side mirror of the reference's Go interface for that path."""
from . import curves, bgls  # noqa: F401
from .curves import Altbn128, Bls12, AggregatePoints, ScalePoints  # noqa: F401
