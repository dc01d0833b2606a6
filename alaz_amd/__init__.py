"""alaz_amd — MI355X-native ServiceGraph engine behind getanteon/alaz's aggregator -> datastore seam.

The compute path is the HIP library ``alaz_amd/lib/libservicegraph.so`` (C ABI in
include/servicegraph.h).  Importing this package does not load it; ``alaz_amd.engine`` does, and
fails loudly if it is missing or no gfx950 device is usable.  There is no CPU fallback.
"""
__all__ = ["replay", "weights"]
