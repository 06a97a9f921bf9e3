"""MI355X-native EPro-PnP layer.  Same import surface as the reference's `epropnp` package
(`from epropnp.epropnp import EProPnP6DoF`, ...); the LM solver and the AMIS sampler run as hand-written
HIP kernels behind a C ABI (include/epropnp_hip.h)."""
