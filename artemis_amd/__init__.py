"""artemis_amd: MI355X-native finite-volume hydro update for Artemis (gas/dust CalculateFluxes
-> ApplyUpdate -> FluxSource -> SetAuxillaryFields -> ConsToPrim/PrimToCons) behind the
C ABI of include/artemis_hip.h.  The product is libartemis_hip.so (hand-written HIP for
gfx950); this package is the thin host-side binding."""
__version__ = "0.1.0"
