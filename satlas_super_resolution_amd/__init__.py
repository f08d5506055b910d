"""MI355X-native ESRGAN hot path of allenai/satlas-super-resolution.

Host side mirrors the reference's plugin interface (registry names SSR_RRDBNet,
SSR_UNetDiscriminatorSN, SSRESRGANModel); device side is libssr_hip.so (include/ssr_hip.h)."""
__version__ = "0.1.0"
