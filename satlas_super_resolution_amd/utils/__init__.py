"""Host-side helpers of the inference path (mirror of /root/reference/ssr/utils)."""
from .infer_utils import format_s2naip_data, stitch, stitch_arrays, quantize_output, infer_chunks  # noqa: F401
from .model_utils import build_network  # noqa: F401
