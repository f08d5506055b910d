"""Dataset plugins (BasicSR DATASET_REGISTRY names of /root/reference/ssr/data)."""
from .s2naip_dataset import S2NAIPDataset, has_black_pixels, read_png  # noqa: F401
