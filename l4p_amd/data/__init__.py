"""Caller side of the hot path: clip preparation on the GPU (mirror of the reference's l4p/data package for the
demo's generic-video case)."""
from .video_dataset import VideoDataset, pil_resize_blur_resize, prepare_clip  # noqa: F401
