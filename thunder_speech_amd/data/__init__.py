"""The callers in front of the hot path: batch collation and per-clip audio preparation on the device.
Mirrors the public names of the reference's src/thunder/data/dataloader_utils.py and data/dataset.py (AudioFileLoader)."""
from .dataloader_utils import asr_collate
from .dataset import AudioFileLoader

__all__ = ["asr_collate", "AudioFileLoader"]
