from .audio_transforms import AddNoise  # noqa: F401
from .video_transforms import (CenterCrop, Compose, Normalise, RandomCrop, RandomHorizontalFlip, TimeMasking,  # noqa: F401
                               VideoClip, VideoSpeedRate)
