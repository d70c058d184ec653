"""Container-only torchvision stand-in (see ../cv2.py). Only what the reference FCN imports."""
from . import transforms  # noqa: F401
