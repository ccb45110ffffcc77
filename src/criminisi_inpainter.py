"""Drop-in for the reference's ``src/criminisi_inpainter.py``.

``src/utils.py:7`` of Theosdoor/Image-Processing does
``from src.criminisi_inpainter import Criminisi_Inpainter``; placing this package's ``src/`` in
front of the reference's makes ``scripts/main.py`` and ``run_pipeline.sh`` use the MI355X engine
unchanged.  This module deliberately imports neither cv2 nor any CPU implementation.
"""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "image-processing_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from crim_hip_inpainter import Criminisi_Inpainter, CriminisiInpainter, inpaint_many  # noqa: E402,F401
