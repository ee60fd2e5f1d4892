"""Drop-in shim: `import fusion_detection` resolves to the MI355X-native module (put this directory on PYTHONPATH
in place of the reference's src/).  See INTEGRATION.md."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from bevfusion_multimodal_3d_object_detection_amd.fusion_detection import *  # noqa: F401,F403,E402
from bevfusion_multimodal_3d_object_detection_amd.fusion_detection import _nms, _topk  # noqa: F401,E402
